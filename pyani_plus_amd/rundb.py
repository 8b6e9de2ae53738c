"""Run driver and persistence for the ``sourmash-hip`` method (stdlib ``sqlite3``).

The reference's Python (Typer CLI, SQLAlchemy ORM, snakemake) does not travel to the GPU
box, so this module is the build's own counterpart of ``cli_sourmash`` +
``start_and_run_method`` + ``run_method`` minus snakemake (pyani_plus/public_cli.py:115-329,
598-639) and of the parts of ``db_orm`` they use (last row):

* FASTA enumeration by the four extensions +- ``.gz`` (pyani_plus/utils.py:226-242)
* genome identity = md5 of the decompressed bytes (utils.py:142-196); length = sum of
  residues, description = first title (db_orm.py:832-866); duplicate md5 aborts
  (public_cli.py:165-171)
* the same tables / constraints as ``Base.metadata.create_all``
* JSON column import with INSERT OR IGNORE (private_cli.py:507-614, db_orm.py:1076)
* ``cache_comparisons``: N x N matrices over sorted md5, pandas ``to_json(orient="split")``
  (db_orm.py:393-466) -- built without the reference's O(N^3) ``hashes.index`` loop.

Databases written here can be opened by the reference and vice versa.
"""

from __future__ import annotations

import datetime
import logging
import sqlite3
import sys
import tempfile
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import wire
from ._capi import HipBackendError
from .methods import sourmash_hip

FASTA_EXTENSIONS = {".fasta", ".fas", ".fna", ".fa"}  # pyani_plus/__init__.py:48

SCHEMA = """
CREATE TABLE IF NOT EXISTS genomes (
    genome_hash VARCHAR NOT NULL, path VARCHAR NOT NULL, length INTEGER NOT NULL, description VARCHAR NOT NULL,
    CONSTRAINT pk_genomes PRIMARY KEY (genome_hash));
CREATE TABLE IF NOT EXISTS configurations (
    configuration_id INTEGER NOT NULL, method VARCHAR NOT NULL, program VARCHAR NOT NULL, version VARCHAR NOT NULL,
    fragsize INTEGER, mode VARCHAR, kmersize INTEGER, minmatch FLOAT, extra VARCHAR,
    CONSTRAINT pk_configurations PRIMARY KEY (configuration_id),
    CONSTRAINT uq_configurations_method UNIQUE (method, program, version, fragsize, mode, kmersize, minmatch, extra));
CREATE TABLE IF NOT EXISTS comparisons (
    comparison_id INTEGER NOT NULL, query_hash VARCHAR NOT NULL, subject_hash VARCHAR NOT NULL,
    configuration_id INTEGER NOT NULL, identity FLOAT, aln_length INTEGER, sim_errors INTEGER, cov_query FLOAT,
    cov_subject FLOAT, uname_system VARCHAR NOT NULL, uname_release VARCHAR NOT NULL, uname_machine VARCHAR NOT NULL,
    CONSTRAINT pk_comparisons PRIMARY KEY (comparison_id),
    CONSTRAINT uq_comparisons_query_hash UNIQUE (query_hash, subject_hash, configuration_id),
    CONSTRAINT fk_comparisons_query_hash_genomes FOREIGN KEY(query_hash) REFERENCES genomes (genome_hash),
    CONSTRAINT fk_comparisons_subject_hash_genomes FOREIGN KEY(subject_hash) REFERENCES genomes (genome_hash),
    CONSTRAINT fk_comparisons_configuration_id_configurations FOREIGN KEY(configuration_id)
        REFERENCES configurations (configuration_id));
CREATE TABLE IF NOT EXISTS runs (
    run_id INTEGER NOT NULL, configuration_id INTEGER NOT NULL, cmdline VARCHAR NOT NULL,
    fasta_directory VARCHAR NOT NULL, date DATETIME NOT NULL, status VARCHAR NOT NULL, name VARCHAR NOT NULL,
    df_identity VARCHAR, df_cov_query VARCHAR, df_aln_length VARCHAR, df_sim_errors VARCHAR, df_hadamard VARCHAR,
    CONSTRAINT pk_runs PRIMARY KEY (run_id),
    CONSTRAINT fk_runs_configuration_id_configurations FOREIGN KEY(configuration_id)
        REFERENCES configurations (configuration_id));
CREATE TABLE IF NOT EXISTS runs_genomes (
    genome_hash VARCHAR NOT NULL, run_id INTEGER NOT NULL, fasta_filename VARCHAR NOT NULL,
    CONSTRAINT pk_runs_genomes PRIMARY KEY (genome_hash, run_id),
    CONSTRAINT fk_runs_genomes_genome_hash_genomes FOREIGN KEY(genome_hash) REFERENCES genomes (genome_hash),
    CONSTRAINT fk_runs_genomes_run_id_runs FOREIGN KEY(run_id) REFERENCES runs (run_id));
"""


# ------------------------------------------------------------------ plain-object mirrors of the ORM rows
@dataclass
class Configuration:
    configuration_id: int
    method: str
    program: str
    version: str
    fragsize: int | None = None
    mode: str | None = None
    kmersize: int | None = None
    minmatch: float | None = None
    extra: str | None = None


@dataclass
class RunGenomeAssociation:
    genome_hash: str
    fasta_filename: str


@dataclass
class Run:
    """Duck-type of ``db_orm.Run`` as far as the method module reads it."""

    run_id: int
    configuration: Configuration
    fasta_directory: str
    fasta_hashes: list[RunGenomeAssociation]
    status: str
    name: str = ""

    @property
    def configuration_id(self) -> int:
        return self.configuration.configuration_id


class Session:
    """Minimal session: ``commit()`` persists ``run.status`` (what the worker's interrupt path needs)."""

    def __init__(self, conn: sqlite3.Connection, run: Run | None = None):
        self.conn = conn
        self.run = run

    def commit(self) -> None:
        if self.run is not None:
            self.conn.execute("UPDATE runs SET status=? WHERE run_id=?", (self.run.status, self.run.run_id))
        self.conn.commit()


# ------------------------------------------------------------------ FASTA bookkeeping
def check_fasta(logger: logging.Logger, fasta: Path) -> list[Path]:
    """FASTA files of a directory by extension (pyani_plus/utils.py:226-242)."""
    fasta = Path(fasta)
    if not fasta.is_dir():
        sourmash_hip.log_sys_exit(logger, f"FASTA input {fasta} is not a directory")
    names: list[Path] = []
    for ext in sorted(FASTA_EXTENSIONS):
        names.extend(fasta.glob("*" + ext))
        names.extend(fasta.glob("*" + ext + ".gz"))
    if not names:
        sourmash_hip.log_sys_exit(
            logger, f"No FASTA input genomes under {fasta} with extensions {', '.join(sorted(FASTA_EXTENSIONS))}"
        )
    return sorted(names)


def fasta_length_and_description(text: bytes) -> tuple[int, str | None]:
    """Sum of residues and first title, as fasta_bytes_iterator sees them (utils.py:67-90)."""
    length = 0
    description = None
    in_record = False
    for line in text.split(b"\n"):
        if line[:1] == b">":
            in_record = True
            if description is None:
                description = line[1:].rstrip().decode()
            continue
        if in_record:
            length += len(line.translate(None, b" \t\r\n"))
    return length, description


# ------------------------------------------------------------------ database
def connect_to_db(database: Path | str) -> sqlite3.Connection:
    conn = sqlite3.connect(str(database), timeout=30.0)
    conn.executescript(SCHEMA)
    conn.commit()
    return conn


def db_configuration(conn, method, program, version, fragsize=None, mode=None, kmersize=None, minmatch=None,
                     extra=None) -> Configuration:
    """Return the matching configuration row, creating it if needed (db_orm.py:705-782)."""
    row = conn.execute(
        "SELECT configuration_id FROM configurations WHERE method=? AND program=? AND version=? AND fragsize IS ? "
        "AND mode IS ? AND kmersize IS ? AND minmatch IS ? AND extra IS ?",
        (method, program, version, fragsize, mode, kmersize, minmatch, extra),
    ).fetchone()
    if row is None:
        cur = conn.execute(
            "INSERT INTO configurations (method, program, version, fragsize, mode, kmersize, minmatch, extra) "
            "VALUES (?,?,?,?,?,?,?,?)",
            (method, program, version, fragsize, mode, kmersize, minmatch, extra),
        )
        conn.commit()
        cid = cur.lastrowid
    else:
        cid = row[0]
    return Configuration(cid, method, program, version, fragsize, mode, kmersize, minmatch, extra)


def db_genome(conn, path: Path, md5: str, length: int, description: str) -> None:
    conn.execute(
        "INSERT OR IGNORE INTO genomes (genome_hash, path, length, description) VALUES (?,?,?,?)",
        (md5, str(path), length, description),
    )


def add_run(conn, config: Configuration, cmdline: str, fasta_directory: Path, status: str, name: str,
            fasta_to_hash: dict[Path, str]) -> Run:
    now = datetime.datetime.now(datetime.timezone.utc).replace(tzinfo=None).isoformat(sep=" ")
    cur = conn.execute(
        "INSERT INTO runs (configuration_id, cmdline, fasta_directory, date, status, name) VALUES (?,?,?,?,?,?)",
        (config.configuration_id, cmdline, str(fasta_directory), now, status, name),
    )
    run_id = cur.lastrowid
    assoc = []
    for filename, md5 in fasta_to_hash.items():
        conn.execute(
            "INSERT INTO runs_genomes (genome_hash, run_id, fasta_filename) VALUES (?,?,?)",
            (md5, run_id, Path(filename).name),
        )
        assoc.append(RunGenomeAssociation(md5, Path(filename).name))
    conn.commit()
    return Run(run_id, config, str(fasta_directory), assoc, status, name)


def load_run(conn, run_id: int) -> Run:
    row = conn.execute(
        "SELECT configuration_id, fasta_directory, status, name FROM runs WHERE run_id=?", (run_id,)
    ).fetchone()
    if row is None:
        msg = f"Database has no run {run_id}"
        raise ValueError(msg)
    crow = conn.execute(
        "SELECT configuration_id, method, program, version, fragsize, mode, kmersize, minmatch, extra "
        "FROM configurations WHERE configuration_id=?",
        (row[0],),
    ).fetchone()
    assoc = [
        RunGenomeAssociation(h, f)
        for h, f in conn.execute("SELECT genome_hash, fasta_filename FROM runs_genomes WHERE run_id=?", (run_id,))
    ]
    return Run(run_id, Configuration(*crow), row[1], assoc, row[2], row[3])


def count_run_comparisons(conn, run: Run) -> int:
    """Comparisons among this run's genomes under its configuration (Run.comparisons(), db_orm.py:353-391)."""
    return conn.execute(
        "SELECT COUNT(*) FROM comparisons c "
        "JOIN runs_genomes q ON c.query_hash = q.genome_hash AND q.run_id = ? "
        "JOIN runs_genomes s ON c.subject_hash = s.genome_hash AND s.run_id = ? "
        "WHERE c.configuration_id = ?",
        (run.run_id, run.run_id, run.configuration_id),
    ).fetchone()[0]


def import_json_comparisons(logger: logging.Logger, conn, json_filename: Path) -> int:
    """Import one column file; the configuration must already exist; ``cov_subject`` is ignored
    (pyani_plus/private_cli.py:507-614)."""
    data = wire.load_json_comparisons(json_filename)
    cfg = data["configuration"]
    row = conn.execute(
        "SELECT configuration_id FROM configurations WHERE method=? AND program=? AND version=? AND fragsize IS ? "
        "AND mode IS ? AND kmersize IS ? AND minmatch IS ? AND extra IS ?",
        tuple(cfg[k] for k in wire.CONFIG_FIELDS),
    ).fetchone()
    if row is None:
        sourmash_hip.log_sys_exit(logger, f"JSON file {json_filename} configuration not in database")
    cid = row[0]
    uname = data["uname"]
    rows = [
        (
            e["query_hash"], e["subject_hash"], cid, e["identity"], e.get("aln_length"), e.get("sim_errors"),
            e.get("cov_query"), uname["system"], uname["release"], uname["machine"],
        )  # fmt: skip
        for e in data["comparisons"]
    ]
    conn.executemany(
        "INSERT OR IGNORE INTO comparisons (query_hash, subject_hash, configuration_id, identity, aln_length, "
        "sim_errors, cov_query, uname_system, uname_release, uname_machine) VALUES (?,?,?,?,?,?,?,?,?,?)",
        rows,
    )
    conn.commit()
    return len(rows)


INSERT_COMPARISON = (
    "INSERT OR IGNORE INTO comparisons (query_hash, subject_hash, configuration_id, identity, aln_length, "
    "sim_errors, cov_query, uname_system, uname_release, uname_machine) VALUES (?,?,?,?,?,?,?,?,?,?)"
)


def _database_file(conn) -> str | None:
    """Path of the connection's main database, or None for an in-memory / temporary one."""
    for _seq, name, path in conn.execute("PRAGMA database_list"):
        if name == "main":
            return path or None
    return None


def ingest_matrices_native(conn, run: Run, queries: list[str], subjects: list[str], identity, cov_query, is_null) -> int | None:
    """``ingest_matrices`` through ``pa_sqlite_insert_comparisons`` (one prepared statement stepped from C on a
    connection of its own).  Returns the number of comparisons handled, or None when the native route does not
    apply (in-memory database, libsqlite3.so.0 not loadable) -- the caller then uses Python's sqlite3 module."""
    import ctypes as C
    import platform

    from . import _capi

    path = _database_file(conn)
    if path is None:
        return None
    nq, ns = len(queries), len(subjects)
    if nq == 0 or ns == 0:
        return 0
    uname = platform.uname()
    identity = np.ascontiguousarray(identity, dtype=np.float64)
    cov_query = np.ascontiguousarray(cov_query, dtype=np.float64)
    null = np.ascontiguousarray(is_null, dtype=np.uint8)
    assert identity.shape == (nq, ns) == cov_query.shape == null.shape
    q_arr = (C.c_char_p * nq)(*[q.encode() for q in queries])
    s_arr = (C.c_char_p * ns)(*[s.encode() for s in subjects])
    conn.commit()  # the call opens its own connection: nothing of ours may hold the write lock
    inserted = C.c_uint64()
    status = _capi.load_library().pa_sqlite_insert_comparisons(
        path.encode(), run.configuration_id, uname.system.encode(), uname.release.encode(), uname.machine.encode(),
        q_arr, nq, s_arr, ns, identity.ctypes.data, cov_query.ctypes.data, null.ctypes.data, C.byref(inserted),
    )  # fmt: skip
    if status == _capi.PA_E_IO and "libsqlite3" in _capi.last_error():
        return None
    _capi.check(status, "pa_sqlite_insert_comparisons")
    return nq * ns


def ingest_matrices(conn, run: Run, queries: list[str], subjects: list[str], identity, cov_query, is_null, *,
                    chunk_rows: int = 1_000_000, native: bool = True) -> int:
    """Comparison rows straight from the result matrices (the reference goes through one
    Python dict per row, a JSON file and its re-parse: pyani_plus/private_cli.py:1863-1888, 507-614).

    Rows go in query-major with ascending subjects -- the order of the UNIQUE(query_hash, subject_hash,
    configuration_id) index when both lists are sorted, so the index grows by appends.  With ``native`` the rows
    are stepped from C (``ingest_matrices_native``); otherwise, or when that route does not apply, in chunks of
    ``chunk_rows`` through one ``executemany`` each."""
    import platform

    if native:
        done = ingest_matrices_native(conn, run, queries, subjects, identity, cov_query, is_null)
        if done is not None:
            return done
    uname = platform.uname()
    cid = run.configuration_id
    nq, ns = len(queries), len(subjects)
    identity = np.asarray(identity, dtype=np.float64)
    cov_query = np.asarray(cov_query, dtype=np.float64)
    null = np.asarray(is_null, dtype=bool)
    rows_per_chunk = max(1, chunk_rows // max(ns, 1))
    constants = (uname.system, uname.release, uname.machine)
    for q0 in range(0, nq, rows_per_chunk):
        q1 = min(nq, q0 + rows_per_chunk)
        ident = identity[q0:q1].astype(object)
        cov = cov_query[q0:q1].astype(object)
        ident[null[q0:q1]] = None
        cov[null[q0:q1]] = None
        rows = (
            (q, s, cid, i, None, None, c, *constants)
            for q, irow, crow in zip(queries[q0:q1], ident, cov)
            for s, i, c in zip(subjects, irow, crow)
        )
        conn.executemany(INSERT_COMPARISON, rows)
    conn.commit()
    return nq * ns


def format_matrix_cache(hashes: list[str], identity, cov_query, is_null) -> dict[str, str] | None:
    """The five ``runs.df_*`` strings from matrices in memory (rows = query, columns = subject, both in ``hashes``
    order = sorted md5); None when they would not fit a SQLite value."""
    import pandas as pd

    assert hashes == sorted(hashes)
    n = len(hashes)
    if _matrix_cache_too_big(n):
        return None
    ident = np.where(is_null, np.nan, identity)
    cov = np.where(is_null, np.nan, cov_query)
    nan = np.full((n, n), np.nan)
    mats = {"identity": ident, "cov_query": cov, "aln_length": nan, "sim_errors": nan, "hadamard": ident * cov}
    return {
        f"df_{key}": pd.DataFrame(data=mat, index=hashes, columns=hashes, dtype=float).to_json(orient="split")
        for key, mat in mats.items()
    }


def cache_matrices(conn, run: Run, hashes: list[str], identity, cov_query, is_null, *, formatted=None) -> dict[str, str]:
    """``cache_comparisons`` from matrices that are already in memory: same strings as the SELECT-based form, without
    reading 10^8 rows back.  ``formatted`` = the result of an earlier ``format_matrix_cache`` of the same matrices."""
    out = formatted if formatted is not None else format_matrix_cache(hashes, identity, cov_query, is_null)
    _store_matrix_cache(conn, run, out)
    return out or {}


def _matrix_cache_too_big(n: int) -> bool:
    """A cached matrix is about 13 characters per cell ("0.9997081124,"): beyond ~7.5e7 cells its JSON text passes
    SQLite's 10^9-byte value limit, so formatting it would be wasted work."""
    return n * n * 13 > 950_000_000


def _store_matrix_cache(conn, run: Run, out: dict[str, str] | None) -> bool:
    """``runs.df_*`` hold the matrices as JSON text (db_orm.py:442-465).  SQLite refuses a value of more than
    10^9 bytes (SQLITE_MAX_LENGTH), which a 10 000 x 10 000 matrix exceeds (about 1.3 GB of text) -- in the
    reference just as here.  The comparisons table is complete either way; the cache columns then stay NULL, which
    the reference treats as "not cached yet" (db_orm.py:393-405)."""
    try:
        if out is None:
            raise sqlite3.DataError("not attempted")
        conn.execute(
            "UPDATE runs SET df_identity=?, df_cov_query=?, df_aln_length=?, df_sim_errors=?, df_hadamard=? WHERE run_id=?",
            (out["df_identity"], out["df_cov_query"], out["df_aln_length"], out["df_sim_errors"], out["df_hadamard"],
             run.run_id),
        )
    except (sqlite3.DataError, OverflowError) as err:
        logging.getLogger("pyani_plus_amd").warning(
            "matrix cache of run %d not stored (%s): %d genomes give JSON strings beyond SQLite's 10^9-byte limit",
            run.run_id, err, len(run.fasta_hashes),
        )
        conn.rollback()
        return False
    conn.commit()
    return True


def cache_comparisons(conn, run: Run) -> dict[str, str]:
    """Fill runs.df_* with the N x N matrices (rows = query, columns = subject, sorted md5)."""
    import pandas as pd

    hashes = sorted(a.genome_hash for a in run.fasta_hashes)
    index = {h: i for i, h in enumerate(hashes)}
    n = len(hashes)
    if _matrix_cache_too_big(n):
        _store_matrix_cache(conn, run, None)
        return {}
    mats = {k: np.full((n, n), np.nan, float) for k in ("identity", "cov_query", "aln_length", "sim_errors")}
    rows = conn.execute(
        "SELECT c.query_hash, c.subject_hash, c.identity, c.cov_query, c.aln_length, c.sim_errors FROM comparisons c "
        "JOIN runs_genomes rq ON c.query_hash = rq.genome_hash AND rq.run_id = ? "
        "JOIN runs_genomes rs ON c.subject_hash = rs.genome_hash AND rs.run_id = ? WHERE c.configuration_id = ?",
        (run.run_id, run.run_id, run.configuration_id),
    ).fetchall()
    if rows:
        # one dictionary lookup per row (the reference does a list.index per row: O(N^3) overall)
        r_idx = np.fromiter((index[r[0]] for r in rows), dtype=np.int64, count=len(rows))
        c_idx = np.fromiter((index[r[1]] for r in rows), dtype=np.int64, count=len(rows))
        for col, key in enumerate(("identity", "cov_query", "aln_length", "sim_errors"), start=2):
            mats[key][r_idx, c_idx] = np.array([r[col] for r in rows], dtype=float)  # None -> NaN
    mats["hadamard"] = mats["identity"] * mats["cov_query"]
    out = {
        f"df_{key}": pd.DataFrame(data=mat, index=hashes, columns=hashes, dtype=float).to_json(orient="split")
        for key, mat in mats.items()
    }
    _store_matrix_cache(conn, run, out)
    return out


def _compute_direct(logger, conn, run: Run, cache_dir: Path, tmp_dir: Path, engine, mark):
    """Subject tiles -> binary column files + matrices in host memory -> rows inserted in index order."""
    config = run.configuration
    hashes = sorted(a.genome_hash for a in run.fasta_hashes)
    n = len(hashes)
    ident = np.empty((n, n), dtype=np.float64)
    cov = np.empty((n, n), dtype=np.float64)
    null = np.empty((n, n), dtype=bool)
    sig_cache = sourmash_hip.sig_cache_dir(cache_dir, config.kmersize, config.extra)
    col = 0
    try:
        for t, (queries, tile, t_cov, t_ident, t_null) in enumerate(
            sourmash_hip.iter_sourmash_tiles(
                logger, hashes, hashes, sig_cache, kmersize=config.kmersize, scaled=sourmash_hip.parse_scaled(config.extra), engine=engine
            )
        ):
            assert queries == hashes and tile == hashes[col : col + len(tile)]
            wire.save_tile(tmp_dir / f"{sourmash_hip.METHOD}.run_{run.run_id}.tile_{t}.npz", config, queries, tile, t_ident, t_cov, t_null)
            ident[:, col : col + len(tile)] = t_ident
            cov[:, col : col + len(tile)] = t_cov
            null[:, col : col + len(tile)] = t_null
            col += len(tile)
    except HipBackendError as err:
        sourmash_hip.backend_failure(logger, f"{sourmash_hip.METHOD} comparison", err)
    mark("pairs_and_tile_files")
    conn.execute("PRAGMA synchronous=OFF")
    conn.execute("PRAGMA cache_size=-1048576")
    # the cached matrices are formatted on a second thread while the rows go in (the native insert releases the GIL)
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=1) as side:
        formatting = side.submit(format_matrix_cache, hashes, ident, cov, null)
        rows = ingest_matrices(conn, run, hashes, hashes, ident, cov, null)
        formatted = formatting.result()
    conn.execute("PRAGMA synchronous=FULL")
    mark("insert_rows")
    return rows, hashes, ident, cov, null, formatted


def import_tile(logger: logging.Logger, conn, run: Run, tile_file: Path) -> int:
    """Import one binary column file written by ``wire.save_tile`` (resuming a direct-ingest run)."""
    config, queries, subjects, ident, cov, null = wire.load_tile(tile_file)
    for key in wire.CONFIG_FIELDS:
        if config[key] != getattr(run.configuration, key):
            sourmash_hip.log_sys_exit(logger, f"Tile file {tile_file} configuration does not match the run ({key})")
    return ingest_matrices(conn, run, queries, subjects, ident, cov, null)


# ------------------------------------------------------------------ the run itself
def run_sourmash_hip(  # noqa: PLR0913
    fasta: Path,
    database: Path | str,
    *,
    cache: Path | None = None,
    name: str | None = None,
    kmersize: int = sourmash_hip.KMER_SIZE,
    scaled: int = sourmash_hip.SCALED,
    temp: Path | None = None,
    logger: logging.Logger | None = None,
    engine=None,
    ingest: str = "json",
    timings: dict | None = None,
) -> Run:
    """FASTA directory -> database with all N^2 comparisons and cached matrices.

    Counterpart of ``pyani-plus sourmash <fasta> -d <db> --create-db`` (call stack in
    SURVEY.md section 3.1) with the snakemake layer replaced by one in-process call.

    ``ingest="json"`` goes through the reference's column file (worker -> JSON -> importer), what two
    processes of the reference would do.  ``ingest="direct"`` keeps the subject tiles as binary column
    files (``wire.save_tile``) plus in-memory matrices, inserts the rows in index order straight from them and
    writes the matrix cache from memory: the form that stays feasible at N = 10^4 (10^8 rows).
    ``timings`` (a dict) receives the wall seconds of the phases."""
    import time

    clock = time.perf_counter
    marks = {"start": clock()}

    def mark(name: str) -> None:
        marks[name] = clock()
        if timings is not None:
            prev = list(marks)[-2]
            timings[name] = marks[name] - marks[prev]

    logger = logger or logging.getLogger("pyani_plus_amd")
    fasta = Path(fasta)
    if not 1 <= int(kmersize) <= 64:  # before any file is read
        sourmash_hip.log_sys_exit(logger, f"{sourmash_hip.METHOD} supports k-mer sizes 1 to 64, not {kmersize}")
    if int(scaled) < 1:
        sourmash_hip.log_sys_exit(logger, f"scaled must be a positive integer, not {scaled}")
    if ingest not in {"json", "direct"}:
        sourmash_hip.log_sys_exit(logger, f"ingest must be 'json' or 'direct', not {ingest!r}")
    fasta_names = check_fasta(logger, fasta)
    tool = sourmash_hip.get_sourmash_hip()
    conn = connect_to_db(database)
    config = db_configuration(
        conn, sourmash_hip.METHOD, tool.exe_path.stem, tool.version, kmersize=kmersize, extra=f"scaled={scaled}"
    )
    filename_to_md5: dict[Path, str] = {}
    seen: set[str] = set()
    # One pass over the files: md5 of the decompressed bytes, length, first title AND the sketches -- the host
    # front-end of the next batch of files runs while the device hashes the current one (sketch_fasta_batches).
    presketched: dict[str, np.ndarray] = {}
    own_cache = cache is None
    cache_dir = Path(tempfile.mkdtemp(prefix="pyani_hip_cache_")) if own_cache else Path(cache)
    cache_dir.mkdir(parents=True, exist_ok=True)
    sig_dir = sourmash_hip.sig_cache_dir(cache_dir, kmersize, f"scaled={scaled}")
    try:
        for batch_paths, infos, sketches in sourmash_hip.sketch_fasta_batches(
            logger, fasta_names, kmersize=kmersize, scaled=scaled, engine=engine, needed=lambda info: not (sig_dir / f"{info.md5}.sig").is_file()
        ):
            for filename, info, mins in zip(batch_paths, infos, sketches):
                md5 = info.md5
                if md5 in seen:
                    dups = "\n" + "\n".join(sorted({str(k) for k, v in filename_to_md5.items() if v == md5} | {str(filename)}))
                    sourmash_hip.log_sys_exit(logger, f"Multiple genomes with same MD5 checksum {md5}:{dups}")
                seen.add(md5)
                filename_to_md5[filename] = md5
                if mins is not None:
                    presketched[md5] = mins
                db_genome(conn, filename, md5, info.length, info.description)
    except HipBackendError as err:
        sourmash_hip.backend_failure(logger, f"{sourmash_hip.METHOD} sketching", err)
    mark("fasta_front_end_and_sketch")
    run = add_run(
        conn, config, " ".join(sys.argv), fasta, "Initialising",
        f"{len(filename_to_md5)} genomes using {sourmash_hip.METHOD}" if name is None else name, filename_to_md5,
    )  # fmt: skip
    session = Session(conn, run)
    n = len(filename_to_md5)
    direct = None
    if count_run_comparisons(conn, run) == n * n:
        logger.info("Database already has all %d=%d^2 comparisons", n * n, n)
    else:
        run.status = "Running"
        session.commit()
        # the genomes were sketched while their checksums were taken: only the signature files remain to be written
        for _ in sourmash_hip.prepare_genomes(logger, run, cache_dir, engine=engine, presketched=presketched):
            pass
        mark("signature_files")
        tmp_dir = Path(temp) if temp else Path(tempfile.mkdtemp(prefix="pyani_hip_"))
        hash_to_filename = {a.genome_hash: a.fasta_filename for a in run.fasta_hashes}
        if ingest == "direct":
            direct = _compute_direct(logger, conn, run, cache_dir, tmp_dir, engine, mark)
        else:
            json_file = tmp_dir / f"{sourmash_hip.METHOD}.run_{run.run_id}.column_0.json"
            lengths = dict(conn.execute("SELECT genome_hash, length FROM genomes"))
            status = sourmash_hip.compute_sourmash_hip(
                logger, tmp_dir, session, run, json_file, fasta, hash_to_filename,
                {v: k for k, v in hash_to_filename.items()}, {h: lengths[h] for h in hash_to_filename}, "",
                cache=cache_dir, engine=engine,
            )  # fmt: skip
            if status:
                sourmash_hip.log_sys_exit(logger, f"Column worker failed with return code {status}")
            mark("pairs_and_column_file")
            import_json_comparisons(logger, conn, json_file)
            mark("import_column_file")
    done = count_run_comparisons(conn, run) if direct is None else direct[0]
    if done != n * n:
        sourmash_hip.log_sys_exit(logger, f"Only have {done} of {n}^2={n * n} {sourmash_hip.METHOD} comparisons needed")
    if direct is None:
        cache_comparisons(conn, run)
    else:
        if direct[5] is None:
            _store_matrix_cache(conn, run, None)
        else:
            cache_matrices(conn, run, *direct[1:5], formatted=direct[5])
    mark("matrix_cache")
    run.status = "Done"
    session.commit()
    conn.close()
    return run
