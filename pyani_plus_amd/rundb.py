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
import os
import sqlite3
import zipfile
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


def ingest_matrices_native(conn, run: Run, queries: list[str], subjects: list[str], identity, cov_query, is_null, *,
                           aln_length=None, sim_errors=None) -> int | None:
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
    aln = err = None
    if aln_length is not None:
        aln = np.ascontiguousarray(aln_length, dtype=np.int64)
        err = np.ascontiguousarray(sim_errors, dtype=np.int64)
        assert aln.shape == (nq, ns) == err.shape
    status = _capi.load_library().pa_sqlite_insert_comparisons_ex(
        path.encode(), run.configuration_id, uname.system.encode(), uname.release.encode(), uname.machine.encode(),
        q_arr, nq, s_arr, ns, identity.ctypes.data, cov_query.ctypes.data, null.ctypes.data,
        None if aln is None else aln.ctypes.data, None if err is None else err.ctypes.data, C.byref(inserted),
    )  # fmt: skip
    if status == _capi.PA_E_IO and "libsqlite3" in _capi.last_error():
        return None
    _capi.check(status, "pa_sqlite_insert_comparisons")
    return nq * ns


def ingest_matrices(conn, run: Run, queries: list[str], subjects: list[str], identity, cov_query, is_null, *,
                    chunk_rows: int = 1_000_000, native: bool = True, aln_length=None, sim_errors=None) -> int:
    """Comparison rows straight from the result matrices (the reference goes through one
    Python dict per row, a JSON file and its re-parse: pyani_plus/private_cli.py:1863-1888, 507-614).

    Rows go in query-major with ascending subjects -- the order of the UNIQUE(query_hash, subject_hash,
    configuration_id) index when both lists are sorted, so the index grows by appends.  With ``native`` the rows
    are stepped from C (``ingest_matrices_native``); otherwise, or when that route does not apply, in chunks of
    ``chunk_rows`` through one ``executemany`` each."""
    import platform

    if native:
        done = ingest_matrices_native(conn, run, queries, subjects, identity, cov_query, is_null, aln_length=aln_length, sim_errors=sim_errors)
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
        if aln_length is None:
            aln = err = np.full(ident.shape, None, dtype=object)
        else:
            aln = np.asarray(aln_length)[q0:q1].astype(np.int64).astype(object)  # Python ints: sqlite3 stores numpy scalars as blobs
            err = np.asarray(sim_errors)[q0:q1].astype(np.int64).astype(object)
            aln[null[q0:q1]] = None
            err[null[q0:q1]] = None
        rows = (
            (q, s, cid, i, a, e, c, *constants)
            for q, irow, crow, arow, erow in zip(queries[q0:q1], ident, cov, aln, err)
            for s, i, c, a, e in zip(subjects, irow, crow, arow, erow)
        )
        conn.executemany(INSERT_COMPARISON, rows)
    conn.commit()
    return nq * ns


def format_matrix_cache(hashes: list[str], identity, cov_query, is_null, *, aln_length=None, sim_errors=None) -> dict[str, str] | None:
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
    aln = nan if aln_length is None else np.where(is_null, np.nan, np.asarray(aln_length, dtype=np.float64))
    err = nan if sim_errors is None else np.where(is_null, np.nan, np.asarray(sim_errors, dtype=np.float64))
    mats = {"identity": ident, "cov_query": cov, "aln_length": aln, "sim_errors": err, "hadamard": ident * cov}
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


def _compute_direct(logger, conn, run: Run, cache_dir: Path, tmp_dir: Path, engine, mark, *, subjects: list[str] | None = None):
    """Subject tiles -> binary column files + matrices in host memory -> rows inserted in index order.
    ``subjects``: only these subject columns (resuming a partial run); the matrices then hold those columns."""
    config = run.configuration
    hashes = sorted(a.genome_hash for a in run.fasta_hashes)
    cols = hashes if subjects is None else sorted(subjects)
    n, nc = len(hashes), len(cols)
    ident = np.empty((n, nc), dtype=np.float64)
    cov = np.empty((n, nc), dtype=np.float64)
    null = np.empty((n, nc), dtype=bool)
    sig_cache = sourmash_hip.sig_cache_dir(cache_dir, config.kmersize, config.extra)
    col = 0
    try:
        for t, (queries, tile, t_cov, t_ident, t_null) in enumerate(
            sourmash_hip.iter_sourmash_tiles(
                logger, cols, hashes, sig_cache, kmersize=config.kmersize, scaled=sourmash_hip.parse_scaled(config.extra), engine=engine
            )
        ):
            assert queries == hashes and tile == cols[col : col + len(tile)]
            wire.save_tile(tmp_dir / f"{sourmash_hip.METHOD}.run_{run.run_id}.tile_{t}.npz", config, queries, tile, t_ident, t_cov, t_null)
            ident[:, col : col + len(tile)] = t_ident
            cov[:, col : col + len(tile)] = t_cov
            null[:, col : col + len(tile)] = t_null
            col += len(tile)
    except HipBackendError as err:
        sourmash_hip.backend_failure(logger, f"{sourmash_hip.METHOD} comparison", err)
    mark("pairs_and_tile_files")
    return _ingest_direct(conn, run, hashes, cols, ident, cov, null, mark)


def _ingest_direct(conn, run: Run, hashes: list[str], cols: list[str], ident, cov, null, mark, *, aln_length=None, sim_errors=None):
    """Matrices in host memory (rows = ``hashes``, columns = ``cols``, both sorted) -> comparison rows in index order,
    the five cached matrices formatted on a second thread meanwhile (when the block is the whole square)."""
    # synchronous=NORMAL for the bulk insert: a handful of fsyncs per transaction instead of one per page group, and
    # -- unlike OFF -- no way for a crash of the machine to corrupt the user's multi-run database
    conn.execute("PRAGMA synchronous=NORMAL")
    conn.execute("PRAGMA cache_size=-1048576")
    from concurrent.futures import ThreadPoolExecutor

    square = cols == hashes
    with ThreadPoolExecutor(max_workers=1) as side:
        formatting = side.submit(format_matrix_cache, hashes, ident, cov, null, aln_length=aln_length, sim_errors=sim_errors) if square else None
        ingest_matrices(conn, run, hashes, cols, ident, cov, null, aln_length=aln_length, sim_errors=sim_errors)
        formatted = formatting.result() if formatting is not None else None
    conn.execute("PRAGMA synchronous=FULL")
    mark("insert_rows")
    # what is in the database now, not what was handed to the insert (INSERT OR IGNORE reports nothing per row)
    rows = count_run_comparisons(conn, run)
    return rows, hashes, ident, cov, null, formatted if square else False


def import_tile(logger: logging.Logger, conn, run: Run, tile_file: Path) -> int:
    """Import one binary column file written by ``wire.save_tile`` (resuming a direct-ingest run)."""
    config, queries, subjects, ident, cov, null = wire.load_tile(tile_file)
    for key in wire.CONFIG_FIELDS:
        if config[key] != getattr(run.configuration, key):
            sourmash_hip.log_sys_exit(logger, f"Tile file {tile_file} configuration does not match the run ({key})")
    return ingest_matrices(conn, run, queries, subjects, ident, cov, null)


# ------------------------------------------------------------------ the run itself
def _phase_clock(timings: dict | None):
    import time

    marks = {"start": time.perf_counter()}

    def mark(name: str) -> None:
        marks[name] = time.perf_counter()
        if timings is not None:
            prev = list(marks)[-2]
            timings[name] = marks[name] - marks[prev]

    return mark


def _duplicate_md5_exit(logger, md5: str, filenames) -> None:
    """Two input files with the same content (pyani_plus/public_cli.py:165-171)."""
    dups = "\n" + "\n".join(sorted({str(f) for f in filenames}))
    sourmash_hip.log_sys_exit(logger, f"Multiple genomes with same MD5 checksum {md5}:{dups}")


def _file_cost(path: Path) -> int:
    """Bases a FASTA file is expected to hold, from its size (gzip: about a quarter of the text)."""
    try:
        size = path.stat().st_size
    except OSError:
        return 0
    return 4 * size if path.name.endswith(".gz") else size


def _sharded_sourmash_tiles(logger, fasta: Path, fasta_names: list[Path], config: Configuration, cache_dir: Path, tmp_dir: Path,
                            gpus: int, engine_factory: str | None, columns: list[str] | None = None):
    """The multi-GPU form of "sketch everything, compare everything" (DESIGN.md section 6): ``gpus`` fresh worker
    processes (``launch.launch_workers`` -- started before this process has touched a GPU), each sketching a
    length-balanced share of the files, ONE all-gather of the sketches, each rank evaluating all queries against its
    own genomes as subject columns (``columns``, checksums: only those of them -- what a resumed run still needs).
    Returns (metadata per file in file order, the ranks' tile files, the ranks' reports); None when the ranks were
    interrupted (their columns need every rank's sketches: there is nothing partial to keep)."""
    from . import launch
    from .distributed import shard_bounds_by_cost

    shards = shard_bounds_by_cost([max(1, _file_cost(p)) for p in fasta_names], gpus)
    work_dir = tmp_dir / f"{sourmash_hip.METHOD}.workers"
    spec = {
        "task": "sourmash", "fasta_dir": str(fasta), "fasta_files": [str(p) for p in fasta_names], "shards": shards,
        "configuration": {k: getattr(config, k) for k in wire.CONFIG_FIELDS}, "cache": str(cache_dir), "work_dir": str(work_dir),
    }  # fmt: skip
    if engine_factory:
        spec["engine_factory"] = engine_factory
    if columns is not None:
        spec["columns"] = list(columns)
    try:
        results = launch.launch_workers(gpus, spec, work_dir)
    except launch.WorkerFailure as err:
        sourmash_hip.log_sys_exit(logger, str(err))
    if any(r.get("interrupted") for r in results):
        return None
    meta = [m for r in results for m in r["meta"]]
    assert [m["path"] for m in meta] == [str(p) for p in fasta_names]
    return meta, [Path(r["tile"]) for r in results if r.get("tile")], results


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
    gpus: int = 1,
    engine_factory: str | None = None,
) -> Run:
    """FASTA directory -> database with all N^2 comparisons and cached matrices.

    Counterpart of ``pyani-plus sourmash <fasta> -d <db> --create-db`` (call stack in
    SURVEY.md section 3.1) with the snakemake layer replaced by one in-process call.

    ``ingest="json"`` goes through the reference's column file (worker -> JSON -> importer), what two
    processes of the reference would do.  ``ingest="direct"`` keeps the subject tiles as binary column
    files (``wire.save_tile``) plus in-memory matrices, inserts the rows in index order straight from them and
    writes the matrix cache from memory: the form that stays feasible at N = 10^4 (10^8 rows).
    ``gpus`` > 1: the sketching and the comparisons are spread over that many worker processes, one per GPU
    (``_sharded_sourmash_tiles``; the caller must not have initialised the GPU in this process); results always take
    the direct route.  ``timings`` (a dict) receives the wall seconds of the phases."""
    mark = _phase_clock(timings)
    logger = logger or logging.getLogger("pyani_plus_amd")
    fasta = Path(fasta)
    if not 1 <= int(kmersize) <= 64:  # before any file is read
        sourmash_hip.log_sys_exit(logger, f"{sourmash_hip.METHOD} supports k-mer sizes 1 to 64, not {kmersize}")
    if int(scaled) < 1:
        sourmash_hip.log_sys_exit(logger, f"scaled must be a positive integer, not {scaled}")
    if ingest not in {"json", "direct"}:
        sourmash_hip.log_sys_exit(logger, f"ingest must be 'json' or 'direct', not {ingest!r}")
    if int(gpus) < 1:
        sourmash_hip.log_sys_exit(logger, f"gpus must be a positive integer, not {gpus}")
    fasta_names = check_fasta(logger, fasta)
    tool = sourmash_hip.get_sourmash_hip()
    conn = connect_to_db(database)
    config = db_configuration(
        conn, sourmash_hip.METHOD, tool.exe_path.stem, tool.version, kmersize=kmersize, extra=f"scaled={scaled}"
    )
    filename_to_md5: dict[Path, str] = {}
    seen: dict[str, Path] = {}
    own_cache = cache is None
    cache_dir = Path(tempfile.mkdtemp(prefix="pyani_hip_cache_")) if own_cache else Path(cache)
    cache_dir.mkdir(parents=True, exist_ok=True)
    tmp_dir = Path(temp) if temp else Path(tempfile.mkdtemp(prefix="pyani_hip_"))
    tmp_dir.mkdir(parents=True, exist_ok=True)
    sig_dir = sourmash_hip.sig_cache_dir(cache_dir, kmersize, f"scaled={scaled}")
    gpus = min(int(gpus), len(fasta_names))
    # PYANI_HIP_FORCE_WORKERS=1 sends even one GPU's worth of work through a worker process (RCCL with world size 1):
    # the multi-GPU code path on a single-GPU box
    if gpus > 1 or os.environ.get("PYANI_HIP_FORCE_WORKERS") == "1":
        sharded = _sharded_sourmash_tiles(logger, fasta, fasta_names, config, cache_dir, tmp_dir, gpus, engine_factory)
        if sharded is None:  # the genomes' checksums come from the ranks: no run has been recorded yet
            sourmash_hip.log_sys_exit(logger, "Interrupted before the sketches were exchanged; no run was recorded")
        meta, tile_files, _results = sharded
        for filename, m in zip(fasta_names, meta):
            if m["md5"] in seen:
                _duplicate_md5_exit(logger, m["md5"], [seen[m["md5"]], filename])
            seen[m["md5"]] = filename
            filename_to_md5[filename] = m["md5"]
            db_genome(conn, filename, m["md5"], m["length"], m["description"])
        mark("workers_front_end_sketch_and_pairs")
        run = add_run(
            conn, config, " ".join(sys.argv), fasta, "Running",
            f"{len(filename_to_md5)} genomes using {sourmash_hip.METHOD}" if name is None else name, filename_to_md5,
        )  # fmt: skip
        session = Session(conn, run)
        n = len(filename_to_md5)
        hashes = sorted(filename_to_md5.values())
        pos = {h: i for i, h in enumerate(hashes)}
        ident = np.empty((n, n), dtype=np.float64)
        cov = np.empty((n, n), dtype=np.float64)
        null = np.empty((n, n), dtype=bool)
        filled = 0
        for tile_file in tile_files:  # rows and columns arrive in the ranks' file order: place them by checksum
            t_config, queries, subjects, t_ident, t_cov, t_null = wire.load_tile(tile_file)
            for key in wire.CONFIG_FIELDS:
                if t_config[key] != getattr(config, key):
                    sourmash_hip.log_sys_exit(logger, f"Tile file {tile_file} configuration does not match the run ({key})")
            rows = np.array([pos[q] for q in queries])
            cols = np.array([pos[x] for x in subjects])
            ident[np.ix_(rows, cols)] = t_ident
            cov[np.ix_(rows, cols)] = t_cov
            null[np.ix_(rows, cols)] = t_null
            filled += len(cols)
        if filled != n:
            sourmash_hip.log_sys_exit(logger, f"The workers returned {filled} of {n} subject columns")
        mark("assemble_tiles")
        direct = _ingest_direct(conn, run, hashes, hashes, ident, cov, null, mark)
        return _finish_run(logger, conn, session, run, direct, mark)
    # One pass over the files: md5 of the decompressed bytes, length, first title AND the sketches -- the host
    # front-end of the next batch of files runs while the device hashes the current one (sketch_fasta_batches).
    presketched: dict[str, np.ndarray] = {}
    try:
        for batch_paths, infos, sketches in sourmash_hip.sketch_fasta_batches(
            logger, fasta_names, kmersize=kmersize, scaled=scaled, engine=engine, needed=lambda info: not (sig_dir / f"{info.md5}.sig").is_file()
        ):
            for filename, info, mins in zip(batch_paths, infos, sketches):
                md5 = info.md5
                if md5 in seen:
                    _duplicate_md5_exit(logger, md5, [k for k, v in filename_to_md5.items() if v == md5] + [filename])
                seen[md5] = filename
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
    direct = _compute_missing(logger, conn, session, run, cache_dir, tmp_dir, engine, ingest, mark, presketched=presketched)
    return _finish_run(logger, conn, session, run, direct, mark)


def _incomplete_columns(conn, run: Run) -> list[str]:
    """Subject genomes of the run with fewer than N comparisons recorded (the columns the reference's ``resume``
    recomputes, pyani_plus/public_cli.py:243-261)."""
    n = len(run.fasta_hashes)
    have = dict(
        conn.execute(
            "SELECT c.subject_hash, COUNT(*) FROM comparisons c "
            "JOIN runs_genomes q ON c.query_hash = q.genome_hash AND q.run_id = ? "
            "JOIN runs_genomes s ON c.subject_hash = s.genome_hash AND s.run_id = ? "
            "WHERE c.configuration_id = ? GROUP BY c.subject_hash",
            (run.run_id, run.run_id, run.configuration_id),
        )
    )
    return sorted(a.genome_hash for a in run.fasta_hashes if have.get(a.genome_hash, 0) < n)


def _compute_missing(logger, conn, session, run: Run, cache_dir: Path, tmp_dir: Path, engine, ingest: str, mark, *, presketched=None):
    """The comparisons the database does not hold yet, for a new run (all of them) or a resumed one (the incomplete
    subject columns only: rows that are there are never recomputed, nor -- INSERT OR IGNORE -- written twice)."""
    n = len(run.fasta_hashes)
    done = count_run_comparisons(conn, run)
    if done == n * n:
        logger.info("Database already has all %d=%d^2 %s comparisons", n * n, n, sourmash_hip.METHOD)
        return None
    logger.info("Database already has %d of %d^2=%d %s comparisons, %d needed", done, n, n * n, sourmash_hip.METHOD, n * n - done)
    columns = None if done == 0 else _incomplete_columns(conn, run)
    run.status = "Running"
    session.commit()
    # the genomes were sketched while their checksums were taken: only the signature files remain to be written
    for _ in sourmash_hip.prepare_genomes(logger, run, cache_dir, engine=engine, presketched=presketched):
        pass
    mark("signature_files")
    hash_to_filename = {a.genome_hash: a.fasta_filename for a in run.fasta_hashes}
    if ingest == "direct":
        return _compute_direct(logger, conn, run, cache_dir, tmp_dir, engine, mark, subjects=columns)
    lengths = dict(conn.execute("SELECT genome_hash, length FROM genomes"))
    for c, subject in enumerate([""] if columns is None else columns):
        json_file = tmp_dir / f"{sourmash_hip.METHOD}.run_{run.run_id}.column_{c if subject else 0}.json"
        status = sourmash_hip.compute_sourmash_hip(
            logger, tmp_dir, session, run, json_file, Path(run.fasta_directory), hash_to_filename,
            {v: k for k, v in hash_to_filename.items()}, {h: lengths[h] for h in hash_to_filename}, subject,
            cache=cache_dir, engine=engine,
        )  # fmt: skip
        if status:
            sourmash_hip.log_sys_exit(logger, f"Column worker failed with return code {status}")
        mark("pairs_and_column_file")
        import_json_comparisons(logger, conn, json_file)
        mark("import_column_file")
        if run.status == "Worker interrupted":
            break
    return None


def _finish_run(logger, conn, session, run: Run, direct, mark) -> Run:
    """Completion test, matrix cache, status "Done" (pyani_plus/public_cli.py:302-324)."""
    n = len(run.fasta_hashes)
    done = count_run_comparisons(conn, run) if direct is None else direct[0]
    if done != n * n and run.status == "Worker interrupted":
        # the reference's worker ends with return code 0 after an interrupt, its partial results recorded and the run
        # marked (pyani_plus/private_cli.py:1889-1902); ``resume`` completes such a run
        logger.warning("Interrupted: %d of %d^2=%d %s comparisons recorded; the run can be resumed", done, n, n * n, run.configuration.method)
        session.commit()
        conn.close()
        return run
    if done != n * n:
        sourmash_hip.log_sys_exit(logger, f"Only have {done} of {n}^2={n * n} {run.configuration.method} comparisons needed")
    if direct is None or direct[5] is False:  # JSON route, or a resumed run that only holds some columns in memory
        cache_comparisons(conn, run)
    elif direct[5] is None:
        _store_matrix_cache(conn, run, None)
    else:
        cache_matrices(conn, run, *direct[1:5], formatted=direct[5])
    mark("matrix_cache")
    run.status = "Done"
    session.commit()
    conn.close()
    return run


# ------------------------------------------------------------------ resume (pyani_plus/public_cli.py:702-828)
def resume(database: Path | str, *, run_id: int | None = None, cache: Path | None = None, temp: Path | None = None,
           logger: logging.Logger | None = None, engine=None, ingest: str = "json", gpus: int = 1,
           engine_factory: str | None = None) -> Run:
    """Complete a partial run of this backend: the missing subject columns are computed, the run is marked done.

    Same checks and messages as the reference's ``resume``: the database and the run must exist, the tool recorded
    with the run must be the one at hand (``We have ... but run-id N used ... instead``), the FASTA directory and
    every file of the run must still be there.  A complete run is left as it is."""
    logger = logger or logging.getLogger("pyani_plus_amd")
    if str(database) == ":memory:" or not Path(database).is_file():
        sourmash_hip.log_sys_exit(logger, f"Database {database} does not exist")
    conn = connect_to_db(database)
    if run_id is None:
        row = conn.execute("SELECT MAX(run_id) FROM runs").fetchone()
        if row is None or row[0] is None:
            sourmash_hip.log_sys_exit(logger, f"Database {database} contains no runs.")
        run_id = row[0]
        logger.info("Resuming run-id %d", run_id)
    try:
        run = load_run(conn, run_id)
    except ValueError:
        sourmash_hip.log_sys_exit(logger, f"Database {database} has no run-id {run_id}.")
    config = run.configuration
    n = len(run.fasta_hashes)
    logger.info("This is a %s run on %d genomes, using %s version %s", config.method, n, config.program, config.version)
    if not n:
        sourmash_hip.log_sys_exit(logger, f"No genomes recorded for run-id {run_id}, cannot resume.")
    from .methods import fastani_hip

    if config.method not in {sourmash_hip.METHOD, fastani_hip.METHOD}:
        sourmash_hip.log_sys_exit(logger, f"Unknown method {config.method} for run-id {run_id} in {database}")
    tool = sourmash_hip.get_sourmash_hip()
    if tool.exe_path.stem != config.program or tool.version != config.version:
        sourmash_hip.log_sys_exit(
            logger,
            f"We have {tool.exe_path.stem} version {tool.version}, but run-id {run_id} used {config.program} version {config.version} instead.",
        )
    fasta = Path(run.fasta_directory)
    if not fasta.is_dir():
        sourmash_hip.log_sys_exit(logger, f"run-id {run_id} used input folder {fasta}, but that is not a directory (now).")
    for link in run.fasta_hashes:
        if not (fasta / link.fasta_filename).is_file():
            sourmash_hip.log_sys_exit(
                logger, f"run-id {run_id} used {fasta / link.fasta_filename} with MD5 {link.genome_hash} but this FASTA file no longer exists"
            )
    session = Session(conn, run)
    run.status = "Resuming"
    session.commit()
    mark = _phase_clock(None)
    tmp_dir = Path(temp) if temp else Path(tempfile.mkdtemp(prefix="pyani_hip_"))
    tmp_dir.mkdir(parents=True, exist_ok=True)
    if config.method == fastani_hip.METHOD:
        direct = _compute_missing_fastani(logger, conn, session, run, tmp_dir, engine, gpus, engine_factory, ingest, mark)
        return _finish_run(logger, conn, session, run, direct, mark)
    cache_dir = Path(tempfile.mkdtemp(prefix="pyani_hip_cache_")) if cache is None else Path(cache)
    cache_dir.mkdir(parents=True, exist_ok=True)
    gpus = max(1, min(int(gpus), n))
    if gpus > 1 and count_run_comparisons(conn, run) != n * n:
        # the same executor as the run itself (pyani_plus/public_cli.py:243-261 re-runs the missing columns through the
        # workflow they came from): worker processes, one all-gather, and each rank's tile cut down to the missing columns
        columns = _incomplete_columns(conn, run)
        logger.info("%d subject columns to compute on %d worker processes", len(columns), gpus)
        run.status = "Running"
        session.commit()
        names = [fasta / a.fasta_filename for a in sorted(run.fasta_hashes, key=lambda a: a.fasta_filename)]
        sharded = _sharded_sourmash_tiles(logger, fasta, names, config, cache_dir, tmp_dir, gpus, engine_factory, columns=columns)
        if sharded is None:
            run.status = "Worker interrupted"
            return _finish_run(logger, conn, session, run, None, mark)
        meta, tile_files, _results = sharded
        recorded = {a.fasta_filename: a.genome_hash for a in run.fasta_hashes}
        for m in meta:  # the files must still be the ones the run was made from
            if recorded[Path(m["path"]).name] != m["md5"]:
                sourmash_hip.log_sys_exit(
                    logger, f"run-id {run_id} used {m['path']} with MD5 {recorded[Path(m['path']).name]} but the file now has MD5 {m['md5']}"
                )
        for tile_file in tile_files:
            import_tile(logger, conn, run, tile_file)
        return _finish_run(logger, conn, session, run, None, mark)
    direct = _compute_missing(logger, conn, session, run, cache_dir, tmp_dir, engine, ingest, mark)
    return _finish_run(logger, conn, session, run, direct, mark)


# ------------------------------------------------------------------ export-run (pyani_plus/public_cli.py:974-1091)
def filename_stem(filename: str) -> str:
    """The file name without directory, ``.gz`` and extension (pyani_plus/utils.py:93-105)."""
    if "/" in filename:
        filename = filename.rsplit("/", 1)[1]
    return Path(filename[:-3]).stem if filename.endswith(".gz") else Path(filename).stem


def export_run(database: Path | str, outdir: Path, *, run_id: int | None = None, label: str = "stem",
               logger: logging.Logger | None = None) -> list[Path]:
    """Write ``<method>_run_<id>.tsv`` (long form) and the six matrices ``<method>_{identity,aln_lengths,sim_errors,
    query_cov,hadamard,tANI}.tsv`` of a run, byte for byte what the reference's ``export-run`` writes from the same
    database: long form in ``Run.comparisons()`` order with ``NA`` for NULL and Python ``str(float)``; matrices from
    the cached ``df_*`` strings through pandas ``to_csv(sep="\t")``, labelled by ``md5``, ``filename`` or ``stem`` and
    sorted by label (db_orm.py:590-624).  A partial run gets the long form only and then the reference's error."""
    import math
    from io import StringIO

    import pandas as pd

    logger = logger or logging.getLogger("pyani_plus_amd")
    if str(database) == ":memory:" or not Path(database).is_file():
        sourmash_hip.log_sys_exit(logger, f"Database {database} does not exist")
    outdir = Path(outdir)
    if not outdir.is_dir():
        logger.warning("Output directory %s does not exist, making it.", outdir)
        outdir.mkdir()
    conn = connect_to_db(database)
    if run_id is None:
        row = conn.execute("SELECT MAX(run_id) FROM runs").fetchone()
        if row is None or row[0] is None:
            sourmash_hip.log_sys_exit(logger, f"Database {database} contains no runs.")
        run_id = row[0]
        logger.info("Exporting run-id %d", run_id)
    try:
        run = load_run(conn, run_id)
    except ValueError:
        sourmash_hip.log_sys_exit(logger, f"Database {database} has no run-id {run_id}.")
    if not run.fasta_hashes:
        sourmash_hip.log_sys_exit(logger, f"Run-id {run_id} has no genomes")
    method = run.configuration.method
    if label == "md5":
        mapping = {a.genome_hash: a.genome_hash for a in run.fasta_hashes}
    elif label == "filename":
        mapping = {a.genome_hash: a.fasta_filename for a in run.fasta_hashes}
    elif label == "stem":
        mapping = {a.genome_hash: filename_stem(a.fasta_filename) for a in run.fasta_hashes}
    else:
        sourmash_hip.log_sys_exit(logger, f"Unexpected label scheme {label!r}")

    def float_or_na(value) -> str:
        return "NA" if value is None else str(value)

    written = [outdir / f"{method}_run_{run_id}.tsv"]
    rows = conn.execute(
        "SELECT c.query_hash, c.subject_hash, c.identity, c.cov_query, c.cov_subject, c.aln_length, c.sim_errors FROM comparisons c "
        "JOIN runs_genomes rq ON c.query_hash = rq.genome_hash AND rq.run_id = ? "
        "JOIN runs_genomes rs ON c.subject_hash = rs.genome_hash AND rs.run_id = ? WHERE c.configuration_id = ? "
        "ORDER BY c.comparison_id",
        (run.run_id, run.run_id, run.configuration_id),
    ).fetchall()
    with written[0].open("w") as handle:
        handle.write("#Query\tSubject\tIdentity\tQuery-Cov\tSubject-Cov\tHadamard\ttANI\tAlign-Len\tSim-Errors\n")
        for q, s_hash, identity, cov_query, cov_subject, aln_length, sim_errors in rows:
            hadamard = None if identity is None or cov_query is None else identity * cov_query
            tani = None if hadamard is None else -math.log(hadamard)
            handle.write(
                f"{mapping[q]}\t{mapping[s_hash]}\t{float_or_na(identity)}\t{float_or_na(cov_query)}\t{float_or_na(cov_subject)}"
                f"\t{float_or_na(hadamard)}\t{float_or_na(tani)}\t{float_or_na(aln_length)}\t{float_or_na(sim_errors)}\n"
            )
    logger.info("Wrote long-form to %s", written[0])
    n = len(run.fasta_hashes)
    if len(rows) != n * n:  # db_orm.load_run(check_complete=True)
        sourmash_hip.log_sys_exit(logger, f"run-id {run_id} has {len(rows)} of {n}^2={n * n} comparisons, {n * n - len(rows)} needed")
    cached = conn.execute(
        "SELECT df_identity, df_aln_length, df_sim_errors, df_cov_query, df_hadamard FROM runs WHERE run_id=?", (run_id,)
    ).fetchone()
    if any(c is None for c in cached):
        out = cache_comparisons(conn, run)
        cached = tuple(out.get(k) for k in ("df_identity", "df_aln_length", "df_sim_errors", "df_cov_query", "df_hadamard"))
        if any(c is None for c in cached):
            sourmash_hip.log_sys_exit(logger, f"Could not load run {method} matrix")
    frames = [pd.read_json(StringIO(c), orient="split", dtype=float) for c in cached]
    with np.errstate(divide="ignore", invalid="ignore"):
        frames.append(-np.log(frames[4]))  # tANI = -ln(hadamard), not cached (db_orm.py:566-588)
    if label == "stem" and len(set(mapping.values())) < len(mapping):
        sourmash_hip.log_sys_exit(logger, "Duplicate filename stems, consider using MD5 labelling.")
    for frame, kind in zip(frames, ("identity", "aln_lengths", "sim_errors", "query_cov", "hadamard", "tANI")):
        if label != "md5":
            frame = frame.rename(index=mapping, columns=mapping).sort_index(axis=0).sort_index(axis=1)  # noqa: PLW2901
        written.append(outdir / f"{method}_{kind}.tsv")
        frame.to_csv(written[-1], sep="\t")
    logger.info("Wrote matrices to %s/%s_*.tsv", outdir, method)
    conn.close()
    return written


# ------------------------------------------------------------------ the fragment-ANI run (pyani_plus/public_cli.py:502-554)
def _compute_missing_fastani(logger, conn, session, run: Run, tmp_dir: Path, engine, gpus: int, engine_factory: str | None,
                             ingest: str = "json", mark=None, query_batch: int | None = None):
    """The incomplete subject columns of a ``fastANI-hip`` run: in this process (one call per run of missing columns; a
    new run is one call for all of them), or as reference ranges of ``pa_fragani`` spread over ``gpus`` worker
    processes -- the reference's own one-process-per-column layout (pyani_plus/public_cli.py:236-261) with a GPU per
    process and no exchange between them.  Every worker writes the reference's JSON column file.  ``ingest="json"``:
    this process imports those files, what the reference's parent does (pyani_plus/workflows/__init__.py:75-87);
    ``"direct"``: the rows go from the result arrays (binary tile files between processes) straight into the table.
    Returns the ``_ingest_direct`` tuple when a new run went in directly (matrix cache from memory), else None."""
    from .methods import fastani_hip

    mark = mark or (lambda _name: None)
    n = len(run.fasta_hashes)
    done = count_run_comparisons(conn, run)
    if done == n * n:
        logger.info("Database already has all %d=%d^2 %s comparisons", n * n, n, fastani_hip.METHOD)
        return None
    logger.info("Database already has %d of %d^2=%d %s comparisons, %d needed", done, n, n * n, fastani_hip.METHOD, n * n - done)
    hashes = sorted(a.genome_hash for a in run.fasta_hashes)
    columns = hashes if done == 0 else _incomplete_columns(conn, run)
    run.status = "Running"
    session.commit()
    hash_to_filename = {a.genome_hash: a.fasta_filename for a in run.fasta_hashes}
    lengths = dict(conn.execute("SELECT genome_hash, length FROM genomes"))
    query_hashes = {h: lengths[h] for h in hash_to_filename}
    fasta_dir = Path(run.fasta_directory)
    col_idx = [hashes.index(c) for c in columns]
    # contiguous runs of missing columns; a new run is one run of all columns
    runs_of_columns: list[tuple[int, int]] = []
    for i in col_idx:
        if runs_of_columns and runs_of_columns[-1][1] == i:
            runs_of_columns[-1] = (runs_of_columns[-1][0], i + 1)
        else:
            runs_of_columns.append((i, i + 1))
    direct = ingest == "direct"
    blocks: list[tuple] = []  # (queries, subjects, identity, aln_length, sim_errors, cov_query, is_null) per block, direct route
    gpus = max(1, min(int(gpus), len(columns)))
    if gpus > 1:
        from . import launch
        from .distributed import shard_bounds_by_cost

        # every rank maps all queries; what differs is the reference range, whose cost follows the subjects' lengths
        pieces = [(a + i, a + i + 1) for a, b in runs_of_columns for i in range(b - a)]
        bounds = shard_bounds_by_cost([max(1, lengths[hashes[a]]) for a, _ in pieces], gpus)
        column_ranges = []
        for a, b in bounds:
            if a == b:
                column_ranges.append((0, 0))
                continue
            lo, hi = pieces[a][0], pieces[b - 1][1]
            if hi - lo != b - a:  # the rank's share is not one range (scattered missing columns): widen it, rows are idempotent
                logger.debug("rank share %s widened to columns %d..%d", (a, b), lo, hi)
            column_ranges.append((lo, hi))
        work_dir = tmp_dir / f"{fastani_hip.METHOD}.run_{run.run_id}.workers"
        spec = {
            "task": "fastani", "run_id": run.run_id, "fasta_dir": str(fasta_dir), "hash_to_filename": hash_to_filename,
            "query_hashes": query_hashes, "column_ranges": column_ranges, "work_dir": str(work_dir), "tiles": direct,
            "configuration": {**{k: getattr(run.configuration, k) for k in wire.CONFIG_FIELDS}, "configuration_id": run.configuration_id},
        }  # fmt: skip
        if engine_factory:
            spec["engine_factory"] = engine_factory
        if query_batch:
            spec["query_batch"] = int(query_batch)
        try:
            results = launch.launch_workers(gpus, spec, work_dir)
        except launch.WorkerFailure as err:
            sourmash_hip.log_sys_exit(logger, str(err))
        mark("workers")
        for rank, r in enumerate(results):
            if r.get("interrupted"):
                run.status = "Worker interrupted"
            if direct:
                for tile in r.get("tiles") or sorted(str(t) for t in work_dir.glob(f"{fastani_hip.METHOD}.rank_{rank}.tile_*.npz")):
                    try:
                        _cfg, queries, subjects, ident, cov, null, aln, sim = wire.load_tile(Path(tile), with_proxies=True)
                    except (ValueError, OSError, KeyError, zipfile.BadZipFile) as err:
                        # tiles are written under another name and renamed: a rank that ended while it wrote (interrupted, or
                        # ended by this process) may leave an unreadable one, and the other ranks' batches still go in; from
                        # a rank that reported success it is damage, and the run must not end quietly incomplete
                        if not r.get("interrupted"):
                            raise
                        logger.warning("Skipping unreadable tile file %s of rank %d: %s", tile, rank, err)
                        continue
                    blocks.append((queries, subjects, ident, aln, sim, cov, null))
            else:
                # the rank's column file: a complete JSON document after every finished query batch, also when the rank
                # was interrupted (or ended by this process while it waited) and reported nothing about it
                c0, c1 = column_ranges[rank]
                json_file = Path(r["json"]) if r.get("json") else work_dir / f"{fastani_hip.METHOD}.run_{run.run_id}.columns_{c0 + 1}_{c1}.json"
                if c0 != c1 and json_file.is_file():
                    try:
                        import_json_comparisons(logger, conn, json_file)
                    except (ValueError, OSError) as err:  # a column file cut short by the end of its rank (it is rewritten whole after every batch)
                        if not r.get("interrupted"):
                            raise
                        logger.warning("Skipping unreadable column file %s of rank %d: %s", json_file, rank, err)
        if run.status == "Worker interrupted":
            session.commit()
    else:
        if engine is None and engine_factory:  # the workers' engine, when their work has shrunk to one process's worth
            import importlib

            module, _, attr = engine_factory.partition(":")
            engine = getattr(importlib.import_module(module), attr)()
        for a, b in runs_of_columns:
            json_file = tmp_dir / f"{fastani_hip.METHOD}.run_{run.run_id}.columns_{a + 1}_{b}.json"
            status = fastani_hip.compute_fastani_hip(
                logger, tmp_dir, session, run, json_file, fasta_dir, hash_to_filename, {v: k for k, v in hash_to_filename.items()},
                query_hashes, "", engine=engine, subject_range=(a, b), on_block=(lambda *blk: blocks.append(blk)) if direct else None,
                **({"query_batch": int(query_batch)} if query_batch else {}),
            )  # fmt: skip
            if status:
                sourmash_hip.log_sys_exit(logger, f"Column worker failed with return code {status}")
            if not direct:
                import_json_comparisons(logger, conn, json_file)
            if run.status == "Worker interrupted":
                break
        mark("worker")
    if not direct:
        mark("import_column_files")
        return None
    if run.status == "Worker interrupted":  # whatever blocks arrived go in as they are; the run stays partial
        for queries, subjects, ident, aln, sim, cov, null in blocks:
            ingest_matrices(conn, run, queries, subjects, ident, cov, null, aln_length=aln, sim_errors=sim)
        return None
    whole = done == 0 and sum(len(b[0]) * len(b[1]) for b in blocks) == n * n
    if whole:  # a new run: one square, rows in index order, matrix cache from memory
        pos = {h: i for i, h in enumerate(hashes)}
        ident = np.full((n, n), np.nan)
        cov = np.full((n, n), np.nan)
        null = np.ones((n, n), dtype=bool)
        aln = np.zeros((n, n), dtype=np.int64)
        sim = np.zeros((n, n), dtype=np.int64)
        for queries, subjects, b_ident, b_aln, b_sim, b_cov, b_null in blocks:
            at = np.ix_([pos[q] for q in queries], [pos[x] for x in subjects])
            ident[at], cov[at], null[at], aln[at], sim[at] = b_ident, b_cov, b_null, b_aln, b_sim
        return _ingest_direct(conn, run, hashes, hashes, ident, cov, null, mark, aln_length=aln, sim_errors=sim)
    for queries, subjects, b_ident, b_aln, b_sim, b_cov, b_null in blocks:
        ingest_matrices(conn, run, queries, subjects, b_ident, b_cov, b_null, aln_length=b_aln, sim_errors=b_sim)
    mark("insert_rows")
    return None


def run_fastani_hip(  # noqa: PLR0913
    fasta: Path,
    database: Path | str,
    *,
    name: str | None = None,
    kmersize: int | None = None,
    fragsize: int | None = None,
    minmatch: float | None = None,
    temp: Path | None = None,
    logger: logging.Logger | None = None,
    engine=None,
    gpus: int = 1,
    engine_factory: str | None = None,
    timings: dict | None = None,
    ingest: str = "json",
    query_batch: int | None = None,
) -> Run:
    """FASTA directory -> database with all N^2 fragment-ANI comparisons and cached matrices: counterpart of
    ``pyani-plus fastani <fasta> -d <db> --create-db`` (pyani_plus/public_cli.py:502-554) with one in-process call --
    or ``gpus`` worker processes, each mapping all queries against its own range of subject columns -- in place of the
    snakemake jobs.  Defaults as pyani_plus/methods/fastani.py:27-30.  The registration pass (checksum, length and
    title of every file) runs on host threads only, so this process never touches a GPU when ``gpus`` > 1."""
    from .engine import load_fasta_files
    from .methods import fastani_hip

    mark = _phase_clock(timings)
    logger = logger or logging.getLogger("pyani_plus_amd")
    kmersize = fastani_hip.KMER_SIZE if kmersize is None else int(kmersize)
    fragsize = fastani_hip.FRAG_LEN if fragsize is None else int(fragsize)
    minmatch = fastani_hip.MIN_FRACTION if minmatch is None else float(minmatch)
    fasta = Path(fasta)
    fasta_names = check_fasta(logger, fasta)
    tool = fastani_hip.get_fastani_hip()
    conn = connect_to_db(database)
    config = db_configuration(conn, fastani_hip.METHOD, tool.exe_path.stem, tool.version, fragsize=fragsize, kmersize=kmersize, minmatch=minmatch)
    infos, _arena = load_fasta_files(fasta_names)
    filename_to_md5: dict[Path, str] = {}
    for filename, info in zip(fasta_names, infos):
        if info.status != 0:
            sourmash_hip.log_sys_exit(logger, info.message)
        if info.md5 in filename_to_md5.values():
            _duplicate_md5_exit(logger, info.md5, [k for k, v in filename_to_md5.items() if v == info.md5] + [filename])
        filename_to_md5[filename] = info.md5
        db_genome(conn, filename, info.md5, info.length, info.description)
    del _arena
    mark("register_genomes")
    run = add_run(
        conn, config, " ".join(sys.argv), fasta, "Initialising",
        f"{len(filename_to_md5)} genomes using {fastani_hip.METHOD}" if name is None else name, filename_to_md5,
    )  # fmt: skip
    session = Session(conn, run)
    tmp_dir = Path(temp) if temp else Path(tempfile.mkdtemp(prefix="pyani_hip_"))
    tmp_dir.mkdir(parents=True, exist_ok=True)
    if ingest not in {"json", "direct"}:
        sourmash_hip.log_sys_exit(logger, f"ingest must be 'json' or 'direct', not {ingest!r}")
    direct = _compute_missing_fastani(logger, conn, session, run, tmp_dir, engine, gpus, engine_factory, ingest, mark, query_batch)
    return _finish_run(logger, conn, session, run, direct, mark)


# ------------------------------------------------------------------ the driver as a process
def main(argv: list[str] | None = None) -> int:
    """``python -m pyani_plus_amd.rundb {sourmash,fastani,resume,export-run} ...``: the run driver as a process of its own,
    with SIGINT and SIGTERM arriving as ``KeyboardInterrupt`` the way the reference's worker command arranges it
    (pyani_plus/private_cli.py:816-823), so that ``scancel`` / ``kill`` leave the finished batches recorded and the run
    marked "Worker interrupted" exactly as Ctrl-C does.  Only what the drivers above take as arguments; the reference's
    Typer front end is out of scope."""
    import argparse

    from . import launch

    parser = argparse.ArgumentParser(prog="python -m pyani_plus_amd.rundb", description=main.__doc__)
    sub = parser.add_subparsers(dest="command", required=True)

    def common(p, *, run_options: bool) -> None:
        p.add_argument("--database", "-d", required=True, type=Path)
        p.add_argument("--temp", type=Path, default=None)
        p.add_argument("--gpus", type=int, default=1)
        p.add_argument("--ingest", choices=("json", "direct"), default="json")
        p.add_argument("--engine-factory", default=None, help="module:callable that makes the engine (tests)")
        p.add_argument("--verbose", "-v", action="store_true")
        if run_options:
            p.add_argument("fasta", type=Path)
            p.add_argument("--name", default=None)

    p_s = sub.add_parser("sourmash", help="FASTA directory -> all N^2 sourmash-hip comparisons")
    common(p_s, run_options=True)
    p_s.add_argument("--cache", type=Path, default=None)
    p_s.add_argument("--kmersize", type=int, default=sourmash_hip.KMER_SIZE)
    p_s.add_argument("--scaled", type=int, default=sourmash_hip.SCALED)
    p_f = sub.add_parser("fastani", help="FASTA directory -> all N^2 fastANI-hip comparisons")
    common(p_f, run_options=True)
    p_f.add_argument("--kmersize", type=int, default=None)
    p_f.add_argument("--fragsize", type=int, default=None)
    p_f.add_argument("--minmatch", type=float, default=None)
    p_f.add_argument("--query-batch", type=int, default=None)
    p_r = sub.add_parser("resume", help="complete a partial run")
    common(p_r, run_options=False)
    p_r.add_argument("--run-id", type=int, default=None)
    p_r.add_argument("--cache", type=Path, default=None)
    p_e = sub.add_parser("export-run", help="long form and matrices of a run as TSV files")
    p_e.add_argument("--database", "-d", required=True, type=Path)
    p_e.add_argument("--outdir", "-o", required=True, type=Path)
    p_e.add_argument("--run-id", type=int, default=None)
    p_e.add_argument("--label", choices=("md5", "filename", "stem"), default="stem")
    p_e.add_argument("--verbose", "-v", action="store_true")
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO, format="%(levelname)s %(message)s")
    logger = logging.getLogger("pyani_plus_amd")
    with launch.signals_as_interrupt():
        if args.command == "sourmash":
            run = run_sourmash_hip(args.fasta, args.database, cache=args.cache, name=args.name, kmersize=args.kmersize, scaled=args.scaled,
                                   temp=args.temp, logger=logger, ingest=args.ingest, gpus=args.gpus, engine_factory=args.engine_factory)
        elif args.command == "fastani":
            run = run_fastani_hip(args.fasta, args.database, name=args.name, kmersize=args.kmersize, fragsize=args.fragsize,
                                  minmatch=args.minmatch, temp=args.temp, logger=logger, ingest=args.ingest, gpus=args.gpus,
                                  engine_factory=args.engine_factory, query_batch=args.query_batch)
        elif args.command == "resume":
            run = resume(args.database, run_id=args.run_id, cache=args.cache, temp=args.temp, logger=logger, ingest=args.ingest,
                         gpus=args.gpus, engine_factory=args.engine_factory)
        else:
            for path in export_run(args.database, args.outdir, run_id=args.run_id, label=args.label, logger=logger):
                print(path)
            return 0
    logger.info("run-id %d: %s", run.run_id, run.status)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
