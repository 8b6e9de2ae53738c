"""The ``sourmash-hip`` method: pyani-plus's sourmash path on an MI355X.

Drop-in shaped like ``pyani_plus/methods/sourmash.py`` plus its column worker
``private_cli.compute_sourmash`` (pyani_plus/private_cli.py:1803-1902):

* module constants ``SCALED`` / ``KMER_SIZE`` read by the CLI for its defaults
  (pyani_plus/public_cli.py:58, 612-613);
* ``prepare_genomes(logger, run, cache)`` -- the hook ``private_cli.prepare`` finds by
  importing ``pyani_plus.methods.<method>`` (pyani_plus/private_cli.py:725-752);
* ``compute_sourmash_hip(...)`` -- same positional signature, return codes, JSON file and
  interrupt behaviour as the ``compute`` dict entries (pyani_plus/private_cli.py:906-968).

``run`` / ``session`` are duck-typed (``run.configuration.{method,program,version,kmersize,
extra}``, ``run.fasta_directory``, ``run.fasta_hashes[*].{fasta_filename,genome_hash}``,
``run.status``, ``session.commit()``): the reference's SQLAlchemy objects satisfy this,
and so do the plain dataclasses of ``pyani_plus_amd.rundb``.

Where the reference launches processes, this module makes library calls:
``sourmash scripts singlesketch`` -> ``HipEngine.sketch`` (one batched launch for all
missing genomes), ``sig collect`` + ``manysearch`` -> ``HipEngine.pair_counts``, and the
CSV columns -> ``ani_host`` (host libm ``pow``: bit-identical to the reference fixtures).
"""

from __future__ import annotations

import logging
import sys
from collections.abc import Iterator
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from .. import __version__, _capi, sig, wire

METHOD = "sourmash-hip"
SCALED = 1000  # same defaults as pyani_plus/methods/sourmash.py:30-31
KMER_SIZE = 31
RECORDING_FAILED = 2  # pyani_plus/private_cli.py:188
PREPARE_BATCH_BASES = 2_000_000_000  # residues sketched per launch by prepare_genomes


def log_sys_exit(logger: logging.Logger, msg: str):
    """Log the message as an error and exit with it (pyani_plus/__init__.py:120-126)."""
    logger.error(msg)
    sys.exit(msg)


@dataclass(frozen=True)
class ExternalToolData:
    """Same two fields as pyani_plus/tools.py:39-43; ``exe_path.stem`` and ``version`` are what
    the reference stores as ``Configuration.program`` / ``.version`` (public_cli.py:145-146)."""

    exe_path: Path
    version: str


def get_sourmash_hip() -> ExternalToolData:
    """The "tool" of this method is the HIP shared library (counterpart of tools.get_sourmash)."""
    _capi.load_library()  # raises HipBackendError when the extension is missing
    return ExternalToolData(_capi.LIB_PATH, __version__)


def _check_tool_version(logger: logging.Logger, tool: ExternalToolData, configuration) -> None:
    """Abort when the run was recorded with another program/version (private_cli.py:191-223)."""
    if configuration.program != tool.exe_path.stem or configuration.version != tool.version:
        msg = (
            f"Run configuration was {configuration.program} {configuration.version}"
            f" but we have {tool.exe_path.stem} {tool.version}"
        )
        log_sys_exit(logger, msg)


def parse_scaled(extra: str) -> int:
    """``extra`` is spliced into the sketch parameters as ``scaled=N`` (sourmash.py:75-76)."""
    if not extra.startswith("scaled="):
        msg = f"sourmash-hip supports extra='scaled=N' only, not {extra!r}"
        raise ValueError(msg)
    try:
        scaled = int(extra[len("scaled=") :])
    except ValueError:
        msg = f"sourmash-hip supports extra='scaled=N' only, not {extra!r}"
        raise ValueError(msg) from None
    if scaled < 1:
        msg = f"scaled must be a positive integer, not {scaled}"
        raise ValueError(msg)
    return scaled


def sig_cache_dir(cache: Path, kmersize: int, extra: str) -> Path:
    """Same sub-directory as the reference so both backends share signatures (sourmash.py:57)."""
    return Path(cache) / f"sourmash_k={kmersize}_{extra}"


_ENGINE = None
DEVICE_ENV = "PYANI_HIP_DEVICE"


def resolve_device(spread_key: int | None = None) -> int:
    """Which GPU this process computes on.

    The reference's column workers are separate processes (pyani_plus/private_cli.py:853-861, one per subject column
    except for sourmash, pyani_plus/public_cli.py:232-261), so a node with several GPUs is used by giving each
    process its own device and nothing else:

    * ``PYANI_HIP_DEVICE=<index>`` -- that device;
    * ``PYANI_HIP_DEVICE=spread`` -- ``spread_key`` modulo the number of devices, where the worker passes the number
      of its subject column: the per-column processes of one run then cover the GPUs round-robin;
    * unset -- ``LOCAL_RANK`` (ranks started by ``pyani_plus_amd.launch`` or ``torch.distributed.run``) modulo the
      number of devices, else device 0.

    Meant for the process that goes on to compute (counting the devices may bring up the HIP runtime: a parent that only
    starts workers counts them in a child process, ``launch.visible_devices``); an index beyond the last device is an
    error of the caller's environment and is reported by ``HipEngine``."""
    import os

    want = os.environ.get(DEVICE_ENV, "").strip().lower()
    if want and want not in {"auto", "spread"}:
        try:
            return int(want)
        except ValueError:
            msg = f"{DEVICE_ENV} must be a device index, 'spread' or 'auto', not {want!r}"
            raise ValueError(msg) from None
    import torch

    n_dev = max(1, torch.cuda.device_count())
    if want == "spread" and spread_key is not None:
        return int(spread_key) % n_dev
    local_rank = os.environ.get("LOCAL_RANK", "")
    return int(local_rank) % n_dev if local_rank.isdigit() else 0


def get_engine(device: int | None = None, *, spread_key: int | None = None):
    """Process-wide HipEngine (created on first use; raises HipBackendError without a GPU).  ``device`` None: the
    device ``resolve_device`` names (environment; 0 by default)."""
    global _ENGINE
    if _ENGINE is None:
        from ..engine import HipEngine

        _ENGINE = HipEngine(resolve_device(spread_key) if device is None else device)
    return _ENGINE


# Sketches this process has just written, keyed by signature file: when prepare and compute run in one
# process (rundb.run_sourmash_hip) the column worker takes them from here instead of parsing the files it
# wrote a moment ago.  The files stay the contract between processes; an entry is only used while its file
# still has the size and mtime it had when written.
_RECENT: dict[str, tuple] = {}
_RECENT_LIMIT_BYTES = 1 << 30
_recent_bytes = 0


def _remember_sketch(sig_file: Path, ksize: int, max_hash: int, mins: np.ndarray) -> None:
    global _recent_bytes  # noqa: PLW0603
    if _recent_bytes + mins.nbytes > _RECENT_LIMIT_BYTES:
        _RECENT.clear()
        _recent_bytes = 0
    stat = sig_file.stat()
    _RECENT[str(sig_file)] = (int(ksize), int(max_hash), stat.st_size, stat.st_mtime_ns, mins)
    _recent_bytes += mins.nbytes


def _recall_sketch(sig_file: Path, ksize: int, max_hash: int):
    entry = _RECENT.get(str(sig_file))
    if entry is None or entry[0] != int(ksize) or entry[1] != int(max_hash):
        return None
    stat = sig_file.stat()
    if (stat.st_size, stat.st_mtime_ns) != entry[2:4]:
        return None
    return entry[4]


def sketch_fasta_batches(logger: logging.Logger, paths: list[Path], *, kmersize: int, scaled: int, engine=None,
                         batch_bases: int = PREPARE_BATCH_BASES, needed=None) -> Iterator[tuple[list, list, list]]:
    """FASTA files -> sketches, batch by batch, with the host front-end of batch i+1 running while batch i is on
    the device (the reference reads every genome serially and three times,
    pyani_plus/public_cli.py:158-173, pyani_plus/methods/sourmash.py:67-83).

    Per batch: the threaded loader (read + gunzip + md5 + parse + 2-bit pack, ``pa_fasta_batch_load``) writes the
    packed bases into page-locked memory on a background thread; the device side takes them with
    ``pa_sketch_streamed`` (mask as runs, chunks uploaded on a copy stream behind the hash kernel).
    Yields ``(batch_paths, infos, sketches)``; a file that fails to load ends the run through ``log_sys_exit``
    with the reference's message.  ``needed(info) -> bool`` (optional) says whether a loaded file still needs its
    sketch: a batch none of whose files does is not sent to the device and yields ``None`` sketches."""
    from concurrent.futures import ThreadPoolExecutor

    from ..engine import load_fasta_files, max_hash_for_scaled

    paths = [Path(p) for p in paths]
    if not paths:
        return
    batches: list[list[Path]] = [[]]
    size = 0
    for path in paths:
        est = 4 * path.stat().st_size if path.name.endswith(".gz") and path.is_file() else (path.stat().st_size if path.is_file() else 0)
        if batches[-1] and size + est > batch_bases:
            batches.append([])
            size = 0
        batches[-1].append(path)
        size += est
    max_hash = max_hash_for_scaled(scaled)
    eng = engine
    streamed = engine is None or hasattr(engine, "sketch_streamed")  # the oracle-backed test engine has no device to stream to
    with ThreadPoolExecutor(max_workers=1) as loader:
        pending = loader.submit(load_fasta_files, batches[0], pinned=streamed)
        for i, batch in enumerate(batches):
            infos, arena = pending.result()
            if i + 1 < len(batches):
                pending = loader.submit(load_fasta_files, batches[i + 1], pinned=streamed)
            for info in infos:
                if info.status != 0:
                    log_sys_exit(logger, info.message)
            if needed is not None and not any(needed(info) for info in infos):
                yield batch, infos, [None] * len(infos)
                continue
            if eng is None:
                eng = get_engine()
            if streamed and hasattr(eng, "sketch_streamed"):
                _dev, sk = eng.sketch_streamed(eng.pin_arena(arena), kmersize, scaled, max_hash=max_hash)
            else:
                sk = eng.sketch(eng.upload(arena), kmersize, scaled, max_hash=max_hash)
            yield batch, infos, sk.to_host()


def prepare_genomes(logger: logging.Logger, run, cache: Path, *, engine=None, presketched: dict | None = None) -> Iterator:
    """Build the sketch signatures in ``cache/sourmash_k={kmersize}_scaled={N}``.

    Yields the run's FASTA entries as their signatures are completed (progress bar),
    skipping genomes whose ``.sig`` already exists -- the contract of
    pyani_plus/methods/sourmash.py:34-84.

    ``presketched`` = ``{genome_hash: mins}`` from an earlier ``sketch_fasta_batches`` pass lets a caller that
    has just read the files for their checksums (``rundb.run_sourmash_hip``) hand the sketches over instead of
    having the genomes read and hashed a second time; genomes not in it are sketched here.
    """
    config = run.configuration
    if config.method != METHOD:
        log_sys_exit(logger, f"Expected run to be for {METHOD}, not method {config.method}")
    if not config.kmersize:
        log_sys_exit(logger, f"{METHOD} requires a k-mer size, default is {KMER_SIZE}")
    if not config.extra:
        log_sys_exit(logger, f"{METHOD} requires extra setting, default is scaled={SCALED}")
    scaled = parse_scaled(config.extra)
    if not 1 <= int(config.kmersize) <= 64:
        log_sys_exit(logger, f"{METHOD} supports k-mer sizes 1 to 64, not {config.kmersize}")
    if not Path(cache).is_dir():
        msg = f"Cache directory '{cache}' does not exist"
        raise ValueError(msg)
    sig_dir = sig_cache_dir(cache, config.kmersize, config.extra)
    logger.debug("Preparing %s signatures in '%s'", METHOD, sig_dir)
    sig_dir.mkdir(exist_ok=True)
    fasta_dir = Path(run.fasta_directory)

    from ..engine import max_hash_for_scaled

    max_hash = max_hash_for_scaled(scaled)

    def write(entries: list, sketches: list[np.ndarray]) -> None:
        sig_files = [sig_dir / f"{entry.genome_hash}.sig" for entry in entries]
        sig.write_sigs(
            sig_files,
            names=[entry.genome_hash for entry in entries],
            filenames=[str(fasta_dir / entry.fasta_filename) for entry in entries],
            ksize=config.kmersize,
            max_hash=max_hash,
            sketches=sketches,
        )
        for sig_file, mins in zip(sig_files, sketches):
            _remember_sketch(sig_file, config.kmersize, max_hash, mins)

    entries = list(run.fasta_hashes)
    missing = [e for e in entries if not (sig_dir / f"{e.genome_hash}.sig").is_file()]
    have = [e for e in missing if presketched is not None and e.genome_hash in presketched]
    if have:
        write(have, [presketched[e.genome_hash] for e in have])
    todo = [e for e in missing if presketched is None or e.genome_hash not in presketched]
    done_until = 0  # entries are yielded in run order, each once its signature exists
    by_path = {fasta_dir / e.fasta_filename: e for e in todo}
    todo_set = {id(e) for e in todo}

    def ready() -> Iterator:
        nonlocal done_until
        while done_until < len(entries) and id(entries[done_until]) not in todo_set:
            yield entries[done_until]
            done_until += 1

    yield from ready()
    if todo:
        try:
            for batch_paths, _infos, sketches in sketch_fasta_batches(
                logger, list(by_path), kmersize=config.kmersize, scaled=scaled, engine=engine
            ):
                batch_entries = [by_path[p] for p in batch_paths]
                write(batch_entries, sketches)
                todo_set.difference_update(id(e) for e in batch_entries)
                yield from ready()
        except _capi.HipBackendError as err:
            backend_failure(logger, f"{METHOD} sketching", err)
    yield from ready()


DEVICE_TILE_COLUMNS = 2048  # subject columns evaluated (and flushed to the column file) per device call


def check_self_comparisons(queries: list[str], subjects: list[str], ident, null) -> None:
    """The reference refuses a self-comparison that is not exactly one (pyani_plus/methods/sourmash.py:119-127)."""
    sub_pos = {s: i for i, s in enumerate(subjects)}
    for qi, q in enumerate(queries):
        si = sub_pos.get(q)
        if si is not None and not null[qi, si] and ident[qi, si] != 1.0:
            msg = f"Expected {METHOD} {q} vs self to be one, not {ident[qi, si]!r}"
            raise ValueError(msg)


def read_cached_sketches(logger: logging.Logger, sig_files: list[Path], kmersize: int, max_hash: int) -> list[np.ndarray]:
    """The sketches of the given signature files: from this process's own writes where they are still current,
    otherwise through the native threaded reader (``sig.read_sigs``) -- the step `sourmash sig collect` +
    `manysearch`'s own loading are in the reference (pyani_plus/methods/sourmash.py:160-200), where prepare and
    compute are separate processes and every file is parsed again.  A missing, damaged or foreign file ends the
    worker the way a failing `sourmash sig collect` does there (through utils.check_output -> log_sys_exit)."""
    sketches: list = [None] * len(sig_files)
    todo = []
    for i, sig_file in enumerate(sig_files):
        if not sig_file.is_file():
            log_sys_exit(logger, f"Missing sourmash signature file '{sig_file}'")
        sketches[i] = _recall_sketch(sig_file, kmersize, max_hash)
        if sketches[i] is None:
            todo.append(i)
    if todo:
        try:
            loaded = sig.read_sigs([sig_files[i] for i in todo], ksize=kmersize, max_hash=max_hash)
        except (ValueError, OSError, KeyError, TypeError) as err:
            log_sys_exit(logger, f"Unreadable sourmash signature file: {err}")
        except _capi.HipBackendError as err:
            backend_failure(logger, f"{METHOD} signature loading", err)
        for i, mins in zip(todo, loaded):
            sketches[i] = mins
    return sketches


def iter_sourmash_tiles(  # noqa: PLR0913
    logger: logging.Logger,
    subject_hashes,
    query_hashes,
    cache: Path,
    *,
    kmersize: int,
    scaled: int,
    engine=None,
    algo: int = _capi.PA_PAIRS_AUTO,
    tile_columns: int = DEVICE_TILE_COLUMNS,
) -> Iterator[tuple]:
    """Yield ``(queries, tile_subjects, query_containment_ani, max_containment_ani, is_null)`` for one tile of
    subject columns after the other (sorted queries x sorted subjects).  The sketches are read and uploaded once;
    every tile is one device call followed by the strict host transform."""
    cache = Path(cache)
    if not cache.is_dir():
        msg = f"Given cache directory '{cache}' does not exist"
        raise ValueError(msg)
    from ..engine import ani_host, max_hash_for_scaled

    queries = sorted(query_hashes)
    subjects = sorted(subject_hashes)
    if not queries or not subjects:
        return
    extra_subjects = sorted(set(subjects) - set(queries))
    order = queries + extra_subjects  # CSR order: queries first, then subjects not among them
    index = {h: i for i, h in enumerate(order)}
    max_hash = max_hash_for_scaled(scaled)
    sketches = read_cached_sketches(logger, [cache / f"{genome_hash}.sig" for genome_hash in order], kmersize, max_hash)
    nq = len(queries)
    sizes = np.array([len(s) for s in sketches], dtype=np.uint64)
    eng = engine or get_engine()
    dsk = eng.sketches_from_host(sketches)
    tile_columns = max(1, int(tile_columns))
    for t0 in range(0, len(subjects), tile_columns):
        tile = subjects[t0 : t0 + tile_columns]
        sub_idx = np.array([index[s] for s in tile])
        # contiguous subject range if possible (all-vs-all, or a single subject column)
        lo, hi = int(sub_idx.min()), int(sub_idx.max()) + 1
        if hi - lo != len(sub_idx):
            lo, hi = 0, len(order)  # scattered subjects: compute the covering block, pick columns below
        counts = eng.pair_counts(dsk, (0, nq), (lo, hi), algo=algo).cpu().numpy().view(np.uint32)
        counts = np.ascontiguousarray(counts[:, sub_idx - lo])
        square = len(tile) == nq and tile == queries  # one tile, all-vs-all: one pow per ordered pair
        ident, cov, null = ani_host(counts, sizes[:nq], sizes[sub_idx], kmersize, symmetric=square)
        check_self_comparisons(queries, tile, ident, null)
        yield queries, tile, cov, ident, null


def compute_sourmash_matrices(
    logger: logging.Logger,
    subject_hashes,
    query_hashes,
    cache: Path,
    *,
    kmersize: int,
    scaled: int,
    engine=None,
    algo: int = _capi.PA_PAIRS_AUTO,
):
    """``(queries, subjects, query_containment_ani, max_containment_ani, is_null)`` for the sorted
    query x subject block: the array form of what ``compute_sourmash_tile`` yields row by row."""
    queries, subjects = sorted(query_hashes), sorted(subject_hashes)
    nq, ns = len(queries), len(subjects)
    if not Path(cache).is_dir():
        msg = f"Given cache directory '{cache}' does not exist"
        raise ValueError(msg)
    if not queries or not subjects:
        empty = np.zeros((nq, ns))
        return queries, subjects, empty, empty.copy(), np.zeros((nq, ns), dtype=bool)
    tiles = list(iter_sourmash_tiles(logger, subjects, queries, cache, kmersize=kmersize, scaled=scaled, engine=engine, algo=algo))
    if len(tiles) == 1:
        return tiles[0]
    return (queries, subjects, np.concatenate([t[2] for t in tiles], axis=1), np.concatenate([t[3] for t in tiles], axis=1),
            np.concatenate([t[4] for t in tiles], axis=1))  # fmt: skip


def compute_sourmash_tile(
    logger: logging.Logger,
    subject_hashes,
    query_hashes,
    cache: Path,
    *,
    kmersize: int,
    scaled: int,
    engine=None,
    algo: int = _capi.PA_PAIRS_AUTO,
) -> Iterator[tuple[str, str, float | None, float | None]]:
    """Yield ``(query_hash, subject_hash, query_containment_ani, max_containment_ani)`` for every
    query x subject pair, ``None, None`` where the sketches share no hash -- what
    ``compute_sourmash_tile`` + ``parse_sourmash_manysearch_csv`` yield in the reference
    (pyani_plus/methods/sourmash.py:87-206).  Row order is deterministic here (query-major)."""
    queries, subjects, cov, ident, null = compute_sourmash_matrices(
        logger, subject_hashes, query_hashes, cache, kmersize=kmersize, scaled=scaled, engine=engine, algo=algo
    )
    for qi, q in enumerate(queries):
        for si, s in enumerate(subjects):
            if null[qi, si]:
                yield q, s, None, None
            else:
                yield q, s, float(cov[qi, si]), float(ident[qi, si])


def backend_failure(logger: logging.Logger, what: str, err: Exception):
    """The reference turns every failed tool call into ``log_sys_exit("Return code N from: <cmd>" + the tool's
    ERROR lines)`` (pyani_plus/utils.py:262-283).  There is no process here, so the analogue of the command is
    the library call and the analogue of the tool's output is ``pa_last_error`` (already in ``err``)."""
    log_sys_exit(logger, f"{what} failed in {_capi.LIB_PATH.name}: {err}")


def compute_sourmash_hip(  # noqa: PLR0913
    logger: logging.Logger,
    tmp_dir: Path,  # noqa: ARG001 - no intermediate files are needed
    session,
    run,
    json_filename: Path,
    fasta_dir: Path,  # noqa: ARG001
    hash_to_filename: dict[str, str],  # noqa: ARG001
    filename_to_hash: dict[str, str],  # noqa: ARG001
    query_hashes: dict[str, int],
    subject_hash: str,
    *,
    cache: Path = Path(),
    engine=None,
    tile_columns: int = DEVICE_TILE_COLUMNS,
) -> int:
    """Run many-vs-subject (or all-vs-all when ``subject_hash == ""``) and log to JSON.

    Field mapping as private_cli.py:1875-1887: ``identity`` <- max-containment ANI,
    ``cov_query`` <- query-containment ANI; ``aln_length``/``sim_errors``/``cov_subject`` unset.
    Subject columns are evaluated ``tile_columns`` at a time and the column file grows by one tile after
    each (complete JSON after every tile), so an interrupt keeps the finished tiles -- the behaviour of the
    reference's flush every 100 000 rows (private_cli.py:1863-1894).  A failing library call ends the worker
    through ``log_sys_exit`` like a failing tool does (utils.py:262-283); a failing save returns 2.
    """
    configuration = run.configuration
    tool = get_sourmash_hip()
    _check_tool_version(logger, tool, configuration)

    sig_cache = sig_cache_dir(cache, configuration.kmersize, configuration.extra)
    if not sig_cache.is_dir():
        log_sys_exit(
            logger,
            f"Missing sourmash signatures directory '{sig_cache}' - check cache setting '{cache}'.",
        )
    scaled = parse_scaled(configuration.extra)
    try:
        writer = wire.ColumnFileWriter(logger, json_filename, configuration)
    except Exception:
        logger.exception("Unexpected exception saving JSON:")
        return RECORDING_FAILED
    try:
        tiles = iter_sourmash_tiles(
            logger,
            {subject_hash} if subject_hash else set(query_hashes),
            set(query_hashes),
            sig_cache,
            kmersize=configuration.kmersize,
            scaled=scaled,
            engine=engine,
            tile_columns=tile_columns,
        )
        for queries, subjects, cov, ident, null in tiles:
            try:
                # identity <- max-containment ANI, cov_query <- query-containment ANI (private_cli.py:1879-1880)
                writer.append(queries, subjects, ident, cov, null)
            except Exception:
                logger.exception("Unexpected exception saving JSON:")
                return RECORDING_FAILED
    except KeyboardInterrupt:
        # abort gracefully (private_cli.py:1889-1894): the finished tiles are in the file already
        logger.error("Interrupted with %d completed %s comparisons", writer.rows, METHOD)  # noqa: TRY400
        run.status = "Worker interrupted"
        session.commit()
    except _capi.HipBackendError as err:
        backend_failure(logger, f"{METHOD} comparison", err)
    return 0
