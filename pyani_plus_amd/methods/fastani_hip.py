"""The ``fastANI-hip`` method: pyani-plus's fastANI column worker on an MI355X.

Drop-in shaped like ``private_cli.compute_fastani`` (pyani_plus/private_cli.py:976-1117) and the
defaults module ``pyani_plus/methods/fastani.py``:

* constants ``KMER_SIZE`` / ``FRAG_LEN`` / ``MIN_FRACTION`` (fastani.py:27-30);
* ``compute_fastani_hip(...)`` with the positional signature of the ``compute`` dict entries
  (private_cli.py:956-968): queries vs one subject column (or all columns when ``subject_hash``
  is ``""``), JSON column file out, 0 / 2 return codes, interrupt handling.

Where the reference runs ``fastANI --ql <queries> -r <subject> -o out --fragLen F -k K
--minFraction M`` in batches of 500 queries (private_cli.py:1029-1063) and parses
``query ref ANI matched total`` lines (fastani.py:98-120), this module makes one
``pa_fragani`` call for all genomes involved and derives the same five fields
(private_cli.py:1070-1080):

    identity   = 0.01 * ANI           (the reference's own product, pyani_plus/methods/fastani.py:113: the same double,
                                       where ANI / 100 differs in the last place for 4 of the 22 values its fixtures
                                       hold; None when fastANI would print no line)
    aln_length = round(fragsize * matched)
    sim_errors = total - matched
    cov_query  = matched / total

Parity with fastANI itself: every value the reference holds -- its 25 fixture rows and the pins of its tests -- comes
out exactly, the identity to the six digits fastANI prints (oracle/fragani_oracle.c, tests/test_fragani_oracle.py,
DESIGN.md 2 and 4.5).
"""

from __future__ import annotations

import logging
import platform
from pathlib import Path

import numpy as np

from .. import wire
from .sourmash_hip import RECORDING_FAILED, ExternalToolData, _check_tool_version, get_engine, get_sourmash_hip, log_sys_exit

METHOD = "fastANI-hip"
KMER_SIZE = 16  # pyani_plus/methods/fastani.py:27-30
FRAG_LEN = 3000
MIN_FRACTION = 0.2


def get_fastani_hip() -> ExternalToolData:
    """Counterpart of ``tools.get_fastani`` (pyani_plus/tools.py:140-164): the HIP library is the tool."""
    return get_sourmash_hip()


def fastani_print_round(ani_percent: float) -> float:
    """fastANI writes the identity through a C++ stream with the default precision: six significant digits
    (``82.9124``, ``99.9953``, ``100``).  The reference parses that text (pyani_plus/methods/fastani.py:98-120),
    so the stored identity carries exactly those digits."""
    return float(f"{ani_percent:.6g}")


def fastani_mean(ident_sum, matched):
    """The ANI of a pair as fastANI computes it: the float sum of the kept fragments' identities (``ident_sum`` holds that
    float, widened) divided by their number IN FLOAT.  NaN where nothing was kept."""
    ident_sum, matched = np.asarray(ident_sum), np.asarray(matched)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = ident_sum.astype(np.float32) / matched.astype(np.float32)
    return np.where(matched > 0, mean.astype(np.float64), np.nan)


def mappable_lengths(contig_len, contig_genome, n_genomes: int, fragsize: int) -> np.ndarray:
    """Genome length as fastANI counts it for its minFraction test: contigs shorter than one fragment do not count."""
    contig_len = np.asarray(contig_len, dtype=np.int64)
    keep = contig_len >= fragsize
    return np.bincount(np.asarray(contig_genome, dtype=np.int64)[keep], weights=contig_len[keep], minlength=n_genomes).astype(np.int64)


def is_reported(matches: int, frags: int, fragsize: int, minmatch: float, len_query: int, len_subject: int) -> bool:
    """fastANI prints a line when the matched fragments cover at least ``minFraction`` of the SHORTER genome:
    ``matches * fragLen >= minFraction * min(len_q, len_s)`` -- not of the query's fragment count.  A long
    query against a short reference is therefore still reported."""
    return frags > 0 and matches > 0 and matches * fragsize >= minmatch * min(len_query, len_subject)


QUERY_BATCH = 500  # query genomes per device call and per rewrite of the column file (private_cli.py:1029)


def load_genomes_for_fragani(fasta_files: list[Path], engine):
    """FASTA files -> (host arena, device arena); a file that does not load raises ``ValueError`` with the loader's message."""
    from ..engine import load_fasta_files

    infos, arena = load_fasta_files(fasta_files)
    for info in infos:
        if info.status != 0:
            raise ValueError(info.message)
    return arena, engine.upload(arena)


def fragment_ani_matrices(fasta_files: list[Path], *, kmersize: int, fragsize: int, engine=None, ref_range=None):
    """(total_frags[n], matched[n, n], ani_percent[n, n], mappable_length[n]) for the given FASTA files, rows = query."""
    eng = engine or get_engine()
    arena, dev = load_genomes_for_fragani(fasta_files, eng)
    total, matched, ident_sum = eng.fragani(dev, arena.contig_start, arena.contig_len, arena.contig_genome, kmersize, fragsize, ref_range=ref_range)
    with np.errstate(invalid="ignore", divide="ignore"):
        ani = fastani_mean(ident_sum, matched)
    return total, matched, ani, mappable_lengths(arena.contig_len, arena.contig_genome, arena.n_genomes, fragsize)


def comparison_entry(q: str, s: str, frags: int, matches: int, ani_percent: float, fragsize: int, minmatch: float, len_q: int,
                     len_s: int, constants: dict) -> dict:
    """One comparison as the reference's worker records it (private_cli.py:1066-1098): None everywhere when fastANI
    would print no line for the pair."""
    reported = is_reported(matches, frags, fragsize, minmatch, len_q, len_s)
    return {
        "query_hash": q,
        "subject_hash": s,
        "identity": 0.01 * fastani_print_round(float(ani_percent)) if reported else None,  # fastani.py:113
        # proxy values, private_cli.py:1072-1080
        "aln_length": round(fragsize * matches) if reported else None,
        "sim_errors": frags - matches if reported else None,
        "cov_query": matches / frags if reported else None,
        **constants,
    }


def round_sig6(values: np.ndarray) -> np.ndarray:
    """``fastani_print_round`` of every element (native ``pa_round_sig6``: printf("%.6g") and back; NaN stays)."""
    from .. import _capi

    out = np.ascontiguousarray(values, dtype=np.float64).copy()
    _capi.check(_capi.load_library().pa_round_sig6(out.ctypes.data, out.size), "pa_round_sig6")
    return out


def comparison_block(total, matched, ident_sum, lengths, rows, cols, fragsize: int, minmatch: float, col0: int = 0):
    """The five fields of the fastANI worker (private_cli.py:1066-1098) for a block of query rows x subject columns
    as arrays: (identity f64, aln_length i64, sim_errors i64, cov_query f64, is_null bool); the array form of
    ``comparison_entry``.  A pair fastANI would print no line for is NULL in all four.  ``col0``: the genome whose
    column is the first of ``matched`` / ``ident_sum`` (result matrices that hold a range of subject columns only)."""
    rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
    m = matched[np.ix_(rows, cols - col0)].astype(np.int64)
    frags = total[rows].astype(np.int64)[:, None]
    shorter = np.minimum(lengths[rows].astype(np.int64)[:, None], lengths[cols].astype(np.int64)[None, :])
    reported = (frags > 0) & (m > 0) & (m * int(fragsize) >= float(minmatch) * shorter)
    with np.errstate(invalid="ignore", divide="ignore"):
        ani = fastani_mean(ident_sum[np.ix_(rows, cols - col0)], m)
        cov = np.where(reported, m / np.maximum(frags, 1), np.nan)
    identity = np.where(reported, 0.01 * round_sig6(np.where(reported, ani, np.nan)), np.nan)  # fastani.py:113: 0.01 * float(text)
    aln = np.where(reported, int(fragsize) * m, 0)
    err = np.where(reported, frags - m, 0)
    return identity, aln.astype(np.int64), err.astype(np.int64), cov, ~reported


def compute_fastani_hip(  # noqa: PLR0913
    logger: logging.Logger,
    tmp_dir: Path,  # noqa: ARG001
    session,
    run,
    json_filename: Path,
    fasta_dir: Path,
    hash_to_filename: dict[str, str],
    filename_to_hash: dict[str, str],  # noqa: ARG001
    query_hashes: dict[str, int],
    subject_hash: str,
    *,
    cache: Path = Path(),  # noqa: ARG001
    engine=None,
    subject_range: tuple[int, int] | None = None,
    query_batch: int = QUERY_BATCH,
    on_block=None,
) -> int:
    """Run many-vs-subject (all-vs-all when ``subject_hash == ""``) and log the column(s) to JSON.

    A single subject column maps the queries against that one reference genome only (``ref_range``), as one
    ``fastANI -r subject`` process does; ``subject_range`` = (c0, c1) with ``subject_hash == ""`` takes the columns
    c0 .. c1-1 of the sorted genomes (one rank's share of a multi-GPU run, ``rundb.run_fastani_hip``).
    The queries go through in batches of ``query_batch`` genomes -- the reference's 500 (private_cli.py:1029-1033) --
    and the column file grows by one batch after each (natively formatted rows, a complete JSON document every time),
    so an interrupt keeps the finished batches (private_cli.py:1101-1110); the reference index is built once and taken
    over by the later batches.  ``on_block(queries, subjects, identity, aln_length, sim_errors, cov_query, is_null)``
    (optional) receives every batch as arrays as well -- the run driver's direct ingest.
    Library failures end the worker through ``log_sys_exit`` like a failing tool (pyani_plus/utils.py:262-283); a
    failing save returns 2."""
    from .._capi import HipBackendError
    from .sourmash_hip import backend_failure

    configuration = run.configuration
    tool = get_fastani_hip()
    _check_tool_version(logger, tool, configuration)
    fragsize = configuration.fragsize
    if not fragsize:
        log_sys_exit(logger, f"{METHOD} run-id {run.run_id} is missing fragsize parameter")
    kmersize = configuration.kmersize
    if not kmersize:
        log_sys_exit(logger, f"{METHOD} run-id {run.run_id} is missing kmersize parameter")
    minmatch = configuration.minmatch
    if not minmatch:
        log_sys_exit(logger, f"{METHOD} run-id {run.run_id} is missing minmatch parameter")

    queries = sorted(query_hashes)
    if subject_hash:
        subjects = [subject_hash]
    elif subject_range is not None:
        subjects = sorted(hash_to_filename)[subject_range[0] : subject_range[1]]
    else:
        subjects = queries
    genomes = sorted(set(queries) | set(subjects))
    index = {h: i for i, h in enumerate(genomes)}
    sub_idx = [index[s] for s in subjects]
    contiguous = bool(sub_idx) and sub_idx == list(range(sub_idx[0], sub_idx[0] + len(sub_idx)))
    ref_range = (sub_idx[0], sub_idx[0] + len(sub_idx)) if contiguous else None
    rows_done = 0
    try:
        writer = wire.ColumnFileWriter(logger, json_filename, configuration)
    except Exception:
        logger.exception("Unexpected exception saving JSON:")
        return RECORDING_FAILED
    try:
        if engine is None:
            # one process per subject column in the reference's flow: PYANI_HIP_DEVICE=spread deals the columns over the GPUs
            column = sorted(hash_to_filename).index(subject_hash) + 1 if subject_hash in hash_to_filename else 0
            engine = get_engine(spread_key=column)
        arena, dev = load_genomes_for_fragani([Path(fasta_dir) / hash_to_filename[h] for h in genomes], engine)
        lengths = mappable_lengths(arena.contig_len, arena.contig_genome, arena.n_genomes, fragsize)
        n = arena.n_genomes
        # a contiguous range of subject columns (one column in the reference's process layout) comes back as exactly
        # those columns: O(n) numbers in host memory for a column, not an n x n matrix
        width = ref_range[1] - ref_range[0] if ref_range is not None else n
        col0 = ref_range[0] if ref_range is not None else 0
        out = (np.zeros(n, dtype=np.uint32), np.zeros((n, width), dtype=np.uint32), np.zeros((n, width), dtype=np.float64))
        query_idx = [index[q] for q in queries]
        # batches of CONSECUTIVE genome indices, at most `query_batch` queries each: a batch is mapped as a range of query
        # genomes, so it ends where the queries leave a gap (a subject that is not among the queries sits between them)
        batches: list[list[int]] = []
        for i in query_idx:
            if batches and batches[-1][-1] + 1 == i and len(batches[-1]) < max(1, int(query_batch)):
                batches[-1].append(i)
            else:
                batches.append([i])
        for b, batch in enumerate(batches):
            total, matched, ident_sum = engine.fragani(
                dev, arena.contig_start, arena.contig_len, arena.contig_genome, kmersize, fragsize, ref_range=ref_range,
                query_range=(batch[0], batch[-1] + 1), reuse_index=b > 0, out=out, columns_only=ref_range is not None,
            )  # fmt: skip
            ident, aln, sim, cov, null = comparison_block(total, matched, ident_sum, lengths, batch, sub_idx, fragsize, minmatch, col0=col0)
            if on_block is not None:
                on_block([genomes[i] for i in batch], subjects, ident, aln, sim, cov, null)
            try:  # the column file grows by one batch of queries (a complete JSON document after each)
                writer.append([genomes[i] for i in batch], subjects, ident, cov, null, aln_length=aln, sim_errors=sim)
            except Exception:
                logger.exception("Unexpected exception saving JSON:")
                return RECORDING_FAILED
            rows_done = writer.rows
    except KeyboardInterrupt:
        logger.error("Interrupted with %d completed %s comparisons", rows_done, METHOD)  # noqa: TRY400
        run.status = "Worker interrupted"
        session.commit()
    except HipBackendError as err:
        backend_failure(logger, f"{METHOD} comparison", err)
    except ValueError as err:  # an input file that does not load: what a failing fastANI process is to the reference
        log_sys_exit(logger, f"{METHOD} comparison failed: {err}")
    return 0
