"""The reference's JSON column-file wire format (SURVEY.md section 8b, row "JSON wire format").

Mirrors ``export_json_db_entries`` (pyani_plus/private_cli.py:454-504): one object
with ``configuration`` (8 fields), ``uname`` (3 fields) and ``comparisons``; the keys
``configuration_id`` and ``uname_*`` are stripped from each comparison.
"""

from __future__ import annotations

import json
import logging
import platform
from pathlib import Path

CONFIG_FIELDS = ("method", "program", "version", "fragsize", "mode", "kmersize", "minmatch", "extra")
UNWANTED_KEYS = frozenset({"configuration_id", "uname_system", "uname_release", "uname_machine"})


def configuration_dict(configuration) -> dict:
    return {name: getattr(configuration, name) for name in CONFIG_FIELDS}


def export_json_db_entries(logger: logging.Logger, json_filename: Path, configuration, db_entries: list[dict]) -> None:
    """Serialise comparison entries to the JSON file the reference's importer reads."""
    uname = platform.uname()
    serialised = json.dumps(
        {
            "configuration": configuration_dict(configuration),
            "uname": {"system": uname.system, "release": uname.release, "machine": uname.machine},
            "comparisons": [{k: v for (k, v) in entry.items() if k not in UNWANTED_KEYS} for entry in db_entries],
        }
    )
    with Path(json_filename).open("w") as handle:
        handle.write(serialised)
    logger.debug("Saved %d comparisons to %s", len(db_entries), json_filename)


def export_json_matrices(
    logger: logging.Logger, json_filename: Path, configuration, queries: list[str], subjects: list[str], identity, cov_query, is_null
) -> None:
    """Same file as ``export_json_db_entries`` for a dense query x subject block, written by the
    native bulk writer (``pa_write_comparisons_json``) instead of one Python dict per comparison."""
    import ctypes as C

    import numpy as np

    from . import _capi

    lib = _capi.load_library()
    uname = platform.uname()
    shell = json.dumps(
        {
            "configuration": configuration_dict(configuration),
            "uname": {"system": uname.system, "release": uname.release, "machine": uname.machine},
            "comparisons": [],
        }
    )
    assert shell.endswith("[]}")
    prefix, suffix = shell[:-2], "]}"
    identity = np.ascontiguousarray(identity, dtype=np.float64)
    cov_query = np.ascontiguousarray(cov_query, dtype=np.float64)
    null = np.ascontiguousarray(is_null, dtype=np.uint8)
    nq, ns = len(queries), len(subjects)
    assert identity.shape == (nq, ns) == cov_query.shape == null.shape
    q_arr = (C.c_char_p * max(nq, 1))(*[q.encode() for q in queries])
    s_arr = (C.c_char_p * max(ns, 1))(*[s.encode() for s in subjects])
    _capi.check(
        lib.pa_write_comparisons_json(
            str(json_filename).encode(), prefix.encode(), suffix.encode(), q_arr, nq, s_arr, ns,
            identity.ctypes.data, cov_query.ctypes.data, null.ctypes.data,
        ),  # fmt: skip
        "pa_write_comparisons_json",
    )
    logger.debug("Saved %d comparisons to %s", nq * ns, json_filename)


def save_tile(path: Path, configuration, queries: list[str], subjects: list[str], identity, cov_query, is_null, *,
              aln_length=None, sim_errors=None) -> None:
    """One subject tile as a binary column file: the JSON form costs ~170 bytes and two
    float-to-text conversions per comparison, 17 GB at N = 10^4; this is 17 bytes per comparison and no text.
    ``aln_length`` / ``sim_errors`` (int64, both or neither): the fastANI worker's proxy columns."""
    import numpy as np

    extra = {}
    if aln_length is not None:
        extra = {"aln_length": np.asarray(aln_length, dtype=np.int64), "sim_errors": np.asarray(sim_errors, dtype=np.int64)}
    # written under another name and renamed: a reader (the parent, after an interrupt) never sees half a file
    path = Path(path)
    tmp = path.with_name("." + path.name + ".part.npz")  # (a name no reader's pattern matches)
    np.savez(
        tmp,
        configuration=np.array(json.dumps(configuration_dict(configuration))),
        queries=np.array(queries),
        subjects=np.array(subjects),
        identity=np.asarray(identity, dtype=np.float64),
        cov_query=np.asarray(cov_query, dtype=np.float64),
        is_null=np.asarray(is_null, dtype=bool),
        **extra,
    )
    tmp.replace(path)


def load_tile(path: Path, *, with_proxies: bool = False):
    """(configuration, queries, subjects, identity, cov_query, is_null) of a tile file; with ``with_proxies`` also
    (aln_length, sim_errors), None for a tile that has none."""
    import numpy as np

    with np.load(path) as data:
        out = (json.loads(str(data["configuration"])), [str(x) for x in data["queries"]], [str(x) for x in data["subjects"]],
               data["identity"], data["cov_query"], data["is_null"])  # fmt: skip
        if with_proxies:
            out += ((data["aln_length"], data["sim_errors"]) if "aln_length" in data else (None, None))
        return out


class ColumnFileWriter:
    """The column file written progressively, one block of comparisons at a time.

    The reference re-dumps its whole list after every 100 000 rows so that an interrupted worker leaves the
    completed comparisons behind (pyani_plus/private_cli.py:1863-1894).  Here the file is created with an empty
    ``comparisons`` list and every ``append`` moves the closing ``]}`` back by one block (native
    ``pa_append_comparisons_json``): the file is a complete JSON document after every call, and no row is
    formatted twice."""

    SUFFIX = "]}"

    def __init__(self, logger: logging.Logger, json_filename: Path, configuration) -> None:
        self.logger = logger
        self.path = Path(json_filename)
        self.rows = 0
        export_json_matrices(logger, self.path, configuration, [], [], _empty(), _empty(), _empty(bool))

    def append(self, queries: list[str], subjects: list[str], identity, cov_query, is_null, *, aln_length=None, sim_errors=None) -> None:
        """One more block of comparisons (query-major).  ``aln_length`` / ``sim_errors`` (int64 matrices, both or neither):
        the rows then carry the fastANI worker's six keys (pyani_plus/private_cli.py:1066-1080)."""
        import ctypes as C

        import numpy as np

        from . import _capi

        nq, ns = len(queries), len(subjects)
        if nq == 0 or ns == 0:
            return
        lib = _capi.load_library()
        identity = np.ascontiguousarray(identity, dtype=np.float64)
        cov_query = np.ascontiguousarray(cov_query, dtype=np.float64)
        null = np.ascontiguousarray(is_null, dtype=np.uint8)
        assert identity.shape == (nq, ns) == cov_query.shape == null.shape
        q_arr = (C.c_char_p * nq)(*[q.encode() for q in queries])
        s_arr = (C.c_char_p * ns)(*[s.encode() for s in subjects])
        if aln_length is not None:
            aln = np.ascontiguousarray(aln_length, dtype=np.int64)
            err = np.ascontiguousarray(sim_errors, dtype=np.int64)
            assert aln.shape == (nq, ns) == err.shape
            _capi.check(
                lib.pa_append_comparisons_json_ex(
                    str(self.path).encode(), self.SUFFIX.encode(), int(self.rows > 0), q_arr, nq, s_arr, ns,
                    identity.ctypes.data, cov_query.ctypes.data, null.ctypes.data, aln.ctypes.data, err.ctypes.data,
                ),  # fmt: skip
                "pa_append_comparisons_json_ex",
            )
        else:
            _capi.check(
                lib.pa_append_comparisons_json(
                    str(self.path).encode(), self.SUFFIX.encode(), int(self.rows > 0), q_arr, nq, s_arr, ns,
                    identity.ctypes.data, cov_query.ctypes.data, null.ctypes.data,
                ),  # fmt: skip
                "pa_append_comparisons_json",
            )
        self.rows += nq * ns
        self.logger.debug("Saved %d comparisons to %s", self.rows, self.path)


def _empty(dtype=float):
    import numpy as np

    return np.zeros((0, 0), dtype=dtype)


def load_json_comparisons(json_filename: Path) -> dict:
    """Parse a column file and check the fields ``import_json_comparisons`` requires
    (pyani_plus/private_cli.py:555-605)."""
    data = json.loads(Path(json_filename).read_text())
    for key in ("configuration", "uname", "comparisons"):
        if key not in data:
            msg = f"JSON file {json_filename} is missing key {key!r}"
            raise ValueError(msg)
    for entry in data["comparisons"]:
        for key in ("query_hash", "subject_hash", "identity"):
            if key not in entry:
                msg = f"JSON file {json_filename} has a comparison without {key!r}"
                raise ValueError(msg)
    return data


MANYSEARCH_COLUMNS = (
    "query_name,query_md5,match_name,containment,intersect_hashes,ksize,scaled,moltype,match_md5,jaccard,"
    "max_containment,query_containment_ani,match_containment_ani,average_containment_ani,max_containment_ani"
)


def _rust_float(x: float) -> str:
    """Shortest round-trip decimal the way Rust's ``Display`` prints an f64 (never an exponent)."""
    text = repr(float(x))
    if "e" not in text:
        return text
    from decimal import Decimal

    fixed = format(Decimal(text), "f")
    return fixed if "." in fixed else fixed + ".0"


def export_manysearch_csv(  # noqa: PLR0913
    csv_filename: Path,
    query_names: list[str],
    query_sig_md5: list[str],
    subject_names: list[str],
    subject_sig_md5: list[str],
    counts,
    query_sizes,
    subject_sizes,
    kmersize: int,
    scaled: int,
) -> int:
    """Write a tile's intersection counts as the CSV ``sourmash scripts manysearch`` would write.

    Same 15 columns and header as the reference's intermediate file (column list pinned by
    tests/fixtures/viral_example/intermediates/sourmash/manysearch.csv:1, read by name in
    pyani_plus/methods/sourmash.py:107-110); pairs without a shared hash get no row, exactly as
    upstream.  Row order is query-major (upstream's is thread-dependent,
    tests/test_public_cli.py:1055-1057).  Returns the number of rows written.  This is an export for
    people who post-process that file; the method itself never parses it back.
    """
    import numpy as np

    from pyani_plus_amd.engine import ani_host

    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    q_sizes = np.asarray(query_sizes, dtype=np.uint64)
    s_sizes = np.asarray(subject_sizes, dtype=np.uint64)
    _ident, q_ani, _null = ani_host(counts, q_sizes, s_sizes, kmersize)
    _ident_t, m_ani_t, _null_t = ani_host(np.ascontiguousarray(counts.T), s_sizes, q_sizes, kmersize)
    rows = 0
    with Path(csv_filename).open("w") as handle:
        handle.write(MANYSEARCH_COLUMNS + "\n")
        for q, s in zip(*np.nonzero(counts)):
            shared = int(counts[q, s])
            nq, ns = int(q_sizes[q]), int(s_sizes[s])
            qa, ma = float(q_ani[q, s]), float(m_ani_t[s, q])
            fields = (
                query_names[q],
                query_sig_md5[q],
                subject_names[s],
                _rust_float(shared / nq),
                str(shared),
                str(kmersize),
                str(scaled),
                "DNA",
                subject_sig_md5[s],
                _rust_float(shared / (nq + ns - shared)),
                _rust_float(shared / min(nq, ns)),
                _rust_float(qa),
                _rust_float(ma),
                _rust_float((qa + ma) / 2.0),
                _rust_float(max(qa, ma)),
            )
            handle.write(",".join(fields) + "\n")
            rows += 1
    return rows
