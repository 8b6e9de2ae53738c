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
