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
