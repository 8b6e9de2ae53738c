#!/usr/bin/env python3
"""Generate golden ``export-run`` files by RUNNING the reference's own ``public_cli.export_run`` (this container only).

For each fixture set a database is made with the reference's plumbing exactly as ``make_boundary_golden.py``
does (``private_cli.log_run`` -> parse the reference's fixture ``manysearch.csv`` -> column JSON ->
``import_json_comparisons`` -> ``cache_comparisons``), then ``pyani_plus.public_cli.export_run`` writes
``sourmash_run_1.tsv`` and the six matrices; only those OUTPUT files are stored, under tests/golden/<set>/export/.

``pyani_plus.public_cli`` imports the whole CLI (snakemake workflow layer, plotting, ANIm); the packages those
need are not installed here and play no part in ``export_run``, so empty stand-in modules satisfy the imports.

    python tests/golden/make_export_golden.py      # needs /root/reference
"""

from __future__ import annotations

import datetime
import logging
import sys
import tempfile
import types
from pathlib import Path

REFERENCE = Path("/root/reference")
HERE = Path(__file__).resolve().parent
SETS = {"viral_example": 300, "bad_alignments": 300, "bacterial_example": 1000}


class _Anything(types.ModuleType):
    """A module whose every attribute exists (import-time stand-in for CLI dependencies export_run never calls)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Anything(f"{self.__name__}.{name}")
        sys.modules[sub.__name__] = sub
        return sub

    def __call__(self, *args, **kwargs):
        return self


def main() -> None:
    if not REFERENCE.is_dir():
        raise SystemExit("the reference checkout is needed to regenerate these vectors")
    sys.dont_write_bytecode = True
    if not hasattr(datetime, "UTC"):
        datetime.UTC = datetime.timezone.utc  # py3.10 shim for db_orm.add_run
    sys.path.insert(0, str(REFERENCE))
    for missing in ("intervaltree", "snakemake", "snakemake.cli", "snakemake.api", "snakemake.settings", "snakemake.settings.types",
                    "snakemake_interface_executor_plugins", "snakemake_interface_executor_plugins.settings", "seaborn", "matplotlib",
                    "matplotlib.pyplot", "matplotlib.colors"):  # fmt: skip
        try:
            __import__(missing)
        except Exception:  # noqa: BLE001
            sys.modules[missing] = _Anything(missing)
    from pyani_plus import db_orm, private_cli, public_cli
    from pyani_plus.methods import sourmash

    logger = logging.getLogger("golden")
    for name, scaled in SETS.items():
        fasta_dir = REFERENCE / "tests/fixtures" / name
        csv = fasta_dir / "intermediates/sourmash/manysearch.csv"
        with tempfile.TemporaryDirectory() as tmp:
            db = Path(tmp) / "golden.db"
            private_cli.log_run(
                fasta=fasta_dir, database=db, cmdline="pyani-plus sourmash ...", status="Testing",
                name=f"golden {name}", method="sourmash", program="sourmash", version="4.8.11",
                kmersize=31, extra=f"scaled={scaled}", create_db=True,
            )  # fmt: skip
            with db_orm.connect_to_db(logger, db) as session:
                run = db_orm.load_run(session, run_id=1)
                hashes = sorted(a.genome_hash for a in run.fasta_hashes)
                config = run.configuration
                entries = [
                    {
                        "query_hash": q, "subject_hash": s, "identity": max_cont, "cov_query": q_cont,
                        "configuration_id": config.configuration_id,
                        "uname_system": "Linux", "uname_release": "x", "uname_machine": "x86_64",
                    }  # fmt: skip
                    for q, s, q_cont, max_cont in sourmash.parse_sourmash_manysearch_csv(
                        logger, csv, {(q, s) for q in hashes for s in hashes}
                    )
                ]
                json_file = Path(tmp) / "sourmash.run_1.column_0.json"
                private_cli.export_json_db_entries(logger, json_file, config, entries)
                private_cli.import_json_comparisons(logger, session, json_file)
                run.cache_comparisons()
                session.commit()
            out = HERE / name / "export"
            out.mkdir(exist_ok=True)
            public_cli.export_run(database=db, outdir=out, run_id=1, label="stem")
        print(name, sorted(p.name for p in out.iterdir()))


if __name__ == "__main__":
    main()
