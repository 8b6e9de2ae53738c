"""Apply the edits INTEGRATION.md lists to a scratch copy of the reference package and run ITS worker commands.

Run by tests/test_reference_interop.py in a process of its own (this container only):
    python tests/tools/patched_reference_worker.py <scratch dir> <fixture dir> <scaled>
The copy lives under the scratch directory given on the command line (never in this repository, never in
/root/reference); the device is replaced by the oracle-backed test engine -- what is under test is that the
reference's `prepare-genomes` and `compute-column --subject 0` find, call and are satisfied by the method module.
Prints one JSON line with what the reference's importer and matrix cache made of the column file.
"""
import datetime
import json
import shutil
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent.parent
REFERENCE = Path("/root/reference")
scratch, fixture, scaled = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3])
sys.dont_write_bytecode = True
sys.path.insert(0, str(REPO))
if not hasattr(datetime, "UTC"):
    datetime.UTC = datetime.timezone.utc  # the reference wants Python >= 3.11

pkg = scratch / "pyani_plus"
shutil.copytree(REFERENCE / "pyani_plus", pkg, ignore=shutil.ignore_patterns("__pycache__"))
integration = (REPO / "INTEGRATION.md").read_text()

# 2. the method module: the re-export printed in INTEGRATION.md section 2
start = integration.index('"""Code to implement the sourmash-hip')
module_text = integration[start : integration.index("```", start)]
(pkg / "methods" / "sourmash_hip.py").write_text(module_text)

# 3. private_cli.py: all-columns mode and the method -> worker dict
cli = (pkg / "private_cli.py").read_text()
for old, new in (
    ('if method == "sourmash":\n                        subject_hash = ""', 'if method in {"sourmash", "sourmash-hip"}:\n                        subject_hash = ""'),
    ('"sourmash": compute_sourmash,\n', '"sourmash": compute_sourmash,\n                    "sourmash-hip": compute_sourmash_hip,\n'),
    ("\napp = typer.Typer(", "\nfrom pyani_plus.methods.sourmash_hip import compute_sourmash_hip  # noqa: E402\n\napp = typer.Typer("),
):
    assert cli.count(old) == 1, old
    cli = cli.replace(old, new)
(pkg / "private_cli.py").write_text(cli)

# 4. tools.py: program / version of the backend
tools_text = (pkg / "tools.py").read_text() + '''

def get_sourmash_hip() -> ExternalToolData:
    from pyani_plus_amd.methods import sourmash_hip
    t = sourmash_hip.get_sourmash_hip()
    return ExternalToolData(t.exe_path, t.version)
'''
(pkg / "tools.py").write_text(tools_text)

# 5. the fastANI-hip method: module, dict entry, tool
(pkg / "methods" / "fastani_hip.py").write_text(
    '"""Code to implement the fastANI-hip (MI355X) Average Nucleotide Identity (ANI) method."""\n'
    "from pyani_plus_amd.methods.fastani_hip import (  # noqa: F401\n"
    "    FRAG_LEN, KMER_SIZE, MIN_FRACTION, compute_fastani_hip, get_fastani_hip,\n)\n"
)
cli = (pkg / "private_cli.py").read_text()
for old, new in (
    ('"fastANI": compute_fastani,\n', '"fastANI": compute_fastani,\n                    "fastANI-hip": compute_fastani_hip,\n'),
    ("\napp = typer.Typer(", "\nfrom pyani_plus.methods.fastani_hip import compute_fastani_hip  # noqa: E402\n\napp = typer.Typer("),
):
    assert cli.count(old) == 1, old
    cli = cli.replace(old, new)
(pkg / "private_cli.py").write_text(cli)
(pkg / "tools.py").write_text((pkg / "tools.py").read_text() + '''

def get_fastani_hip() -> ExternalToolData:
    from pyani_plus_amd.methods import fastani_hip
    t = fastani_hip.get_fastani_hip()
    return ExternalToolData(t.exe_path, t.version)
''')

sys.path.insert(0, str(scratch))
from pyani_plus import db_orm, private_cli, tools  # noqa: E402  (the patched copy)

assert Path(private_cli.__file__).is_relative_to(scratch)
from pyani_plus_amd.methods import sourmash_hip  # noqa: E402
from tests.fake_engine import OracleEngine  # noqa: E402

from pyani_plus_amd.methods import fastani_hip  # noqa: E402

sourmash_hip.get_engine = lambda *_a, **_k: OracleEngine()  # no GPU here: the boundary is under test, not the arithmetic
fastani_hip.get_engine = sourmash_hip.get_engine

database, cache, json_file = scratch / "run.sqlite", scratch / "cache", scratch / "column_0.json"
cache.mkdir()
tool = tools.get_sourmash_hip()
private_cli.log_run(
    fasta=fixture, database=database, cmdline="pyani-plus sourmash-hip ...", status="Initialising", name="patched reference",
    method="sourmash-hip", program=tool.exe_path.stem, version=tool.version, kmersize=31, extra=f"scaled={scaled}", create_db=True,
)
assert private_cli.prepare_genomes(database=database, run_id=1, cache=cache) == 0
status = private_cli.compute_column(database=database, run_id=1, subject="0", json=json_file, cache=cache, temp=Path("-"), log=scratch / "worker.log")
import logging  # noqa: E402

logger = logging.getLogger("patched")
with db_orm.connect_to_db(logger, database) as session:
    private_cli.import_json_comparisons(logger, session, json_file)
    run = db_orm.load_run(session, run_id=1, check_complete=True)
    run.cache_comparisons()
    session.commit()
    out = {
        "compute_column": status, "comparisons": run.comparisons().count(), "genomes": run.genomes.count(),
        "df_identity": run.df_identity, "df_cov_query": run.df_cov_query, "df_hadamard": run.df_hadamard,
        "method": run.configuration.method, "program": run.configuration.program,
        "signatures": len(list((cache / f"sourmash_k=31_scaled={scaled}").glob("*.sig"))),
    }

# ---- fastANI-hip, one subject column per worker call as the unpatched reference schedules it (only on request:
# the oracle maps every pair on the CPU)
if len(sys.argv) > 4 and sys.argv[4] == "fastani":
    database2 = scratch / "fastani.sqlite"
    ftool = tools.get_fastani_hip()
    private_cli.log_run(
        fasta=fixture, database=database2, cmdline="pyani-plus fastANI-hip ...", status="Initialising", name="patched reference, fastANI-hip",
        method="fastANI-hip", program=ftool.exe_path.stem, version=ftool.version, fragsize=fastani_hip.FRAG_LEN,
        kmersize=fastani_hip.KMER_SIZE, minmatch=fastani_hip.MIN_FRACTION, create_db=True,
    )
    assert private_cli.prepare_genomes(database=database2, run_id=1, cache=cache) == 0  # nothing to prepare for this method
    codes = []
    with db_orm.connect_to_db(logger, database2) as session:
        n_columns = db_orm.load_run(session, run_id=1).genomes.count()
    for column in range(1, n_columns + 1):
        column_file = scratch / f"fastani_column_{column}.json"
        codes.append(private_cli.compute_column(database=database2, run_id=1, subject=str(column), json=column_file, cache=cache,
                                                temp=Path("-"), log=scratch / "worker.log"))
        with db_orm.connect_to_db(logger, database2) as session:
            private_cli.import_json_comparisons(logger, session, column_file)
    with db_orm.connect_to_db(logger, database2) as session:
        run = db_orm.load_run(session, run_id=1, check_complete=True)
        run.cache_comparisons()
        session.commit()
        out["fastani"] = {"codes": codes, "comparisons": run.comparisons().count(), "df_identity": run.df_identity,
                          "df_aln_length": run.df_aln_length, "df_sim_errors": run.df_sim_errors, "df_cov_query": run.df_cov_query}
print(json.dumps(out))
