"""Real signals to real processes, after the reference's own fault injection (/root/reference/tests/test_interrupt.py:61-134:
``compute-column`` as a subprocess, SIGINT after a while, return code 0, partial JSON, ``run.status == "Worker
interrupted"``).  Here for the build's own run driver, ``python -m pyani_plus_amd.rundb fastani --gpus 2`` (worker
processes, the oracle-backed engine standing in for the GPUs):

* SIGTERM -- what ``scancel`` sends -- to ONE rank after its first query batch: that rank keeps its finished batches and
  reports the interrupt, the other rank finishes its columns, the driver exits with code 0, the run is marked and
  partial, nobody is left running;
* SIGINT to the PARENT: passed on to both ranks, same outcome;
* ``resume`` then completes the run, and the database equals an uninterrupted one.
"""

from __future__ import annotations

import json
import os
import signal
import sqlite3
import subprocess
import sys
import time
from pathlib import Path

import pytest

from pyani_plus_amd import rundb
from tests.fake_engine import OracleEngine
from tests.helpers import GOLDEN

ROOT = Path(__file__).resolve().parent.parent
FACTORY = "tests.fake_engine:SlowOracleEngine"


def _alive(pid: int) -> bool:
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    # a zombie still answers kill(0): look at its state
    try:
        return Path(f"/proc/{pid}/stat").read_text().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def _start_driver(tmp_path: Path) -> tuple[subprocess.Popen, Path, Path]:
    env = dict(os.environ)
    env["PYTHONPATH"] = str(ROOT) + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    env["PYANI_HIP_DIST_BACKEND"] = "gloo"
    db, temp = tmp_path / "run.sqlite", tmp_path / "temp"
    proc = subprocess.Popen(
        [sys.executable, "-m", "pyani_plus_amd.rundb", "fastani", str(GOLDEN / "viral_example"), "-d", str(db), "--temp", str(temp),
         "--gpus", "2", "--query-batch", "1", "--engine-factory", FACTORY],
        env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
    )  # fmt: skip
    return proc, db, temp


def _wait_for_first_batch(work_dir: Path, proc: subprocess.Popen, timeout: float = 120.0) -> tuple[int, Path]:
    """(rank, its column file) of the first rank whose column file holds a finished query batch"""
    t0 = time.monotonic()
    while time.monotonic() - t0 < timeout:
        assert proc.poll() is None, proc.stdout.read()
        for path in sorted(work_dir.glob("fastANI-hip.run_*.columns_*.json")):
            try:
                rows = json.loads(path.read_text())["comparisons"]
            except (ValueError, KeyError):
                continue
            if rows:
                pids = json.loads((work_dir / "pids.json").read_text())
                spec = json.loads((work_dir / "spec.json").read_text())
                first = int(path.name.split("columns_")[1].split("_")[0]) - 1
                rank = [r for r, (c0, c1) in enumerate(spec["column_ranges"]) if c0 == first and c1 > c0][0]
                return rank, path, pids
        time.sleep(0.05)
    raise AssertionError("no query batch finished in time")


def _status_and_rows(db: Path) -> tuple[str, int]:
    conn = sqlite3.connect(db)
    status = conn.execute("SELECT status FROM runs").fetchone()[0]
    rows = conn.execute("SELECT COUNT(*) FROM comparisons").fetchone()[0]
    conn.close()
    return status, rows


def _check_partial_then_resume(tmp_path: Path, db: Path, pids: list[int], partial_file: Path) -> None:
    assert not any(_alive(pid) for pid in pids), "a worker outlived the driver"
    status, rows = _status_and_rows(db)
    assert status == "Worker interrupted"
    assert 0 < rows < 9, rows  # the finished batches are recorded, the rest is not
    kept = json.loads(partial_file.read_text())["comparisons"]  # a complete JSON document holding whole batches
    assert kept and len(kept) < 9
    # resume completes it, and nothing distinguishes the result from an uninterrupted run
    run = rundb.resume(db, temp=tmp_path / "resume", engine=OracleEngine())
    assert run.status == "Done"
    whole = rundb.run_fastani_hip(GOLDEN / "viral_example", tmp_path / "whole.sqlite", engine=OracleEngine(), temp=tmp_path / "w")
    assert whole.status == "Done"

    def table(path):
        conn = sqlite3.connect(path)
        out = conn.execute("SELECT query_hash, subject_hash, identity, aln_length, sim_errors, cov_query FROM comparisons ORDER BY 1, 2").fetchall()
        conn.close()
        return out

    assert table(db) == table(tmp_path / "whole.sqlite") and len(table(db)) == 9


@pytest.mark.timeout(600)
def test_sigterm_to_one_rank_keeps_its_finished_batches(tmp_path):
    proc, db, temp = _start_driver(tmp_path)
    try:
        work_dir = temp / "fastANI-hip.run_1.workers"
        t0 = time.monotonic()
        while not (work_dir / "pids.json").is_file():
            assert proc.poll() is None and time.monotonic() - t0 < 120, proc.stdout.read() if proc.poll() is not None else "no workers"
            time.sleep(0.05)
        rank, partial_file, pids = _wait_for_first_batch(work_dir, proc)
        os.kill(pids[rank], signal.SIGTERM)  # what a scheduler sends
        out, _ = proc.communicate(timeout=300)
    finally:
        if proc.poll() is None:
            proc.kill()
    assert proc.returncode == 0, out
    result = json.loads((work_dir / f"result_rank{rank}.json").read_text())
    assert result["ok"] and result["interrupted"]
    other = json.loads((work_dir / f"result_rank{1 - rank}.json").read_text())
    assert other["ok"] and not other.get("interrupted")  # no exchange between fragment-ANI ranks: it ran to its end
    assert "Interrupted with" in (work_dir / f"worker_rank{rank}.log").read_text()
    _check_partial_then_resume(tmp_path, db, pids, partial_file)


@pytest.mark.timeout(600)
def test_sigint_to_the_parent_reaches_every_rank(tmp_path):
    proc, db, temp = _start_driver(tmp_path)
    try:
        work_dir = temp / "fastANI-hip.run_1.workers"
        t0 = time.monotonic()
        while not (work_dir / "pids.json").is_file():
            assert proc.poll() is None and time.monotonic() - t0 < 120, proc.stdout.read() if proc.poll() is not None else "no workers"
            time.sleep(0.05)
        _rank, partial_file, pids = _wait_for_first_batch(work_dir, proc)
        proc.send_signal(signal.SIGINT)
        out, _ = proc.communicate(timeout=300)
    finally:
        if proc.poll() is None:
            proc.kill()
    assert proc.returncode == 0, out
    results = [json.loads((work_dir / f"result_rank{r}.json").read_text()) for r in range(2)]
    assert all(r["ok"] for r in results) and any(r.get("interrupted") for r in results)
    _check_partial_then_resume(tmp_path, db, pids, partial_file)


def test_workers_are_ended_when_the_launch_fails_midway(tmp_path, monkeypatch):
    """Any exception in the parent between starting the workers and collecting them ends the children on the way out."""
    from pyani_plus_amd import launch

    monkeypatch.setenv("PYANI_HIP_DIST_BACKEND", "gloo")
    real_sleep = time.sleep
    calls = {"n": 0}

    def failing_sleep(seconds):
        calls["n"] += 1
        if calls["n"] == 3:
            raise RuntimeError("the parent stumbles")
        real_sleep(seconds)

    monkeypatch.setattr(launch.time, "sleep", failing_sleep)
    spec = {"task": "fastani", "run_id": 1, "fasta_dir": str(GOLDEN / "viral_example"), "hash_to_filename": {}, "query_hashes": {},
            "column_ranges": [(0, 1), (1, 2)], "work_dir": str(tmp_path / "w"), "tiles": False, "engine_factory": FACTORY,
            "configuration": {"method": "fastANI-hip", "program": "x", "version": "0", "fragsize": 3000, "mode": None, "kmersize": 16,
                              "minmatch": 0.2, "extra": None, "configuration_id": 1}}  # fmt: skip
    with pytest.raises(RuntimeError, match="the parent stumbles"):
        launch.launch_workers(2, spec, tmp_path / "w")
    pids = json.loads((tmp_path / "w" / "pids.json").read_text())
    real_sleep(0.2)
    assert not any(_alive(pid) for pid in pids)


def test_worker_notices_a_parent_that_is_already_gone():
    """The worker arms PR_SET_PDEATHSIG itself, first thing (no Python between fork and exec in the parent), and compares its
    parent with the pid the launcher put in the environment: a parent that died in between is noticed at once."""
    import os
    import subprocess
    import sys

    from pyani_plus_amd import launch

    root = str(Path(__file__).resolve().parent.parent)
    code = "from pyani_plus_amd.launch import die_with_parent; die_with_parent(); print('armed')"
    env = {**os.environ, "PYTHONPATH": root}
    ok = subprocess.run([sys.executable, "-c", code], env={**env, launch.PARENT_PID_ENV: str(os.getpid())}, capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0 and "armed" in ok.stdout
    gone = subprocess.run([sys.executable, "-c", code], env={**env, launch.PARENT_PID_ENV: "1"}, capture_output=True, text=True, timeout=60)
    assert gone.returncode != 0 and "armed" not in gone.stdout and "gone" in gone.stderr
