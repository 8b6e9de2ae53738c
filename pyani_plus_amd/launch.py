"""Start one worker process per GPU of this node and wait for them.

The reference spreads a run over processes -- one ``compute-column`` worker per subject column, started by
snakemake (pyani_plus/public_cli.py:236-261, pyani_plus/private_cli.py:757-973) -- and exchanges results through
files on a shared directory (pyani_plus/workflows/__init__.py:71-109).  The MI355X counterpart is one worker per
GPU: ``launch_workers`` starts ``python -m pyani_plus_amd.worker <spec.json>`` N times with
``RANK``/``LOCAL_RANK``/``WORLD_SIZE``/``MASTER_ADDR``/``MASTER_PORT`` set, as ``torch.distributed.run`` would.

The calling process must not have initialised the GPU (this module imports neither torch nor the HIP library):
on the MI355X pool a process that has touched the GPU must not fork or exec workers.
"""

from __future__ import annotations

import contextlib
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_DEVICE_COUNT: int | None = None


def visible_devices() -> int:
    """HIP devices a worker would see.  Counted in a short-lived child process (once; the answer is kept): asking the
    runtime here -- ``torch.cuda.device_count()`` falls back to ``hipGetDeviceCount`` -- would initialise the GPU in the
    very process that goes on to start the workers, which this module promises not to do."""
    global _DEVICE_COUNT
    if _DEVICE_COUNT is None:
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                                 timeout=300, check=False)  # fmt: skip
            _DEVICE_COUNT = int(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else 0
        except (OSError, ValueError, subprocess.TimeoutExpired):
            _DEVICE_COUNT = 0
    return _DEVICE_COUNT


@contextlib.contextmanager
def signals_as_interrupt():
    """SIGINT and SIGTERM (``scancel``, ``kill``) both raise ``KeyboardInterrupt`` while the block runs, as the reference's
    worker arranges for itself (pyani_plus/private_cli.py:816-823): the code that flushes partial results on Ctrl-C then
    also runs when a scheduler ends the job.  Main thread only (elsewhere the block just runs)."""
    if threading.current_thread() is not threading.main_thread():
        yield
        return
    previous = {sig: signal.signal(sig, signal.default_int_handler) for sig in (signal.SIGINT, signal.SIGTERM)}
    try:
        yield
    finally:
        for sig, handler in previous.items():
            signal.signal(sig, handler)


PARENT_PID_ENV = "PYANI_HIP_PARENT_PID"


def die_with_parent() -> None:
    """Called by the worker itself, first thing in ``worker.main`` (no Python runs between fork and exec in the parent, which
    may have threads): the worker gets SIGTERM when the process that started it goes away, however that happens
    (``kill -9`` included) -- no rank is left computing on a GPU for a parent that no longer exists.  The parent's pid
    comes in the environment: a parent that died before this call is noticed right here."""
    import ctypes

    try:
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)  # PR_SET_PDEATHSIG
    except (OSError, AttributeError):
        pass
    want = os.environ.get(PARENT_PID_ENV)
    if want and want.isdigit() and os.getppid() != int(want):
        raise SystemExit("the process that started this worker is gone")


def choose_backend(world: int) -> str:
    """``nccl`` (RCCL over xGMI) when every rank gets a GPU of its own; ``gloo`` (collectives on host copies, ranks
    sharing devices) otherwise -- a plumbing mode for boxes with fewer GPUs than ranks, and what the CPU tests use.
    ``PYANI_HIP_DIST_BACKEND`` overrides."""
    forced = os.environ.get("PYANI_HIP_DIST_BACKEND", "").strip().lower()
    if forced:
        return forced
    return "nccl" if visible_devices() >= world else "gloo"


def launch_workers(world: int, spec: dict, work_dir: Path, *, timeout: float | None = None, poll: float = 0.2, grace: float = 60.0) -> list[dict]:
    """Run ``world`` workers on ``spec`` (written to ``work_dir/spec.json``); returns each rank's result dict.

    A worker reports through ``work_dir/result_rank<r>.json`` (``{"ok": true, ...}`` or ``{"ok": false, "error": msg}``);
    its process id is in ``work_dir/pids.json``.  The workers are fresh children in sessions of their own (a Ctrl-C at
    the terminal reaches this process only), and none outlives this call, whatever way it ends:

    * a rank fails or dies: the others are ended (by handle) -- they would otherwise wait in a collective for ever --
      and ``WorkerFailure`` carries the first failure's message;
    * this process is interrupted (``KeyboardInterrupt``; with ``signals_as_interrupt`` also SIGTERM): the interrupt is
      passed on to the ranks once, they flush what they have and report ``"interrupted": true`` (the reference's worker:
      pyani_plus/private_cli.py:1889-1894), and after ``grace`` seconds whoever is still there is ended; the results
      are returned, not raised -- the caller records the partial run;
    * a rank reports an interrupt of its own (a signal sent to that rank): ranks that exchange data with it
      (``spec["task"] == "sourmash"``) cannot finish and are ended, independent ranks run to their end;
    * any other exception here: every child is ended on the way out."""
    work_dir = Path(work_dir)
    work_dir.mkdir(parents=True, exist_ok=True)
    spec = dict(spec)
    spec.setdefault("backend", choose_backend(world))
    spec_file = work_dir / "spec.json"
    spec_file.write_text(json.dumps(spec))
    for pattern in ("result_rank*.json", "*.tile_*.npz", "*.tile.npz", "*.columns_*.json", "*.part", ".*.part.npz"):  # a resumed run reuses the directory:
        for stale in work_dir.glob(pattern):  # nothing an earlier attempt left may be taken for this one's
            stale.unlink()
    port = free_port()
    procs: list[tuple[subprocess.Popen, object]] = []
    exchanging = spec.get("task") == "sourmash"  # ranks that take part in a collective
    interrupted_here = False
    ended_here: set[int] = set()  # ranks THIS function ended (kill by handle): only those are "ended by the parent"
    try:
        for rank in range(world):
            env = dict(os.environ)
            env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **{PARENT_PID_ENV: str(os.getpid())})  # fmt: skip
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
            root = str(Path(__file__).resolve().parent.parent)
            env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
            log = (work_dir / f"worker_rank{rank}.log").open("w")
            procs.append((subprocess.Popen([sys.executable, "-m", "pyani_plus_amd.worker", str(spec_file)], env=env, stdout=log,
                                           stderr=subprocess.STDOUT, start_new_session=True), log))  # fmt: skip
        (work_dir / "pids.json").write_text(json.dumps([p.pid for p, _ in procs]))
        t0 = time.monotonic()
        deadline = None  # when the ranks still running are ended

        def rank_interrupted(rank: int) -> bool:
            rfile = work_dir / f"result_rank{rank}.json"
            try:
                return bool(json.loads(rfile.read_text()).get("interrupted")) if rfile.is_file() else False
            except ValueError:
                return False

        while True:
            try:
                codes = [p.poll() for p, _ in procs]
                if all(c is not None for c in codes):
                    break
                now = time.monotonic()
                if deadline is None:
                    if any(c not in (None, 0) for c in codes):
                        deadline = now + 5.0  # let the others report the failure themselves if they can
                    elif exchanging and any(c == 0 and rank_interrupted(r) for r, c in enumerate(codes)):
                        deadline = now + 5.0  # its peers wait for it in a collective
                if (deadline is not None and now > deadline) or (timeout is not None and now - t0 > timeout):
                    for rank, (p, _) in enumerate(procs):
                        if p.poll() is None:
                            ended_here.add(rank)
                            p.kill()
                time.sleep(poll)
            except KeyboardInterrupt:
                if interrupted_here:
                    raise  # a second interrupt: give up at once (the children are ended on the way out)
                interrupted_here = True
                for p, _ in procs:
                    if p.poll() is None:
                        p.send_signal(signal.SIGINT)
                deadline = time.monotonic() + grace
    finally:
        for rank, (p, log) in enumerate(procs):
            if p.poll() is None:
                ended_here.add(rank)
                p.kill()
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:  # pragma: no cover - a process that does not die on SIGKILL
                pass
            log.close()
    results = []
    for rank, (p, _) in enumerate(procs):
        rfile = work_dir / f"result_rank{rank}.json"
        try:
            res = json.loads(rfile.read_text()) if rfile.is_file() else {"ok": False, "error": None}
        except ValueError:
            res = {"ok": False, "error": None}
        res["returncode"] = p.returncode
        results.append(res)
    any_interrupt = interrupted_here or any(r.get("interrupted") for r in results)
    if any_interrupt:
        # A rank this function ended while it waited for an interrupted peer (or for too long after the interrupt) and that
        # left no report is part of the interrupt.  A rank that went down by itself at the same moment (a crash, the OOM
        # killer: no report, not ended here) is NOT: it stays a failure below.
        for rank, r in enumerate(results):
            if rank in ended_here and not r.get("ok") and not r.get("error"):
                r.update(ok=True, interrupted=True, ended_by_parent=True)
    # a rank whose report says "interrupted" has done its part whatever its exit code (it may have been ended after the
    # grace period, report written); any other non-zero exit is a failure
    errors = [r for r in results if not r.get("ok") or (r["returncode"] != 0 and not r.get("ended_by_parent") and not r.get("interrupted"))]
    if errors:
        # a rank's own message first; a rank that was killed while waiting for a failed peer has none
        told = [r["error"] for r in errors if r.get("error")]
        if told:
            raise WorkerFailure(told[0])
        tail = ""
        for rank, r in enumerate(results):
            if r["returncode"] != 0:
                text = (work_dir / f"worker_rank{rank}.log").read_text(errors="replace")[-2000:]
                tail = f"rank {rank} exited with code {r['returncode']}:\n{text}"
                break
        raise WorkerFailure(tail or "a worker failed without a message")
    return results


class WorkerFailure(RuntimeError):
    """A rank ended with an error; the message is the rank's own (``log_sys_exit`` text of the worker)."""
