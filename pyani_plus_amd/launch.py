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

import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_devices() -> int:
    """HIP devices a child would see, counted without initialising the GPU in this process."""
    import torch

    return int(torch.cuda.device_count())


def choose_backend(world: int) -> str:
    """``nccl`` (RCCL over xGMI) when every rank gets a GPU of its own; ``gloo`` (collectives on host copies, ranks
    sharing devices) otherwise -- a plumbing mode for boxes with fewer GPUs than ranks, and what the CPU tests use.
    ``PYANI_HIP_DIST_BACKEND`` overrides."""
    forced = os.environ.get("PYANI_HIP_DIST_BACKEND", "").strip().lower()
    if forced:
        return forced
    return "nccl" if visible_devices() >= world else "gloo"


def launch_workers(world: int, spec: dict, work_dir: Path, *, timeout: float | None = None, poll: float = 0.2) -> list[dict]:
    """Run ``world`` workers on ``spec`` (written to ``work_dir/spec.json``); returns each rank's result dict.

    A worker reports through ``work_dir/result_rank<r>.json`` (``{"ok": true, ...}`` or ``{"ok": false, "error": msg}``).
    When one fails or dies the others are ended (by handle) -- they would otherwise wait in a collective for ever --
    and ``RuntimeError`` carries the first failure's message."""
    work_dir = Path(work_dir)
    work_dir.mkdir(parents=True, exist_ok=True)
    spec = dict(spec)
    spec.setdefault("backend", choose_backend(world))
    spec_file = work_dir / "spec.json"
    spec_file.write_text(json.dumps(spec))
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))  # fmt: skip
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        root = str(Path(__file__).resolve().parent.parent)
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        log = (work_dir / f"worker_rank{rank}.log").open("w")
        procs.append((subprocess.Popen([sys.executable, "-m", "pyani_plus_amd.worker", str(spec_file)], env=env, stdout=log,
                                       stderr=subprocess.STDOUT), log))  # fmt: skip
    t0 = time.monotonic()
    failed = None
    while True:
        codes = [p.poll() for p, _ in procs]
        if all(c is not None for c in codes):
            break
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad and failed is None:
            failed = bad[0]
            deadline = time.monotonic() + 5.0  # let the others report the failure themselves if they can
        if failed is not None and time.monotonic() > deadline or (timeout is not None and time.monotonic() - t0 > timeout):
            for p, _ in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(poll)
    for _, log in procs:
        log.close()
    results = []
    for rank, (p, _) in enumerate(procs):
        rfile = work_dir / f"result_rank{rank}.json"
        res = json.loads(rfile.read_text()) if rfile.is_file() else {"ok": False, "error": None}
        res["returncode"] = p.returncode
        results.append(res)
    errors = [r for r in results if not r.get("ok") or r["returncode"] != 0]
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
