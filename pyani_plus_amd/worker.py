"""One rank of a multi-GPU run: ``python -m pyani_plus_amd.worker <spec.json>`` (started by ``launch.launch_workers``).

Counterpart of the reference's ``compute-column`` worker process (pyani_plus/private_cli.py:757-973) with one
worker per GPU instead of one per subject column:

* ``sourmash``: the rank loads, checksums and sketches its length-balanced share of the FASTA files, writes their
  ``.sig`` files, takes part in ONE all-gather of the sketches (RCCL over xGMI; ``distributed.allgather_sketches``,
  overlapped with the dictionary build of its own columns) and evaluates all queries against its own genomes as
  subject columns; the result is a binary column-tile file (``wire.save_tile``) the parent ingests.  No other
  data-path collective.
* ``fastani``: reference ranges of ``pa_fragani`` over the ranks -- each rank maps all queries against its own range
  of subject columns and writes the reference's JSON column file for them; no collective at all.

The rank reports through ``result_rank<r>.json`` next to the spec; a failure carries the message the reference
would have ended the worker with (``log_sys_exit``).  SIGINT and SIGTERM both arrive as ``KeyboardInterrupt``
(pyani_plus/private_cli.py:816-823): a fragment-ANI rank keeps the query batches it has finished in its column file and
reports ``"interrupted": true`` with exit code 0 (private_cli.py:1889-1902); a sketch rank has nothing partial to keep
(its columns need every rank's sketches) and reports the interrupt.
"""

from __future__ import annotations

import importlib
import json
import logging
import os
import sys
import traceback
from pathlib import Path
from types import SimpleNamespace

import numpy as np


def _make_engine(spec: dict):
    """The rank's HipEngine on the device ``resolve_device`` names (LOCAL_RANK modulo the devices).  ``engine_factory``
    (``"module:callable"``) exists for the CPU tests, which have no GPU to give the ranks."""
    factory = spec.get("engine_factory")
    if factory:
        module, _, name = factory.partition(":")
        return getattr(importlib.import_module(module), name)()
    from .engine import HipEngine
    from .methods.sourmash_hip import resolve_device

    return HipEngine(resolve_device())


def _configuration(spec: dict):
    return SimpleNamespace(**spec["configuration"])


def sourmash_rank(spec: dict, rank: int, world: int, dist, torch, logger: logging.Logger) -> dict:
    from . import wire
    from .distributed import sharded_pair_step
    from .methods import sourmash_hip

    config = _configuration(spec)
    kmersize, scaled = int(config.kmersize), sourmash_hip.parse_scaled(config.extra)
    files = [Path(p) for p in spec["fasta_files"]]
    bounds = [tuple(b) for b in spec["shards"]]
    g0, g1 = bounds[rank]
    shard_sizes = [b - a for a, b in bounds]
    engine = _make_engine(spec)
    backend = spec["backend"]
    # ---- 1. this rank's genomes: one pass for checksum, length, title and sketch (host front-end of batch i+1
    #         overlapped with the device work of batch i)
    meta, local = [], []
    for batch_paths, infos, sketches in sourmash_hip.sketch_fasta_batches(logger, files[g0:g1], kmersize=kmersize, scaled=scaled, engine=engine):
        for path, info, mins in zip(batch_paths, infos, sketches):
            meta.append({"path": str(path), "md5": info.md5, "length": info.length, "description": info.description})
            local.append(mins)
    # ---- 2. signature files of the own genomes (same cache the single-process driver and the reference use)
    sig_dir = sourmash_hip.sig_cache_dir(Path(spec["cache"]), kmersize, config.extra)
    sig_dir.mkdir(parents=True, exist_ok=True)
    from . import sig
    from .engine import max_hash_for_scaled

    max_hash = max_hash_for_scaled(scaled)
    fasta_dir = Path(spec["fasta_dir"])
    seen: set[str] = set()
    missing = []
    for i, m in enumerate(meta):  # a checksum met twice inside the shard gets one file; the parent reports the duplicate
        if m["md5"] not in seen and not (sig_dir / f"{m['md5']}.sig").is_file():
            missing.append(i)
        seen.add(m["md5"])
    if missing:
        sig.write_sigs([sig_dir / f"{meta[i]['md5']}.sig" for i in missing], names=[meta[i]["md5"] for i in missing],
                       filenames=[str(fasta_dir / Path(meta[i]["path"]).name) for i in missing], ksize=kmersize, max_hash=max_hash,
                       sketches=[local[i] for i in missing])  # fmt: skip
    # ---- 3. every rank learns every genome's identity (small Python objects), then the ONE data-path collective
    all_meta: list = [None] * world
    dist.all_gather_object(all_meta, meta)
    order = [m["md5"] for shard in all_meta for m in shard]
    n_total = len(order)
    assert n_total == len(files) and [len(x) for x in all_meta] == shard_sizes
    dup = len(set(order)) != n_total
    if dup:  # the parent words the message (it names the files); nothing to compute
        return {"ok": True, "meta": meta, "duplicate_md5": True}
    sk_local = engine.sketches_from_host(local)
    own = (g0, g1)
    sk_all, counts = sharded_pair_step(engine, torch, dist, sk_local, shard_sizes, (0, n_total), own, backend=backend)
    # ---- 4. the strict transform (host libm pow: the reference's doubles) and the column-tile file
    tile_file = None
    if counts is not None:
        from .engine import ani_host

        sizes = sk_all.sizes()
        c = counts.cpu().numpy().view(np.uint32)
        ident, cov, null = ani_host(c, sizes, sizes[g0:g1], kmersize)
        sourmash_hip.check_self_comparisons(order, order[g0:g1], ident, null)
        subjects = order[g0:g1]
        if spec.get("columns") is not None:  # a resumed run: only the subject columns it still needs leave the rank
            wanted = set(spec["columns"])
            keep = [i for i, h in enumerate(subjects) if h in wanted]
            subjects = [subjects[i] for i in keep]
            ident, cov, null = ident[:, keep], cov[:, keep], null[:, keep]
        if subjects:
            tile_file = Path(spec["work_dir"]) / f"{sourmash_hip.METHOD}.rank_{rank}.tile.npz"
            wire.save_tile(tile_file, config, order, subjects, np.ascontiguousarray(ident), np.ascontiguousarray(cov), np.ascontiguousarray(null))
    return {"ok": True, "meta": meta, "tile": str(tile_file) if tile_file else None, "device": str(getattr(engine, "device", "test engine")),
            "backend": backend}  # fmt: skip


def fastani_rank(spec: dict, rank: int, world: int, dist, torch, logger: logging.Logger) -> dict:  # noqa: ARG001
    from .methods import fastani_hip

    config = _configuration(spec)
    run = SimpleNamespace(run_id=spec["run_id"], configuration=config, status="Running")
    session = SimpleNamespace(commit=lambda: None)  # the parent owns the database; an interrupt shows in the result file
    hash_to_filename = dict(spec["hash_to_filename"])
    query_hashes = {h: int(n) for h, n in spec["query_hashes"].items()}
    c0, c1 = spec["column_ranges"][rank]
    if c0 == c1:
        return {"ok": True, "json": None}
    json_file = Path(spec["work_dir"]) / f"{fastani_hip.METHOD}.run_{spec['run_id']}.columns_{c0 + 1}_{c1}.json"
    engine = _make_engine(spec)
    tiles: list[str] = []

    def keep_tile(queries, subjects, ident, aln, sim, cov, null) -> None:
        """The batch as a binary tile file next to the JSON column file (the parent's direct ingest)."""
        from . import wire

        path = Path(spec["work_dir"]) / f"{fastani_hip.METHOD}.rank_{rank}.tile_{len(tiles)}.npz"
        wire.save_tile(path, config, queries, subjects, ident, cov, null, aln_length=aln, sim_errors=sim)
        tiles.append(str(path))

    status = fastani_hip.compute_fastani_hip(
        logger, Path(spec["work_dir"]), session, run, json_file, Path(spec["fasta_dir"]), hash_to_filename,
        {v: k for k, v in hash_to_filename.items()}, query_hashes, "", engine=engine, subject_range=(c0, c1),
        on_block=keep_tile if spec.get("tiles") else None, **({"query_batch": int(spec["query_batch"])} if spec.get("query_batch") else {}),
    )  # fmt: skip
    if status:
        return {"ok": False, "error": f"Column worker failed with return code {status}"}
    return {"ok": True, "json": str(json_file), "tiles": tiles, "interrupted": run.status == "Worker interrupted",
            "device": str(getattr(engine, "device", "test engine"))}  # fmt: skip


TASKS = {"sourmash": sourmash_rank, "fastani": fastani_rank}


def main(argv: list[str]) -> int:
    import signal

    from .launch import die_with_parent

    die_with_parent()  # before any GPU call: this rank ends with the process that started it
    # Ctrl-C and a scheduler's SIGTERM alike end the work through KeyboardInterrupt (pyani_plus/private_cli.py:816-823)
    signal.signal(signal.SIGINT, signal.default_int_handler)
    signal.signal(signal.SIGTERM, signal.default_int_handler)
    spec_file = Path(argv[1])
    spec = json.loads(spec_file.read_text())
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    result_file = spec_file.parent / f"result_rank{rank}.json"
    logging.basicConfig(level=logging.INFO, format=f"[rank {rank}] %(levelname)s %(message)s")
    logger = logging.getLogger("pyani_plus_amd.worker")
    result: dict
    dist = None
    needs_group = False
    try:
        import torch
        import torch.distributed as dist

        backend = spec["backend"]
        needs_group = spec["task"] == "sourmash"  # the fragment-ANI shards exchange nothing
        if needs_group:
            if backend == "nccl":
                from .methods.sourmash_hip import resolve_device

                device = resolve_device()
                torch.cuda.set_device(device)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        try:
            result = TASKS[spec["task"]](spec, rank, world, dist, torch, logger)
        finally:
            if needs_group and dist.is_initialized():
                if sys.exc_info()[0] is None:
                    dist.barrier()
                    dist.destroy_process_group()
    except KeyboardInterrupt:  # outside the column worker's own handling: nothing partial to keep
        logger.error("Interrupted")  # noqa: TRY400
        result = {"ok": True, "interrupted": True}
    except SystemExit as err:  # log_sys_exit: the worker's own error message
        result = {"ok": False, "error": str(err.code) if err.code not in (None, 0) else "worker exited"}
    except Exception as err:  # noqa: BLE001
        traceback.print_exc()
        result = {"ok": False, "error": f"{type(err).__name__}: {err}"}
    signal.signal(signal.SIGINT, signal.SIG_IGN)  # the report is written whatever arrives now
    signal.signal(signal.SIGTERM, signal.SIG_IGN)
    tmp = result_file.with_suffix(".json.part")
    tmp.write_text(json.dumps(result))
    tmp.replace(result_file)  # the parent never reads half a report
    sys.stdout.flush()
    sys.stderr.flush()
    if not result.get("ok"):
        os._exit(1)  # peers may be blocked in a collective with this rank: leave without waiting for them
    if result.get("interrupted") and needs_group:
        os._exit(0)  # likewise: no barrier, no orderly shutdown of a process group whose peers are stuck
    return 0


if __name__ == "__main__":
    raise SystemExit(main(sys.argv))
