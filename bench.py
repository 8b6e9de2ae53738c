#!/usr/bin/env python3
"""bench.py -- pairwise genome comparisons / second on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path over one batch of synthetic genomes
already resident in HBM as a 2-bit arena:
    k-mer hash + FracMinHash filter -> sort/unique -> (RCCL all-gather of sketches)
    -> dictionary + bit-row intersection counts -> containment ANI (f64 matrices in HBM).

    python bench.py --gpus N --steps K --warmup W
With N > 1 and no RANK in the environment the script launches its N ranks itself (fresh child
processes, started before this process has touched the GPU) and relays rank 0's JSON line; under
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
it is one of the ranks.

Workloads (BASELINE.json configs): N=1 -> configs[1] "1 000 synthetic 5 Mb genomes, k=31,
scaled=1000"; N=8 -> configs[2] "10 000 genomes tiled across 8 MI355X"; N=2/4 use the same
1 250 genomes per GPU as configs[2].  Rank 0 prints ONE JSON line.  At N=1 the line also carries,
under "also", short runs of the other BASELINE configs (bottom-m mode, 10 000 genomes on one GPU,
the mixed-length set, fastANI-style fragment ANI) and of the headline workload at k = 51, each with its
own parity check.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SIMDS = 1024  # 256 CUs x 4 SIMDs


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genomes", type=int, default=0, help="total genomes (default: 1000 at 1 GPU, 1250 per GPU otherwise)")
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--kmer", type=int, default=31)
    ap.add_argument("--scaled", type=int, default=1000)
    ap.add_argument("--species", type=int, default=40)
    ap.add_argument("--sketch-mode", choices=["scaled", "bottom"], default="scaled",
                    help="scaled = the reference's FracMinHash + containment ANI (default, parity-pinned); "
                    "bottom = bottom-m MinHash + Mash Jaccard ANI as BASELINE configs[1] words it (parity unpinned)")
    ap.add_argument("--bottom-m", type=int, default=1000)
    ap.add_argument("--mixed-lengths", action="store_true", help="log-uniform 100 kb - 10 Mb genomes (BASELINE configs[4])")
    ap.add_argument("--cpu-sample-genomes", type=int, default=0, help="genomes sketched by the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive passes (keeps a rocprof kernel trace to the timed steps' launches)")
    ap.add_argument("--no-also", action="store_true", help="skip the short runs of the other BASELINE configs")
    ap.add_argument("--also", default="bottom,k51,mixed,n10000,fragani,rearranged,config1", help="comma list of the extra runs at N=1 (rearranged: fragment ANI on the set with indels, rearrangements, repeats and contigs; config1: BASELINE configs[0] from files, the last key of the line)")
    ap.add_argument("--also-fragani-genomes", type=int, default=1000)
    ap.add_argument("--also-n", type=int, default=10000)
    ap.add_argument("--no-fresh-child", action="store_true", help="skip the fresh-process fragment-ANI call (a child started before this process touches the GPU)")
    ap.add_argument("--fresh-fragani-child", action="store_true", help=argparse.SUPPRESS)  # the child's own mode
    ap.add_argument("--dry-run-plan", action="store_true",
                    help="print, without touching a GPU (or importing torch), what every rank of `--gpus N` would hold and exchange: "
                    "shards, all-gather sizes, tile buffers, the strong_basis memory need")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------- fragment ANI in a fresh process
def fresh_fragani_child(args) -> None:
    """The child's side: one all-columns fragment-ANI call as the first device work of a new process (after the arena
    is there): what a `fastANI-hip` worker pays, the first use of its workspace's device memory included.  Prints one JSON line."""
    t_proc = time.perf_counter()
    import torch

    from pyani_plus_amd.engine import HipEngine
    from pyani_plus_amd.synth import synth_arena_torch

    n, k, frag = args.also_fragani_genomes, 16, 3000
    engine = HipEngine(0)
    arena = synth_arena_torch(engine, n, args.length, n_species=args.species)
    starts = np.ascontiguousarray(arena.genome_start[:-1])
    lens = np.full(n, args.length, dtype=np.uint32)
    genome = np.arange(n, dtype=np.uint32)
    torch.cuda.synchronize()
    free0, total_mem = torch.cuda.mem_get_info()
    t_ready = time.perf_counter()
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        total, matched, _sums = engine.fragani(arena, starts, lens, genome, k, frag)
        times.append(time.perf_counter() - t0)
    free1, _ = torch.cuda.mem_get_info()
    print(json.dumps({
        "genomes": n, "first_call_seconds": times[0], "second_call_seconds": times[1],
        "workspace_device_bytes": int(free0 - free1), "arena_device_bytes": int(arena.packed.numel() * 4 + arena.mask.numel() * 4),
        "device_bytes_in_use_after": int(total_mem - free1), "seconds_from_process_start_to_arena": t_ready - t_proc,
        "kept_fragments_checksum": int(matched.astype(np.uint64).sum()), "fragments": int(total.astype(np.uint64).sum()),
    }), flush=True)
    engine.close()


def run_fresh_fragani_child(args) -> dict:
    """Start the child above from THIS process, which has not touched the GPU (no torch import yet): a fresh child, not a
    re-exec of a process that holds the device.  Its one JSON line, or the reason there is none."""
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"skipped": "under a profiler the parent holds the device before it starts"}
    cmd = [sys.executable, str(Path(__file__).resolve()), "--fresh-fragani-child", "--also-fragani-genomes", str(args.also_fragani_genomes),
           "--length", str(args.length), "--species", str(args.species)]
    try:
        proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        return {"error": "the child did not finish in 600 s"}
    for line in reversed(proc.stdout.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return {"error": f"child exited with {proc.returncode}: {proc.stderr[-400:]}"}


# --------------------------------------------------------------------------- the plan of a multi-GPU run, on paper
def dry_run_plan(args) -> dict:
    """What `bench.py --gpus N` allocates and exchanges per rank, from the same shard arithmetic the ranks use and the
    expected sketch size (L - k + 1) / scaled of i.i.d. synthetic genomes -- so that an out-of-memory condition or a
    collective that outlasts the watchdog in the `world > 1` RCCL branch (never executed on hardware: no 8-GPU node has
    been available to any round) is found on paper.  Touches no GPU and does not import torch."""
    from pyani_plus_amd.distributed import shard_bounds, shard_bounds_by_cost
    from pyani_plus_amd.synth import mixed_lengths

    world = max(1, args.gpus)
    n_total = args.genomes or (1000 if world == 1 else 1250 * world)
    lengths = mixed_lengths(n_total) if args.mixed_lengths else [args.length] * n_total
    bounds = shard_bounds_by_cost(lengths, world) if args.mixed_lengths else shard_bounds(n_total, world)
    bottom = args.sketch_mode == "bottom"

    def sketch_size(length: int) -> int:
        windows = max(0, length - args.kmer + 1)
        return min(args.bottom_m, windows) if bottom else int(round(windows / args.scaled))

    sizes = [sketch_size(x) for x in lengths]
    padded = [(x // 64 + 1) * 64 for x in lengths]  # synth._padded: at least one invalid position after the last base
    shard_sizes = [b - a for a, b in bounds]
    totals = [sum(sizes[a:b]) for a, b in bounds]
    max_n, max_total = max(max(shard_sizes), 1), max(1, max(totals))
    hbm = 288e9
    # FracMinHash sizes are binomial around their expectation: the per-rank totals of a real run differ by ~sqrt(total)
    # hashes, so every rank but the one with the largest total pads its payload (a staging copy) and the gathered buffer is
    # cut and joined (torch.cat: a second buffer of the same size).  Only bottom-m sketches have equal totals by construction.
    equal_totals = bottom and all(t == max_total for t in totals)
    spread = 0 if bottom else int(3 * max(totals) ** 0.5)  # three sigma on the largest total: the padded length in practice
    max_total += spread
    ranks = []
    for r, (a, b) in enumerate(bounds):
        arena_bytes = sum(padded[a:b]) // 4 + sum(padded[a:b]) // 8  # 2-bit bases + 1-bit mask
        cols = b - a
        gathered = world * max_total * 8
        own_payload = 0 if equal_totals else max_total * 8  # a staging copy whenever the rank's own buffer is shorter than the padded length
        counts = n_total * cols * 4
        matrices_dev = 2 * n_total * cols * 8       # value_t_dev's device-pow matrices
        pinned_host = n_total * cols * (4 + 8 + 8)  # counts + identity + cov_query of the timed (strict) step
        sketch_ws = 2 * (totals[r] * 5 // 4 + 128 * cols) * 8  # candidate regions (expectation + 25 % + 128 slots per genome), hashes and sorted copy
        ranks.append({
            "rank": r, "genomes": [a, b], "n_genomes": cols, "bases": sum(lengths[a:b]), "arena_bytes": arena_bytes,
            "own_hashes": totals[r], "payload_padded_to": max_total, "payload_staging_copy": own_payload > 0,
            "allgather_sizes_bytes_sent": max_n * 8, "allgather_sizes_bytes_received": world * max_n * 8,
            "allgather_payload_bytes_sent": max_total * 8, "allgather_payload_bytes_received": gathered,
            "torch_cat_after_gather": not equal_totals,
            "subject_columns": [a, b], "counts_tile_bytes": counts, "device_matrices_bytes": matrices_dev, "pinned_host_bytes": pinned_host,
            "device_bytes_estimate": arena_bytes + sketch_ws + gathered * (1 if equal_totals else 2) + own_payload + counts + matrices_dev,
        })
    # per link: a ring all-gather moves (world - 1) / world of the gathered buffer through every rank, over up to 7 xGMI links
    ring_bytes = (world - 1) * max_total * 8
    xgmi_link_gbs = 153.0  # MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point
    plan = {
        "plan_of": f"python bench.py --gpus {world}" + (f" --genomes {args.genomes}" if args.genomes else "") + (" --mixed-lengths" if args.mixed_lengths else ""),
        "world": world, "genomes": n_total, "k": args.kmer, "sketch_mode": args.sketch_mode, "scaled": None if bottom else args.scaled,
        "expected_sketch_size": {"mean": sum(sizes) / max(1, n_total), "min": min(sizes), "max": max(sizes)},
        "shards": "contiguous genome ranges balanced by " + ("cumulative length" if args.mixed_lengths else "count"),
        "shard_balance_bases_max_over_mean": max(x["bases"] for x in ranks) / (sum(x["bases"] for x in ranks) / world),
        "collectives_per_step": [
            {"what": "per-genome sketch sizes, padded to the largest shard", "call": "all_gather_into_tensor(int64)", "bytes_per_rank": max_n * 8},
            {"what": "sketch payload, padded to the largest per-rank total", "call": "all_gather_into_tensor(int64, async) overlapped with pa_pair_dict_prepare",
             "bytes_per_rank": max_total * 8, "ring_bytes_through_each_rank": ring_bytes,
             "seconds_at_one_xgmi_link": ring_bytes / (xgmi_link_gbs * 1e9)},
        ],
        "equal_per_rank_totals": equal_totals,
        "padded_payload_hashes": max_total,
        "ranks": ranks,
        "max_device_bytes_estimate": max(x["device_bytes_estimate"] for x in ranks),
        "hbm_bytes": hbm,
    }
    if world > 1:
        full_arena = sum(padded) // 4 + sum(padded) // 8
        full_total = sum(sizes)
        # rank 0 runs the whole workload alone after the timed steps (strong_basis) while the others wait in dist.barrier()
        basis_bytes = full_arena + 2 * (full_total * 5 // 4 + 128 * n_total) * 8 + n_total * n_total * (4 + 16) + ranks[0]["arena_bytes"]
        hash_s = sum(lengths) / 5e6 * 11.4e-6      # kmer_hash_kernel<31>: 11.4 ms per 1 000 x 5 Mb (BENCH_r04)
        pair_s = 25.5e-9 * n_total * n_total       # pair phase + transform at N = 10 000: 2.55 s per 10^8 ordered pairs (r04 `also.n10000_one_gpu`)
        gen_s = sum(lengths) / 5e9 * 4.0           # the synthetic generator: ~4 s per 1 000 x 5 Mb on the device
        plan["strong_basis"] = {
            "what": "after the timed steps rank 0 generates ALL genomes and runs 1 warm-up + 2 steps alone; ranks 1.. wait in dist.barrier()",
            "rank0_device_bytes_estimate": basis_bytes, "fits_hbm": basis_bytes < 0.9 * hbm,
            "full_arena_bytes": full_arena,
            "estimated_seconds_others_wait": gen_s + 3 * (hash_s + pair_s),
            "watchdog": "torch.distributed's default collective timeout is 600 s: the wait must stay below it (PA_BENCH_NO_BASIS=1 skips the leg)",
            "within_watchdog": gen_s + 3 * (hash_s + pair_s) < 600.0,
        }
    return plan


# --------------------------------------------------------------------------- launcher (parent of the ranks)
def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks as fresh child processes.

    This process never initialises HIP (it does not even import torch): a process that has touched the GPU
    must not fork/exec workers on this pool.  Rank 0's stdout is captured and its JSON line relayed."""
    port = _free_port()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(
            subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                             stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=rank == 0)
        )  # fmt: skip
    # A rank that dies (bad device index, out of memory) leaves the others waiting in the rendezvous or in a
    # barrier: watch all of them and, when one has failed, end the ones this process started (by handle, not by name).
    out = None
    while out is None:
        try:
            out, _ = procs[0].communicate(timeout=1.0)
        except subprocess.TimeoutExpired:
            if any(p.poll() not in (None, 0) for p in procs[1:]):
                time.sleep(5.0)  # let rank 0 report the failure itself if it can
                for p in procs:
                    if p.poll() is None:
                        p.kill()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for cand in reversed((out or "").splitlines()):
        cand = cand.strip()
        if cand.startswith("{") and cand.endswith("}"):
            try:
                json.loads(cand)
            except ValueError:
                continue
            line = cand
            break
    rest = [x for x in (out or "").splitlines() if x.strip() != (line or "")]
    if rest:
        print("\n".join(rest), file=sys.stderr)
    bad = [c for c in codes if c != 0]
    if bad or line is None:
        print(f"bench.py: ranks exited with {codes}" + ("" if line else " and rank 0 printed no JSON line"), file=sys.stderr)
        return bad[0] if bad else 1
    print(line, flush=True)
    return 0


# --------------------------------------------------------------------------- CPU baseline (oracle, checker only)
def _ascii_genomes(engine, arena, sample, lengths):
    """Unpack sampled genomes to ASCII on the GPU (plumbing) for the oracle."""
    t = engine.torch
    lut = t.tensor(list(b"ACGT"), dtype=t.uint8, device=engine.device)
    shifts = (t.arange(16, device=engine.device, dtype=t.int32) * 2)[None, :]
    seqs = []
    for g in sample:
        s0, s1 = int(arena.genome_start[g]), int(arena.genome_start[g + 1])
        words = arena.packed[s0 // 16 : s1 // 16]
        codes = ((words[:, None] >> shifts) & 3).reshape(-1)[: lengths[g]].to(t.int64)
        seqs.append(lut[codes].cpu().numpy())
    return seqs


def cpu_baseline(engine, arena, sk, args, n_total: int, lengths: list[int]) -> dict:
    """Time the oracle's tuned scalar form on the workload and check GPU == CPU on everything it computed.

    Up to 2 000 genomes the WHOLE workload runs (every genome sketched, all N x N intersections and the ANI transform:
    about 7 s on the box's 16 quota CPUs for 1 000 x 5 Mb) -- the text of the genomes passes through host memory in
    chunks of 128, unpacked on the GPU (plumbing, not timed).  Larger sets are sampled and extrapolated."""
    import oracle
    from pyani_plus_amd.synth import arena_to_ascii, device_arena_to_host

    # threads that can run at once: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 CPUs to a
    # container that is allowed 16 CPUs' worth of time; 256 busy threads there are throttled, not faster)
    from pyani_plus_amd import _capi

    visible = len(os.sched_getaffinity(0))
    cores = max(1, min(visible, int(_capi.load_library().pa_host_cpu_budget())))
    whole = arena.n_genomes <= 2000 and not args.cpu_sample_genomes
    n_samp = arena.n_genomes if whole else (args.cpu_sample_genomes or max(2, min(arena.n_genomes, max(8 * cores, 16), 512)))
    sample = list(range(n_samp))
    gpu_sk = sk.to_host()
    first = _ascii_genomes(engine, arena, sample[:1], lengths)
    if n_samp <= 4:  # tiny runs: also exercise the host-side unpacker
        seqs = _ascii_genomes(engine, arena, sample, lengths)
        host = device_arena_to_host(arena, sample, lengths[: n_samp])
        assert all(arena_to_ascii(host, i) == seqs[i].tobytes() for i in range(n_samp))
    oracle.sketch_many(first, args.kmer, args.scaled, threads=1, fast=True)  # warm (table init, page-in)
    t_sketch_total, sample_bases = 0.0, 0
    chunk = 128
    for c0 in range(0, n_samp, chunk):
        part = sample[c0 : c0 + chunk]
        seqs = _ascii_genomes(engine, arena, part, lengths)
        t0 = time.perf_counter()
        cpu_sk = oracle.sketch_many(seqs, args.kmer, args.scaled, threads=cores, fast=True)
        t_sketch_total += time.perf_counter() - t0
        sample_bases += sum(lengths[g] for g in part)
        for i, g in enumerate(part):
            if not np.array_equal(cpu_sk[i], gpu_sk[g]):
                raise SystemExit(f"PARITY FAILURE: sketch of genome {g} differs between HIP and oracle")
        del seqs, cpu_sk
    t_base = t_sketch_total / sample_bases  # wall seconds per base with `cores` threads
    t_sketch = t_base * sum(lengths) / n_total  # per average genome
    # pairs: the GPU sketches (proven equal above), all of them when the whole workload runs
    n_pair = len(gpu_sk) if whole else min(len(gpu_sk), 384)
    block = gpu_sk[:n_pair]
    oracle.pair_counts(block[:8], threads=cores)
    t0 = time.perf_counter()
    cpu_counts = oracle.pair_counts(block, threads=cores)
    sizes = [len(s) for s in block]
    cpu_ani = oracle.ani(cpu_counts, sizes, sizes, args.kmer)
    t_pairs_total = time.perf_counter() - t0
    t_pair = t_pairs_total / (n_pair * n_pair)
    est = (t_sketch_total + t_pairs_total) if whole else n_total * t_sketch + n_total * n_total * t_pair
    # the same two steps on one thread (SURVEY.md 8d asks for both figures): 2 genomes, a 96 x 96 block
    seqs2 = _ascii_genomes(engine, arena, sample[:2], lengths)
    t0 = time.perf_counter()
    oracle.sketch_many(seqs2, args.kmer, args.scaled, threads=1, fast=True)
    t_sketch_1 = (time.perf_counter() - t0) / sum(lengths[g] for g in sample[:2]) * sum(lengths) / n_total
    n1 = min(n_pair, 96)
    t0 = time.perf_counter()
    oracle.pair_counts(block[:n1], threads=1)
    t_pair_1 = (time.perf_counter() - t0) / (n1 * n1)
    est_1 = n_total * t_sketch_1 + n_total * n_total * t_pair_1
    if whole:
        what = (f"the whole workload, not a sample: all {n_samp} genomes ({sample_bases / 1e6:.0f} Mb) sketched in {t_sketch_total:.2f} s + all "
                f"{n_pair}x{n_pair} sketch intersections and the ANI transform in {t_pairs_total:.2f} s, {cores} OpenMP threads (oracle tuned scalar form)")
    else:
        what = (f"{n_samp} genomes ({sample_bases / 1e6:.0f} Mb) sketched + {n_pair}x{n_pair} sketch pairs+ANI with {cores} OpenMP threads "
                f"(oracle tuned scalar form); extrapolated to N={n_total}: N*{t_sketch:.4f}s + N^2*{t_pair * 1e6:.3f}us")
    return {
        "value": n_total * n_total / est,
        "unit": "pairs/s",
        "cores": cores,
        "cpus_visible": visible,
        "kind": "port",
        "sample": what,
        "extrapolated": not whole,
        "seconds": est,
        "sketch_s_per_genome": t_sketch,
        "pair_us": t_pair * 1e6,
        "one_thread": {"value": n_total * n_total / est_1, "sketch_s_per_genome": t_sketch_1, "pair_us": t_pair_1 * 1e6,
                       "sample": "2 genomes + a 96x96 block on one thread, extrapolated"},
        "_cpu_counts": cpu_counts,
        "_cpu_ani": cpu_ani,
        "_n_pair": n_pair,
    }


def reference_tools_probe() -> dict:
    """time the real tools when the box has them.  It does not (no network, Rust/C++ third-party
    binaries), so this records their absence instead of leaving the question open."""
    import shutil

    found = {name: shutil.which(name) for name in ("sourmash", "fastANI")}
    return {"sourmash": found["sourmash"] or "absent", "fastANI": found["fastANI"] or "absent",
            "timed": False if not any(found.values()) else "not implemented: reference tools present but their timing leg is not wired"}


# --------------------------------------------------------------------------- the extra single-GPU runs ("also")
def _time_steps(torch, fn, steps: int, warmup: int = 1):
    for _ in range(warmup):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def also_bottom(engine, arena, args, n_total, lengths) -> dict:
    """BASELINE configs[1] as worded: bottom-m MinHash + Mash Jaccard (parity unpinned: not a reference mode)."""
    import oracle

    torch = engine.torch
    m = args.bottom_m

    def step():
        sk = engine.sketch_bottom(arena, args.kmer, m)
        common, denom = engine.pair_mash(sk, m, (0, n_total), (0, n_total))
        return sk, common, denom, engine.ani_mash(common, denom, args.kmer)

    sec, (sk, common, denom, _ani) = _time_steps(torch, step, 3)
    # parity on a sample: bottom-m sketches of 3 genomes and a 48 x 48 Mash block against the oracle
    sample = [0, 1, n_total - 1]
    seqs = _ascii_genomes(engine, arena, sample, lengths)
    got = sk.to_host()
    for g, seq in zip(sample, seqs):
        want = oracle.sketch_bottom_seq(seq.tobytes(), args.kmer, m)
        if not np.array_equal(got[g], want):
            raise SystemExit(f"PARITY FAILURE (bottom-m): sketch of genome {g} differs from the oracle")
    nb = min(n_total, 48)
    o_common, o_denom = oracle.mash_pairs(got[:nb], m)
    if not (np.array_equal(common[:nb, :nb].cpu().numpy().view(np.uint32), o_common)
            and np.array_equal(denom[:nb, :nb].cpu().numpy().view(np.uint32), o_denom)):
        raise SystemExit("PARITY FAILURE (bottom-m): Mash common/denominator block differs from the oracle")
    return {
        "workload": f"{n_total} synthetic {args.length / 1e6:g} Mb genomes, k={args.kmer} bottom-m={m} MinHash + NxN Mash-Jaccard ANI (BASELINE configs[1] as worded)",
        "ms_per_step": sec * 1e3, "pairs_per_s": n_total * n_total / sec, "steps": 3,
        "parity": f"unpinned mode (the reference never uses num>0); sketches of 3 genomes and a {nb}x{nb} common/denominator block equal the oracle's Mash restatement",
    }


def also_long_kmer(engine, arena, args, n_total, lengths) -> dict:
    """The headline workload at k = 51, sourmash's third default size (the reference passes any --kmersize on,
    pyani_plus/public_cli_args.py:229): k above 32 takes kmer_hash_long_kernel."""
    import oracle

    torch = engine.torch
    k = 51

    def step():
        sk = engine.sketch(arena, k, args.scaled)
        counts = engine.pair_counts(sk)
        return sk, counts, engine.ani(counts, sk, k)

    sec, (sk, counts, ani) = _time_steps(torch, step, 3)
    del ani
    sample = [0, n_total // 2, n_total - 1]
    seqs = _ascii_genomes(engine, arena, sample, lengths)
    off = sk.offsets_host().astype(np.int64)
    mine = []
    for g, seq in zip(sample, seqs):
        mine.append(sk.hashes[int(off[g]) : int(off[g + 1])].cpu().numpy().view(np.uint64))
        if not np.array_equal(mine[-1], oracle.sketch_seq(seq.tobytes(), k, args.scaled)):
            raise SystemExit(f"PARITY FAILURE (k={k}): sketch of genome {g} differs from the oracle")
    block = counts[sample][:, sample].cpu().numpy().view(np.uint32)
    if not np.array_equal(block, oracle.pair_counts(mine)):
        raise SystemExit(f"PARITY FAILURE (k={k}): the sampled 3x3 count block differs from the oracle")
    return {
        "workload": f"{n_total} synthetic {args.length / 1e6:g} Mb genomes, k={k} scaled={args.scaled} sketch + NxN ANI on one GPU",
        "ms_per_step": sec * 1e3, "pairs_per_s": n_total * n_total / sec, "steps": 3,
        "parity": "sketches of the first, middle and last genome and their 3x3 count block equal the oracle",
    }


def pair_tile_balance(sizes, bounds) -> dict:
    """How evenly the PAIR phase is spread when rank r evaluates all queries against its own genomes as subject
    columns, from the sketch sizes: dictionary inserts = the rank's own postings; merge-cost
    model of its tile, sum over (query, own subject) of |S_q| + |S_s| = n_own * sum|S| + N * own postings (SURVEY.md
    section 8d counts a pair as its two sketches); subject columns.  Each as max over ranks / mean over ranks."""
    sizes = [int(x) for x in sizes]
    n, total = len(sizes), sum(sizes)
    own = [sum(sizes[a:b]) for a, b in bounds]
    cols = [b - a for a, b in bounds]
    model = [c * total + n * o for c, o in zip(cols, own)]

    def spread(values):
        mean = sum(values) / len(values)
        return max(values) / mean if mean > 0 else None

    return {"dictionary_inserts_max_over_mean": spread(own), "pair_cost_model_max_over_mean": spread(model),
            "subject_columns_max_over_mean": spread(cols), "own_postings_by_rank": own, "subject_columns_by_rank": cols}


def also_mixed(engine, args) -> dict:
    """BASELINE configs[4]: 2 000 genomes of 100 kb - 10 Mb on one GPU (its 8-GPU tiling is the driver's to run)."""
    import oracle
    from pyani_plus_amd import _capi
    from pyani_plus_amd.distributed import shard_bounds_by_cost
    from pyani_plus_amd.synth import mixed_lengths, synth_arena_torch

    torch = engine.torch
    n = 2000
    lengths = mixed_lengths(n)
    arena = synth_arena_torch(engine, n, lengths, n_species=args.species)

    def step():
        sk = engine.sketch(arena, args.kmer, args.scaled)
        counts = engine.pair_counts(sk)
        return sk, counts, engine.ani(counts, sk, args.kmer)

    sec, (sk, counts, _ani) = _time_steps(torch, step, 3)
    # parity: the shortest, the longest and one middle genome against the oracle; a block against the merge kernel
    order = np.argsort(lengths)
    sample = [int(order[0]), int(order[n // 2]), int(order[-1])]
    seqs = _ascii_genomes(engine, arena, sample, lengths)
    got = sk.to_host()
    for g, seq in zip(sample, seqs):
        if not np.array_equal(got[g], oracle.sketch_seq(seq.tobytes(), args.kmer, args.scaled)):
            raise SystemExit(f"PARITY FAILURE (mixed lengths): sketch of genome {g} ({lengths[g]} bp) differs from the oracle")
    chk = engine.pair_counts(sk, (0, 256), (0, 256), algo=_capi.PA_PAIRS_MERGE)
    if not torch.equal(chk, counts[:256, :256]):
        raise SystemExit("PARITY FAILURE (mixed lengths): bit-row and merge counts differ")
    ocounts = oracle.pair_counts(got[:64])
    if not np.array_equal(counts[:64, :64].cpu().numpy().view(np.uint32), ocounts):
        raise SystemExit("PARITY FAILURE (mixed lengths): pair counts differ from the oracle")
    # how evenly would the 8-GPU shards of this set be loaded: hash cost ~ bases per shard
    bounds8 = shard_bounds_by_cost(lengths, 8)
    loads = [sum(lengths[a:b]) for a, b in bounds8]
    tiles = pair_tile_balance(sk.sizes(), bounds8)
    del arena
    return {
        "workload": f"{n} synthetic genomes, log-uniform 100 kb-10 Mb ({sum(lengths) / 1e9:.2f} Gb), k={args.kmer} scaled={args.scaled} (BASELINE configs[4] on one GPU)",
        "ms_per_step": sec * 1e3, "pairs_per_s": n * n / sec, "steps": 3,
        "shard_balance": {"shards": 8, "bases_max_over_mean": max(loads) / (sum(loads) / len(loads)), "pair_tiles": tiles,
                          "note": "length-balanced contiguous shards (distributed.shard_bounds_by_cost); hashing cost is proportional to bases; "
                                  "pair_tiles: the pair phase of rank r = all queries x its own genomes, priced from the sketch sizes "
                                  "(a rank's postings follow its bases, its column count does not: short genomes mean many columns)"},
        "parity": "sketches of the shortest, median and longest genome equal the oracle; a 64x64 count block equals the oracle and a 256x256 block equals the merge kernel",
    }


def also_n10000(engine, args) -> dict:
    """BASELINE configs[2]'s 10 000 genomes on ONE GPU (five 2 048-column subject tiles)."""
    import oracle
    from pyani_plus_amd import _capi
    from pyani_plus_amd.synth import synth_arena_torch

    torch = engine.torch
    n = args.also_n
    lengths = [args.length] * n
    arena = synth_arena_torch(engine, n, lengths, n_species=args.species)

    def step():
        sk = engine.sketch(arena, args.kmer, args.scaled)
        counts = engine.pair_counts(sk)
        return sk, counts, engine.ani(counts, sk, args.kmer)

    sec, (sk, counts, ani) = _time_steps(torch, step, 2)
    engine.prof_enable(True)
    engine.prof_reset()
    step()
    phases = {name: v[0] for name, v in engine.prof_get().items() if v[1] and not name.startswith("frag")}
    engine.prof_enable(False)
    sample = [0, n // 2, n - 1]
    seqs = _ascii_genomes(engine, arena, sample, lengths)
    off = sk.offsets_host().astype(np.int64)
    flat = sk.hashes
    for g, seq in zip(sample, seqs):
        mine = flat[int(off[g]) : int(off[g + 1])].cpu().numpy().view(np.uint64)
        if not np.array_equal(mine, oracle.sketch_seq(seq.tobytes(), args.kmer, args.scaled)):
            raise SystemExit(f"PARITY FAILURE (N={n}): sketch of genome {g} differs from the oracle")
    # a block that straddles a tile boundary (columns 2000..2100) against the merge kernel, a corner against the oracle
    lo, hi = min(2000, n - 1), min(2100, n)
    chk = engine.pair_counts(sk, (0, 128), (lo, hi), algo=_capi.PA_PAIRS_MERGE)
    if not torch.equal(chk, counts[:128, lo:hi]):
        raise SystemExit(f"PARITY FAILURE (N={n}): bit-row and merge counts differ across the tile boundary")
    if n > 4200:  # rows of the third tile against columns of the first two: computed as the transposed block and mirrored
        chk = engine.pair_counts(sk, (4100, 4200), (lo, hi), algo=_capi.PA_PAIRS_MERGE)
        if not torch.equal(chk, counts[4100:4200, lo:hi]):
            raise SystemExit(f"PARITY FAILURE (N={n}): a mirrored block differs from the merge kernel")
    rows = list(range(n - 24, n))
    sub = [flat[int(off[g]) : int(off[g + 1])].cpu().numpy().view(np.uint64) for g in rows]
    if not np.array_equal(counts[n - 24 :, n - 24 :].cpu().numpy().view(np.uint32), oracle.pair_counts(sub)):
        raise SystemExit(f"PARITY FAILURE (N={n}): the last 24x24 count block differs from the oracle")
    # T_e2e with the bit-identical transform at this size too (packed genomes in pinned host memory -> f64
    # identity and cov_query in host memory; the counts come back and glibc's pow runs on the host threads)
    strict = None
    if not args.no_pcie:
        from pyani_plus_amd.engine import PinnedArena, ani_host, mask_runs

        h_packed = torch.empty(arena.packed.shape, dtype=arena.packed.dtype, pin_memory=True)
        h_packed.copy_(arena.packed)
        h_mask = arena.mask.cpu()
        run_start, run_len = mask_runs(h_mask.numpy().view(np.uint32), int(arena.genome_start[-1]))
        del h_mask
        pinned = PinnedArena(h_packed, run_start, run_len, np.ascontiguousarray(arena.genome_start, dtype=np.uint64))
        h_counts = torch.empty((n, n), dtype=torch.int32, pin_memory=True)
        h_ident, h_cov, h_null = np.empty((n, n)), np.empty((n, n)), np.empty((n, n), dtype=np.uint8)
        runs, pows = [], []
        for it in range(3):  # the first pass warms the buffers of the streamed upload up
            arena.packed.zero_()
            arena.mask.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _dev, sk3 = engine.sketch_streamed(pinned, args.kmer, args.scaled, arena=arena)
            c3 = engine.pair_counts(sk3)
            h_counts.copy_(c3, non_blocking=True)
            sizes3 = sk3.sizes()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            ani_host(h_counts.numpy().view(np.uint32), sizes3, sizes3, args.kmer, symmetric=True, out=(h_ident, h_cov, h_null))
            t1 = time.perf_counter()
            if it:
                runs.append(t1 - t0)
                pows.append(t1 - tp)
        if not torch.equal(c3, counts):
            raise SystemExit(f"PARITY FAILURE (N={n}): streamed and resident pair counts differ")
        nul = h_null.view(np.bool_)
        d_ident = ani[0][-64:, -64:].cpu().numpy()
        if not (np.array_equal(np.isnan(d_ident), nul[-64:, -64:]) and np.allclose(d_ident[~nul[-64:, -64:]], h_ident[-64:, -64:][~nul[-64:, -64:]], rtol=2.3e-16, atol=0)):
            raise SystemExit(f"PARITY FAILURE (N={n}): device-pow and host-libm identities differ by more than 1 ulp")
        mean = sum(runs) / len(runs)
        strict = {"ms_per_step": mean * 1e3, "pairs_per_s": n * n / mean, "runs": len(runs), "host_pow_ms": sum(pows) / len(pows) * 1e3,
                  "non_null_pairs": int((~nul).sum()), "h2d_bytes": int(h_packed.numel() * 4 + 16 * len(run_start)), "d2h_bytes": int(h_counts.numel() * 4),
                  "ani_transform": "host glibc pow on host threads (pa_ani_host): bit-identical to the reference's doubles"}
        del h_packed, h_counts, h_ident, h_cov, h_null, pinned
    del arena, counts, ani
    return {
        "workload": f"{n} synthetic {args.length / 1e6:g} Mb genomes, k={args.kmer} scaled={args.scaled} on one GPU (BASELINE configs[2]'s set; its 8-GPU form is --gpus 8)",
        "ms_per_step": sec * 1e3, "pairs_per_s": n * n / sec, "steps": 2, "phases_ms_per_step": phases, "t_e2e": {"strict": strict},
        "pair_phase": "tile pairs on and above the diagonal of the 5 x 5 tile grid are evaluated, the rest mirrored (|A n B| = |B n A|)",
        "parity": "sketches of 3 genomes equal the oracle; counts across the 2 048-column tile boundary (a block below the tile diagonal, i.e. mirrored) "
        "equal the merge kernel; the last 24x24 block equals the oracle",
    }


def also_fragani_rearranged(engine, args, n_total: int) -> dict:
    """BASELINE configs[3] on genomes that are NOT substitution-only: 1 000 x ~5 Mb with indels (geometric lengths), 3-5
    inversions / translocations, repeat families (5-20 copies of 1-2 kb elements) and 30-200 contigs per genome
    (synth.synth_rearranged_arena_torch) -- the regime of the reference's bacterial fastANI fixtures
    (tests/fixtures/bacterial_example/intermediates/fastANI/*.fastani).  Timed like also.fragment_ani; every genome of one
    species plus a few strangers against one reference is checked against the oracle (integers and float mean exact)."""
    import oracle
    from oracle import pyoracle
    from pyani_plus_amd import _capi
    from pyani_plus_amd.methods.fastani_hip import fastani_mean
    from pyani_plus_amd.synth import species_and_rate, synth_rearranged_arena_torch

    torch = engine.torch
    n = min(n_total, args.also_fragani_genomes)
    k, frag = 16, 3000
    arena, c_start, c_len, c_genome = synth_rearranged_arena_torch(engine, n, args.length, n_species=args.species)
    engine.prof_reset()
    times = []
    for _rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total, matched, ident_sum = engine.fragani(arena, c_start, c_len, c_genome, k, frag)
        times.append(time.perf_counter() - t0)
        if _rep == 0:
            engine.prof_reset()
    sec = min(times[1:])
    prof = engine.prof_get()
    ani = fastani_mean(ident_sum, matched)
    if not np.array_equal(total, np.bincount(c_genome, weights=(c_len // frag), minlength=n).astype(np.uint32)):
        raise SystemExit("PARITY FAILURE (fragment ANI, rearranged set): fragment totals differ from floor(contig length / fragLen) summed over the contigs")

    def contigs_of(g: int) -> list[bytes]:
        s0, e0 = int(arena.genome_start[g]), int(arena.genome_start[g + 1])
        words = arena.packed[s0 // 16 : e0 // 16].cpu().numpy().view(np.uint32)
        codes = ((words[:, None] >> (np.arange(16, dtype=np.uint32) * 2)[None, :]) & 3).astype(np.uint8).reshape(-1)
        text = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
        sel = c_genome == g
        return [text[int(a) - s0 : int(a) - s0 + int(m)].tobytes() for a, m in zip(c_start[sel], c_len[sel])]

    cores = max(1, min(len(os.sched_getaffinity(0)), int(_capi.load_library().pa_host_cpu_budget())))
    ref_g = 0
    sp = [species_and_rate(g, args.species)[0] for g in range(n)]
    queries = [g for g in range(n) if sp[g] == sp[ref_g]] + [g for g in range(n) if sp[g] != sp[ref_g]][:5]
    ref_contigs = contigs_of(ref_g)
    q_contigs = [contigs_of(g) for g in queries]
    pyoracle.fragani_set_fast(True)
    try:
        t0 = time.perf_counter()
        o_ani, o_m, o_t = oracle.fragani_many(q_contigs, ref_contigs, k, frag, 0.0, threads=cores)
        cpu_sec = time.perf_counter() - t0
    finally:
        pyoracle.fragani_set_fast(False)
    bad = [(q, int(matched[q, ref_g]), int(o_m[i])) for i, q in enumerate(queries)
           if matched[q, ref_g] != o_m[i] or total[q] != o_t[i] or (o_m[i] and ani[q, ref_g] != o_ani[i])]
    if bad:
        raise SystemExit(f"PARITY FAILURE (fragment ANI, rearranged set): {len(bad)} of {len(queries)} queries against genome {ref_g} differ from the oracle, first {bad[0]}")
    contigs_per_genome = np.bincount(c_genome, minlength=n)
    out = {
        "workload": f"{n} synthetic ~{args.length / 1e6:g} Mb genomes of {args.species} species that differ from their roots by substitutions, indels (geometric lengths, one per ~8 "
                    "substitutions), 3-5 inversions / translocations of 20-300 kb, three repeat families of 5-20 copies of 1-2 kb elements, and come as 30-200 contigs each; "
                    f"fastANI-style fragment ANI k={k} fragLen={frag}, all ordered pairs in one pa_fragani call",
        "seconds_per_run": sec, "seconds_per_million_pairs": sec * 1e6 / (n * n), "pairs_per_s": n * n / sec, "runs": 2,
        "contigs_per_genome": {"min": int(contigs_per_genome.min()), "mean": float(contigs_per_genome.mean()), "max": int(contigs_per_genome.max())},
        "fragments_per_genome": {"min": int(total.min()), "mean": float(total.mean()), "max": int(total.max())},
        "pairs_with_mappings": int((~np.isnan(ani)).sum()),
        "phases_ms_per_run": {name: v[0] / 2 for name, v in prof.items() if name.startswith("frag")},
        "parity": f"kept/total fragments and ANI of {len(queries)} queries (every genome of species {sp[ref_g]} and five strangers) against genome {ref_g} equal the oracle's "
                  f"(tuned form, {cpu_sec:.1f} s on {cores} CPUs); at 200 genomes tests/test_gpu_rearranged.py holds 56 pairs to the tuned and eight to the checking form",
    }
    # event counts of the mapping kernels on this set (candidates per segment, share of one-run segments): from the committed
    # stats-build run (tools/map_stats.py with PA_SYNTH=rearranged), labelled as such -- the product build does not count
    efile = ROOT / "profiles" / "fragani_rearranged_events.json"
    if efile.is_file():
        out["mapping_events"] = {**json.loads(efile.read_text()), "source": f"profiles/{efile.name} (stats build, one batch of 2^17 query fragments; not measured inside this run)"}
    return out


def config1_files() -> dict:
    """BASELINE configs[0] on SURVEY.md 8(d)'s third clock, T_file: the four gzipped bacteria of the reference's own
    fixture set (tests/golden/bacterial_example = /root/reference/tests/fixtures/bacterial_example, copied as data) go from
    FASTA files on disk through this build's run driver -- the counterpart of `pyani-plus sourmash <dir> -d <db> --create-db`
    (pyani_plus/public_cli.py:598-639): file discovery, checksums, sketches, `.sig` cache, the worker's JSON column file,
    import, matrix cache -- to a SQLite database on disk.  IN THE RUN the identity and query-coverage matrices exported from
    that database are compared with the reference's own tests/fixtures/bacterial_example/matrices/sourmash_{identity,
    coverage}.tsv at the reference's tolerance (atol 2e-8, tests/snakemake/__init__.py:125-144), NULL pattern included."""
    import tempfile

    from pyani_plus_amd import rundb

    golden = ROOT / "tests" / "golden" / "bacterial_example"

    def read_matrix(path: Path):
        rows = [line.rstrip("\n").split("\t") for line in path.read_text().splitlines()]
        labels = rows[0][1:]
        assert [r[0] for r in rows[1:]] == labels, f"{path.name}: row and column labels differ"
        return labels, np.array([[float(v) if v not in ("", "nan", "NaN") else np.nan for v in r[1:]] for r in rows[1:]])

    seconds, phases = [], None
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        for rep in range(3):  # the first run of the process also pays for the context and the first touch of its buffers
            timings: dict = {}
            t0 = time.perf_counter()
            run = rundb.run_sourmash_hip(golden, tmp / f"run{rep}.sqlite", cache=tmp / f"cache{rep}", temp=tmp / f"work{rep}", timings=timings)
            seconds.append(time.perf_counter() - t0)
            phases = {k: round(float(v), 4) for k, v in timings.items()}
        n = len(run.fasta_hashes)
        written = rundb.export_run(tmp / "run2.sqlite", tmp / "export")
        by_name = {w.name: w for w in written}
        worst = 0.0
        for ours, theirs in ((f"{run.configuration.method}_identity.tsv", "sourmash_identity.tsv"), (f"{run.configuration.method}_query_cov.tsv", "sourmash_coverage.tsv")):
            got_labels, got = read_matrix(by_name[ours])
            want_labels, want = read_matrix(golden / "matrices" / theirs)
            if got_labels != want_labels or not np.array_equal(np.isnan(got), np.isnan(want)):
                raise SystemExit(f"PARITY FAILURE: config1_files: labels or NULL pattern of {ours} differ from the reference's {theirs}")
            diff = float(np.nanmax(np.abs(got - want)))
            worst = max(worst, diff)
            if not diff <= 2e-8:
                raise SystemExit(f"PARITY FAILURE: config1_files: {ours} differs from the reference's {theirs} by {diff}")
        rows = int(rundb.count_run_comparisons(rundb.connect_to_db(tmp / "run2.sqlite"), run))
    return {
        "workload": "BASELINE configs[0]: the 4 gzipped bacterial genomes of the reference's fixture set (5.4 MB of .gz, 17 Mb), k=31 scaled=1000, "
                    "FASTA files on disk -> .sig cache -> JSON column file -> SQLite database with all N^2 comparisons and the cached matrices (T_file, SURVEY.md 8d)",
        "genomes": n, "comparisons": rows, "seconds": min(seconds[1:]), "seconds_first_run_of_the_process": seconds[0], "seconds_all_runs": [round(x, 4) for x in seconds],
        "phases_s_last_run": phases, "matrices": "equal",
        "matrices_note": f"identity and query-coverage matrices exported from the database equal the reference's matrices/sourmash_{{identity,coverage}}.tsv "
                         f"(labels, NULL pattern, values within atol 2e-8: largest difference {worst:.3g}), asserted in this run",
        "pairs_per_s": rows / min(seconds[1:]),
    }


def also_fragani(engine, arena, args, n_total, lengths) -> dict:
    """BASELINE configs[3]: fastANI-style fragment ANI, k=16, fragLen=3000 (the restatement reproduces every fastANI value the reference holds)."""
    import oracle
    from pyani_plus_amd import _capi

    torch = engine.torch
    n = min(n_total, args.also_fragani_genomes)
    k, frag = 16, 3000
    sub_start = np.ascontiguousarray(arena.genome_start[: n + 1])
    from pyani_plus_amd.engine import DeviceArena

    sub = DeviceArena(arena.packed[: int(sub_start[-1]) // 16], arena.mask[: int(sub_start[-1]) // 32], sub_start, getattr(arena, "dirty", None))
    starts = sub_start[:-1].copy()
    lens = np.asarray(lengths[:n], dtype=np.uint32)
    genome = np.arange(n, dtype=np.uint32)
    engine.prof_reset()
    times = []
    for _rep in range(3):  # the first call also allocates the workspace
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total, matched, ident_sum = engine.fragani(sub, starts, lens, genome, k, frag)
        times.append(time.perf_counter() - t0)
        if _rep == 0:
            engine.prof_reset()
    sec = min(times[1:])
    prof = engine.prof_get()
    # one subject column, as the reference's worker asks for it (one process per column): the dictionary, the seed hits and
    # the bins of that genome only, results as a column
    col_times = []
    for _rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c_total, c_matched, c_sum = engine.fragani(sub, starts, lens, genome, k, frag, ref_range=(0, 1), columns_only=True)
        col_times.append(time.perf_counter() - t0)
    if not (np.array_equal(c_total, total) and np.array_equal(c_matched[:, 0], matched[:, 0]) and np.array_equal(c_sum[:, 0], ident_sum[:, 0])):
        raise SystemExit("PARITY FAILURE (fragment ANI): the subject column computed alone differs from the column of the all-against-all run")
    # The workspace: device bytes the context holds after the all-against-all call (it only grows: the peak), and what fresh
    # contexts hold after ONE subject column and after an eighth of the columns (the share of one rank of run_fastani_hip on
    # eight GPUs) -- the reference bounds a worker's memory by 500-query batches (pyani_plus/private_cli.py:1029-1033), here
    # the reference range does: the dictionary, the postings and the best-fragment table hold that range only.
    workspace = {"all_columns": engine.fragani_workspace()}
    from pyani_plus_amd.engine import HipEngine

    for label, r1 in (("one_column", 1), ("one_eighth_of_the_columns", max(1, n // 8))):
        other = HipEngine(0)
        try:
            other.fragani(sub, starts, lens, genome, k, frag, ref_range=(0, r1), columns_only=True)
            workspace[label] = {**other.fragani_workspace(), "reference_genomes": r1}
        finally:
            other.close()
        torch.cuda.empty_cache()
    workspace["what"] = ("device bytes of the fragment-ANI workspace as allocated (pa_fragani_workspace; buffers grow by a quarter when they grow): "
                         "~2.3 bytes per arena residue for all genomes' minimizers, ids and links + ~5.2 bytes per residue of the reference range for its dictionary, "
                         "postings and sort buffers + 8 bytes per seed hit of the largest query batch and ~1 GB of per-batch tables (DESIGN.md 4.5, Memory); "
                         "pa_fragani_set_workspace_cap bounds it")
    from pyani_plus_amd.methods.fastani_hip import fastani_mean

    ani = fastani_mean(ident_sum, matched)  # fastANI's own mean: a float sum by a float count
    # against itself a genome keeps (nearly) every fragment: fastANI's own self rows read 1820/1825, 1346/1347, ...,
    # because fragments can compete for one reference bucket of fragLen - 20 positions
    if not np.all(np.diag(matched) >= 0.99 * total):
        raise SystemExit("PARITY FAILURE (fragment ANI): a genome maps fewer than 99 % of its fragments onto itself")
    # CPU leg + parity.  The shape of the reference's own call, `fastANI --ql queries -r subject`
    # (pyani_plus/private_cli.py:1044-1063): ONE reference genome indexed once, a sample of query genomes mapped
    # against it on all the CPUs the quota allows.  The sample takes ten whole cycles of the species (10 x 40
    # genomes), so related pairs -- where nearly all the work is -- have the share they have in the N x N matrix (1 in 40).
    visible = len(os.sched_getaffinity(0))
    cores = max(1, min(visible, int(_capi.load_library().pa_host_cpu_budget())))
    n_q = min(n, 10 * args.species)
    ref_g = 0
    q_list = list(range(n_q))
    seqs = _ascii_genomes(engine, sub, q_list, lengths)
    contigs = [[x.tobytes()] for x in seqs]
    del seqs
    # Timed: the oracle's TUNED form (the window kept as the slide moves, as Mashmap's L2 keeps it; oracle/fragani_oracle.c) --
    # the checking form re-sorts every window and would flatter the device by another factor.  Checked: the device against the
    # tuned form on all the sampled queries, and against the checking form (the one pinned to the reference's fixtures) on the
    # first four cycles of the species.
    from oracle import pyoracle

    oracle.fragani_many(contigs[:1], contigs[ref_g], k, frag, 0.0, threads=1)  # warm
    pyoracle.fragani_set_fast(True)
    try:
        t0 = time.perf_counter()
        o_ani, o_m, o_t = oracle.fragani_many(contigs, contigs[ref_g], k, frag, 0.0, threads=cores)
        cpu_sec = time.perf_counter() - t0
    finally:
        pyoracle.fragani_set_fast(False)
    n_check = min(n_q, 4 * args.species)
    t0 = time.perf_counter()
    s_ani, s_m, s_t = oracle.fragani_many(contigs[:n_check], contigs[ref_g], k, frag, 0.0, threads=cores)
    check_sec = time.perf_counter() - t0
    bad = []
    for i, q in enumerate(q_list):
        if matched[q, ref_g] != o_m[i] or total[q] != o_t[i] or (o_m[i] and ani[q, ref_g] != o_ani[i]):
            bad.append((q, int(matched[q, ref_g]), int(o_m[i]), float(ani[q, ref_g]), float(o_ani[i])))
        if i < n_check and (matched[q, ref_g] != s_m[i] or total[q] != s_t[i] or (s_m[i] and ani[q, ref_g] != s_ani[i])):
            bad.append((q, int(matched[q, ref_g]), int(s_m[i]), float(ani[q, ref_g]), float(s_ani[i])))
    if bad:
        raise SystemExit(f"PARITY FAILURE (fragment ANI): {len(bad)} of {n_q} query genomes against genome {ref_g} differ from the oracle, first {bad[0]}")
    # the transposed direction of a few related pairs (reference index of another genome)
    g1 = min(n - 1, args.species)
    rev = oracle.fragani_pair(contigs[0], contigs[g1] if g1 < n_q else [_ascii_genomes(engine, sub, [g1], lengths)[0].tobytes()], k, frag, 0.0)
    if matched[0, g1] != rev[1] or (rev[1] and ani[0, g1] != rev[0]):
        raise SystemExit(f"PARITY FAILURE (fragment ANI): pair (0,{g1}) HIP {matched[0, g1]} {ani[0, g1]} vs oracle {rev[1]} {rev[0]}")
    del contigs
    related = int((~np.isnan(ani)).sum())
    out = {
        "workload": f"{n} synthetic {args.length / 1e6:g} Mb genomes, fastANI-style fragment ANI k={k} fragLen={frag} (BASELINE configs[3]), all ordered pairs in one pa_fragani call",
        "seconds_per_run": sec, "pairs_per_s": n * n / sec, "runs": 2, "first_run_seconds_incl_workspace_alloc": times[0],
        "pairs_with_mappings": related,
        "one_subject_column": {"seconds": min(col_times), "pairs_per_s": n / min(col_times),
                               "what": f"the {n} genomes against genome 0 alone (pa_fragani reference range [0, 1), columns only): what one per-column "
                               "worker of the reference's layout asks for; equals column 0 of the all-against-all run"},
        "phases_ms_per_run": {name: v[0] / 2 for name, v in prof.items() if name.startswith("frag")},
        "workspace_device_bytes": workspace,
        "cpu_baseline": {"value": n_q / cpu_sec, "unit": "pairs/s", "cores": cores, "kind": "port", "seconds": cpu_sec,
                         "sample": f"{n_q} query genomes (ten cycles of the {args.species} species: 1 related query in {args.species}, as in the N x N matrix) against "
                         f"ONE reference genome whose index is built once (oracle.fragani_many: the shape of `fastANI --ql queries -r subject`), {cores} OpenMP threads over the queries; "
                         "the oracle's tuned L2 (window kept while sliding: two Fenwick trees over the fragment's ranks), not the checking form",
                         "checking_form_pairs_per_s": n_check / check_sec, "checking_form_sample": f"the first {n_check} of those queries"},
        "parity": f"kept/total fragments and ANI of all {n_q} sampled queries against genome {ref_g} equal the oracle's tuned form, those of the first {n_check} its checking form, "
        "and one pair the other way round, oracle/fragani_oracle.c "
        "(integers and the float mean exact); the restatement itself reproduces all 25 fastANI rows and the pins the reference holds exactly (tests/test_gpu_fragani.py, tests/test_fastani_pins.py, DESIGN.md section 2)",
    }
    # rooflines of the two dominant kernels from the committed rocprofv3 counter passes (profiles/fragani_counters.json,
    # made by tools/pmc_fragani_to_json.py from tools/pmc_passes.sh runs of tools/bench_fragani.py) -- labelled as such
    cfile = ROOT / "profiles" / "fragani_counters.json"
    counters = None
    if cfile.is_file():
        try:
            counters = json.loads(cfile.read_text())
        except Exception:  # noqa: BLE001
            counters = None
    map_ms = prof.get("frag_map", (0.0, 0))[0] / 2
    seed_ms = prof.get("frag_seed", (0.0, 0))[0] / 2
    if map_ms > 0:
        mc = (counters or {}).get("map_segments_kernel", {})
        work = mc.get("work") or {}
        out["roofline"] = {
            "kernel": "map_segments_kernel", "bound": "valu-issue",
            "what": "vector instruction issue of one wave per (fragment, reference genome) segment; no HBM or MFMA roof applies "
            "(integer/index work on LDS-resident data).  frac = the vector instructions the dispatch NEEDS (work model) / the vector "
            "instructions it ISSUED (SQ_INSTS_VALU); the pipe itself is valu_busy full while the kernel runs",
            "frac": work.get("frac"), "frac_first_group": work.get("frac_first_group"), "frac_minimal_sort": work.get("frac_minimal_sort"),
            "work_model": work.get("work_model"),
            "algorithmic_units_per_dispatch": work.get("algorithmic_units_per_dispatch"),
            "valu_instructions_per_phase_per_segment": work.get("valu_instructions_per_phase_per_segment"),
            "share_of_a_round_needed": work.get("share_of_a_round_needed"),
            "algorithmic_valu_instructions_per_dispatch": work.get("algorithmic_valu_instructions_per_dispatch"),
            "algorithmic_valu_instructions_by_unit": work.get("algorithmic_valu_instructions_by_unit"),
            "counted_valu_instructions_per_dispatch": work.get("counted_valu_instructions_per_dispatch"),
            "valu_busy": mc.get("valu_busy"), "salu_busy": mc.get("salu_busy"),
            "wait_share_of_wave_time": mc.get("wait_share"), "waves_per_simd": mc.get("waves_per_simd"),
            "definition": "valu_busy = SQ_ACTIVE_INST_VALU x 4 / SIMDs / (GRBM_GUI_ACTIVE / 8); salu_busy = SQ_INSTS_SALU / CUs / (GRBM_GUI_ACTIVE / 8) "
            "(one scalar unit per CU); the two pipes issue side by side, so the busier one is the fraction of the issue roof",
            "source": mc.get("source", "no counter passes committed"), "avg_ms_per_run": map_ms, "share_of_run": map_ms / (sec * 1e3),
            "counted_fetch_bytes_per_dispatch": mc.get("fetch_bytes_per_dispatch_as_counted"), "counted_write_bytes_per_dispatch": mc.get("write_bytes_per_dispatch"),
            "counted_bytes_note": "the kernel's results are a few MB per dispatch; the counted writes are its register spills (scratch memory)",
        }
        sc = (counters or {}).get("map_sparse_kernel", {})
        if sc:
            out["roofline_sparse"] = {
                "kernel": "map_sparse_kernel", "bound": "valu-issue", "frac": sc.get("frac"), "frac_what": sc.get("frac_what"), "valu_busy": sc.get("valu_busy"),
                "valu_instructions_per_segment": sc.get("valu_instructions_per_segment"), "valu_instructions_per_state_evaluated": sc.get("valu_instructions_per_state_evaluated"),
                "events_per_dispatch": sc.get("events_per_dispatch"), "groups_of_begins_per_segment": sc.get("groups_per_segment"),
                "avg_ms_per_dispatch": sc.get("avg_ms_per_dispatch"), "waves_per_simd": sc.get("waves_per_simd"), "source": sc.get("source"),
            }
        ic = (counters or {}).get("minimizer_kernel", {})
        if ic:
            out["roofline_index"] = {
                "kernel": "minimizer_kernel<16>", "bound": "valu-issue", "frac": ic.get("frac"), "frac_what": ic.get("frac_what"), "valu_busy": ic.get("valu_busy"),
                "valu_instructions_per_position": ic.get("valu_instructions_per_position"), "hash_valu_instructions_per_position": ic.get("hash_valu_instructions_per_position"),
                "avg_ms_per_dispatch": ic.get("avg_ms_per_dispatch"), "waves_per_simd": ic.get("waves_per_simd"), "wait_share_of_wave_time": ic.get("wait_share"),
                "algorithmic_gbs": ic.get("algorithmic_gbs"), "algorithmic_bytes_note": ic.get("algorithmic_bytes_note"), "source": ic.get("source"),
                "avg_ms_per_run_of_the_index_phase": prof.get("frag_index", (0.0, 0))[0] / 2,
            }
        bc = (counters or {}).get("bucket_hits_kernel", {})
        out["roofline_seeding"] = {
            "kernel": "bucket_hits_staged_kernel", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "algorithmic_bytes_per_hit": bc.get("algorithmic_bytes_per_hit"), "counter_bytes_per_hit": bc.get("counter_bytes_per_hit"),
            "achieved": bc.get("algorithmic_gbs"), "frac": (bc.get("algorithmic_gbs") or 0) / HBM_PEAK_GBS if bc.get("algorithmic_gbs") else None,
            "traffic_gbs": bc.get("counter_gbs"), "traffic_over_algorithmic": bc.get("traffic_over_algorithmic"),
            "fetch_calibration": bc.get("fetch_calibration"),
            "counter_bytes_note": "FETCH_SIZE turned into 64-byte lines moved with the factor measured for this access pattern (runs of 2- and 8-byte items at unrelated "
                                  "places: tools/fetch_calib, profiles/r06_fetch_calibration.txt) + WRITE_SIZE (exact); one calibrated number, not two alternatives",
            "source": bc.get("source", "no counter passes committed"),
            "avg_ms_per_run_of_the_seeding_phase": seed_ms,
        }
    return out


# --------------------------------------------------------------------------- one rank
def run_rank(args, fresh_fragani: dict | None = None) -> None:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    # Everything the libraries below print (RCCL's version banner goes to stdout through C stdio) is sent to stderr:
    # this process's stdout carries the one JSON line and nothing else.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from pyani_plus_amd import _capi
    from pyani_plus_amd.distributed import shard_bounds, shard_bounds_by_cost, sharded_pair_step
    from pyani_plus_amd.engine import HipEngine, ani_host
    from pyani_plus_amd.synth import mixed_lengths, synth_arena_torch

    # PA_BENCH_BACKEND=gloo is a plumbing check for boxes with fewer GPUs than ranks: ranks share
    # GPUs and the collectives run on host tensors.  The measured configuration is always nccl (RCCL).
    backend = os.environ.get("PA_BENCH_BACKEND", "nccl")
    # PA_BENCH_FORCE_DIST=1 runs the distributed code path (process group, all-gather, column tile)
    # even with one rank, so the RCCL calls can be exercised on a single-GPU box.
    dist_path = world > 1 or os.environ.get("PA_BENCH_FORCE_DIST") == "1"
    n_dev = torch.cuda.device_count()
    if dist_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            if local_rank >= n_dev:
                raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but {n_dev} are visible (PA_BENCH_BACKEND=gloo shares GPUs for a plumbing check)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(1, n_dev)
            dist.init_process_group(backend, rank=rank, world_size=world)
    engine = HipEngine(local_rank)

    n_total = args.genomes or (1000 if world == 1 else 1250 * world)
    # sketch shards: contiguous genome ranges balanced by length.  Pair tiles: each rank's subject columns are its
    # own genomes (the dictionary build then overlaps the all-gather, and with mixed lengths its cost -- one insert
    # per subject hash -- is balanced the way the hashing is)
    lengths = mixed_lengths(n_total) if args.mixed_lengths else [args.length] * n_total
    bounds = shard_bounds_by_cost(lengths, world) if args.mixed_lengths else shard_bounds(n_total, world)
    g0, g1 = bounds[rank]
    shard_sizes = [b - a for a, b in bounds]
    c0, c1 = g0, g1  # a rank's subject columns are its own genomes: the dictionary build follows sketch size and overlaps the all-gather
    arena = synth_arena_torch(engine, g1 - g0, lengths[g0:g1], n_species=args.species, genome_offset=g0)

    bottom = args.sketch_mode == "bottom"
    overlap = os.environ.get("PA_BENCH_NO_OVERLAP") != "1"

    # The timed step ends in the reference's own doubles: u32 counts -> pinned host memory -> glibc pow (bit-identical to
    # sourmash's f64, where the device's pow is within 1 ulp).  Inputs are resident in HBM when the clock starts (the
    # upload is timed under t_e2e, beside `value`, never in it).  The device-pow form of the step is timed after the
    # contract's K steps as `value_t_dev`.
    strict_out = not bottom
    if strict_out:
        h_counts = torch.empty((n_total, c1 - c0), dtype=torch.int32).pin_memory()
        h_ident = torch.empty((n_total, c1 - c0), dtype=torch.float64).pin_memory()
        h_cov = torch.empty_like(h_ident).pin_memory()
        h_null = np.empty((n_total, c1 - c0), dtype=np.uint8)

    def step(device_pow: bool = not strict_out):
        sk_local = engine.sketch_bottom(arena, args.kmer, args.bottom_m) if bottom else engine.sketch(arena, args.kmer, args.scaled)
        if dist_path and not bottom:
            sk, counts = sharded_pair_step(engine, torch, dist, sk_local, shard_sizes, (0, n_total), (c0, c1), backend=backend, overlap=overlap)
        elif dist_path:
            from pyani_plus_amd.distributed import allgather_sketches
            from pyani_plus_amd.engine import DeviceSketches

            sizes = sk_local.off[1:] - sk_local.off[:-1]
            if backend == "nccl":
                hashes, off, off_host = allgather_sketches(torch, dist, sk_local.hashes, sizes, shard_sizes)
            else:
                hashes, off, off_host = allgather_sketches(torch, dist, sk_local.hashes[: max(1, sk_local.total)].cpu(), sizes.cpu(), shard_sizes)
                hashes, off = hashes.to(engine.device), off.to(engine.device)
            sk = DeviceSketches(hashes, off, n_total, int(off_host[-1]), off_host)
            counts = None
        else:
            sk, counts = sk_local, None
        if bottom:
            counts, denom = engine.pair_mash(sk, args.bottom_m, (0, n_total), (c0, c1))
            ident = engine.ani_mash(counts, denom, args.kmer)
            return sk_local, sk, counts, ident, ident
        if counts is None:
            counts = engine.pair_counts(sk, (0, n_total), (c0, c1))
        if not device_pow:
            # the reference's doubles: counts to pinned host memory, glibc pow on the host's threads (pa_ani_host), the two
            # f64 matrices of this rank's column tile in pinned host memory -- what the JSON / database boundary is handed
            h_counts.copy_(counts, non_blocking=True)
            sizes = sk.sizes()
            torch.cuda.synchronize()
            ani_host(h_counts.numpy().view(np.uint32), sizes, sizes[c0:c1], args.kmer, symmetric=(c0 == 0 and c1 == n_total),
                     out=(h_ident.numpy(), h_cov.numpy(), h_null))
            return sk_local, sk, counts, h_ident, h_cov
        ident, cov = engine.ani(counts, sk, args.kmer, (0, n_total), (c0, c1))
        return sk_local, sk, counts, ident, cov

    def fence():
        torch.cuda.synchronize()
        if dist_path:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    engine.prof_enable(True)
    engine.prof_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist_path:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=engine.device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # every rank cross-checks a block of its column tile with the independent merge kernel
        nq_chk, ns_chk = min(n_total, 256), min(c1 - c0, 64)
        if not bottom:
            chk = engine.pair_counts(out[1], (0, nq_chk), (c0, c0 + ns_chk), algo=_capi.PA_PAIRS_MERGE)
            if not torch.equal(chk, out[2][:nq_chk, :ns_chk]):
                raise SystemExit(f"PARITY FAILURE on rank {rank}: bit-row and merge counts differ")
    prof = engine.prof_get()
    engine.prof_enable(False)
    t_dev_elapsed = None
    if strict_out:  # the same K steps with the transform on the device (f64 pow, <= 1 ulp): matrices stay in HBM
        step(True)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out_dev = step(True)
        fence()
        t_dev_elapsed = time.perf_counter() - t0
        if dist_path:
            tmax = torch.tensor([t_dev_elapsed], dtype=torch.float64, device=engine.device if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            t_dev_elapsed = float(tmax.item())
        # the two transforms agree to 1 ulp (and the NULL pattern exactly) on this rank's tile
        d_ident, d_cov = out_dev[3].cpu().numpy(), out_dev[4].cpu().numpy()
        nul = h_null.view(np.bool_)
        if not (np.array_equal(np.isnan(d_ident), nul) and np.allclose(d_ident[~nul], h_ident.numpy()[~nul], rtol=2.3e-16, atol=0)
                and np.allclose(d_cov[~nul], h_cov.numpy()[~nul], rtol=2.3e-16, atol=0)):
            raise SystemExit(f"PARITY FAILURE on rank {rank}: device-pow and host-libm ANI matrices differ by more than 1 ulp")
        del out_dev, d_ident, d_cov
    sk_local, sk, counts, ident, cov = out
    # BASELINE configs[4] / SURVEY.md 8(d): how evenly the shards load the GPUs -- device-busy time of each
    # rank's own kernels per step (HIP events around the phases; waits for other ranks are not in it)
    busy = sum(v[0] for v in prof.values()) / max(1, args.steps)
    rank_busy = [busy]
    if dist_path:
        tb = torch.tensor([busy], dtype=torch.float64, device=engine.device if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(tb) for _ in range(world)]
        dist.all_gather(gathered, tb)
        rank_busy = [float(x.item()) for x in gathered]

    # roofline of the dominant kernel (k-mer hash + filter), this rank's launches
    hash_ms, hash_launches = prof["kmer_hash"]
    n_local = g1 - g0
    local_hashes = int(sk_local.total)
    # per genome read ceil(L/4) B of 2-bit input + write 8*|S| B of sketch
    alg_bytes = sum((x + 3) // 4 for x in lengths[g0:g1]) + 8 * local_hashes
    per_launch_s = (hash_ms / max(1, hash_launches)) * 1e-3
    achieved = alg_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
    # HBM traffic and SQ counters cannot be read from inside the run: they come from the committed rocprofv3
    # passes of tools/pmc_passes.sh (profiles/hash_counters.json), and are labelled as such
    counters = None
    cfile = ROOT / "profiles" / "hash_counters.json"
    # ... of the profiled workload only: kmer_hash_kernel<31>, scaled mode, 1 000 genomes of 5 Mb per launch
    profiled = args.kmer == 31 and not bottom and not args.mixed_lengths and args.length == 5_000_000 and (g1 - g0) == 1000
    if profiled and cfile.is_file():
        try:
            counters = json.loads(cfile.read_text())
        except Exception:
            counters = None

    # a device copy on the same box, so fractions can be read against the nominal
    # 8 TB/s and against what this GPU actually streams (1 GiB read + 1 GiB written per copy)
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 30, dtype=torch.uint8, device=engine.device)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    result = None
    if rank == 0:
        windows = max(1.0, sum(lengths[g0:g1]) / 64.0)  # wave-steps of one launch: one window per lane
        clock_ghz = (counters or {}).get("effective_clock_ghz")
        valu = None if args.kmer != 31 or bottom else {  # the instruction mix below is that of kmer_hash_kernel<31>
            "instr_per_window_static": 92.5,
            "model_cycles_per_wave_step": 24.6875 * 2.6 + 67.8125 * 4.35,
            "issue_cost_cycles": {"plain_vop2_add_logic": 2.6, "everything_else": 4.35, "source": "profiles/r02_ubench_valu_gfx950.txt"},
            "simds": SIMDS,
            "note": "model = static instruction mix x measured per-instruction issue costs; measured/model near 1 means the "
            "kernel runs at the VALU issue limit of its instruction stream",
        }
        if valu is not None and clock_ghz:
            valu["clock_ghz_measured"] = clock_ghz
            valu["clock_source"] = (counters or {}).get("clock_source")
            valu["measured_cycles_per_wave_step"] = per_launch_s * clock_ghz * 1e9 * SIMDS / windows
        if valu is not None and counters and counters.get("valu_busy_pct") is not None:
            valu["valu_busy_pct_from_counters"] = counters["valu_busy_pct"]
            valu["counters"] = counters.get("sq")
        result = {
            "metric": "pairwise genome comparisons/sec (N x N ANI matrix)",
            "value": n_total * n_total * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "value_clock": ("2-bit genomes resident in HBM -> the reference's own f64 identity/cov_query matrices in pinned host memory "
                            "(u32 counts copied back, glibc pow on the host's threads: bit-identical to sourmash's doubles)" if strict_out else
                            "2-bit genomes resident in HBM -> f64 Mash-ANI matrix in HBM"),
            "value_clock_note": "inputs are resident when the clock starts, as the bench contract prescribes; the same step from packed "
            "genomes in pinned HOST memory (SURVEY.md 8d T_e2e, PCIe-bound: 1.25 GB per step) is value_e2e_strict, the step "
            "with the transform left on the device (f64 pow, <= 1 ulp, matrices in HBM) is value_t_dev",
            "ani_transform_in_value": "host glibc pow (pa_ani_host), one pow per ordered pair: the reference's doubles" if strict_out else "device",
            "value_t_dev": (n_total * n_total * args.steps / t_dev_elapsed) if t_dev_elapsed else None,
            "ms_per_step_t_dev": (t_dev_elapsed / args.steps * 1e3) if t_dev_elapsed else None,
            "config": {
                "workload": f"{n_total} synthetic "
                + ("100 kb-10 Mb (log-uniform)" if args.mixed_lengths else f"{args.length / 1e6:g} Mb")
                + (f" genomes, k={args.kmer} bottom-m={args.bottom_m} MinHash sketch + NxN Mash-Jaccard ANI (parity unpinned)" if bottom
                   else f" genomes, k={args.kmer} scaled={args.scaled} FracMinHash sketch + NxN containment ANI"),
                "sketch_mode": args.sketch_mode,
                "genomes": n_total,
                "genomes_per_gpu": n_local,
                "length": args.length,
                "k": args.kmer,
                "scaled": args.scaled,
                "species": args.species,
                "mean_sketch_size": local_hashes / max(1, n_local),
                "parallelism": f"genome shards + {'RCCL' if backend == 'nccl' else backend} sketch all-gather"
                + (" overlapped with the local dictionary build" if overlap and not bottom else "")
                + f" + subject-column tiles x{world}" if dist_path else "single GPU",
            },
            "rccl_ranks": (dist.get_world_size() if dist_path else 0),
            "collective_backend": (backend if dist_path else None),
            "roofline": {
                "kernel": f"kmer_hash_kernel<{args.kmer}>" if args.kmer <= 32 else f"kmer_hash_long_kernel (k={args.kmer})",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (counters or {}).get("hbm_bytes_per_launch"),
                "traffic_source": (counters or {}).get("traffic_source", None if profiled else "no counter passes committed for this workload"),
                "measured_copy_gbs": copy_gbs,
                "frac_of_measured_copy": achieved / copy_gbs if copy_gbs else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": per_launch_s * 1e3,
                "note": "kernel is integer-VALU bound (MurmurHash3 per window), see DESIGN.md",
                "valu": valu,
            },
            "phases_ms_per_step": {k: v[0] / args.steps for k, v in prof.items() if v[1]},
            "shard_balance": {
                "busy_ms_per_step_by_rank": rank_busy,
                "max_over_mean": max(rank_busy) / (sum(rank_busy) / len(rank_busy)) if sum(rank_busy) > 0 else None,
                "pair_tiles": pair_tile_balance(sk.sizes(), bounds) if sk is not None and getattr(sk, "n", 0) == n_total else None,
            },
            "device": engine.device_info()["name"],
            "reference_tools": reference_tools_probe(),
        }
        cb = None
        if world == 1 and not dist_path and not args.no_cpu_baseline and not bottom:
            cb = cpu_baseline(engine, arena, sk, args, n_total, lengths)
            n_pair = cb.pop("_n_pair")
            cpu_counts = cb.pop("_cpu_counts")
            cpu_ani = cb.pop("_cpu_ani")
            gpu_counts = counts[:n_pair, :n_pair].cpu().numpy().view(np.uint32)
            if not np.array_equal(gpu_counts, cpu_counts):
                raise SystemExit("PARITY FAILURE: pair counts differ between HIP and oracle on the sample block")
            result["cpu_baseline"] = cb
            result["parity_checked"] = (f"sketches of {'all' if not cb['extrapolated'] else 'the sampled'} genomes and "
                                        f"{'all ' if not cb['extrapolated'] else 'a block of '}{n_pair}x{n_pair} pair counts equal the oracle")
        else:
            result["cpu_baseline"] = None
        if world == 1 and not dist_path and not args.no_pcie:
            # T_e2e (packed genomes in pinned host RAM -> f64 identity & cov_query in host RAM), reported
            # beside `value`, never inside it.  "plain": the whole arena (bases + mask bitmap) is copied, then the resident
            # step runs.  "streamed": the mask crosses as runs and the bases go up in chunks behind the hash kernel
            # (pa_sketch_streamed).  "strict": streamed, but the u32 counts come back and the host's libm pow makes the
            # matrices (pa_ani_host on host threads) -- the transform that is bit-identical to the reference's.
            h_packed = arena.packed.cpu().pin_memory()
            h_mask = arena.mask.cpu().pin_memory()
            e_ident = torch.empty((n_total, n_total), dtype=torch.float64).pin_memory()
            e_cov = torch.empty_like(e_ident).pin_memory()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            arena.packed.copy_(h_packed, non_blocking=True)
            arena.mask.copy_(h_mask, non_blocking=True)
            o = step()  # (ends in the two f64 matrices in pinned host memory already)
            torch.cuda.synchronize()
            plain_ms = (time.perf_counter() - t0) * 1e3
            t_e2e = {
                "definition": "packed genomes in pinned host memory -> f64 identity and cov_query matrices in host memory (SURVEY.md 8d T_e2e); never part of value",
                "plain": {"ms_per_step": plain_ms, "pairs_per_s": n_total * n_total / (plain_ms * 1e-3),
                          "h2d_bytes": int(h_packed.numel() * 4 + h_mask.numel() * 4), "d2h_bytes": int(n_total * n_total * 4 if strict_out else 2 * e_ident.numel() * 8),
                          "ani_transform": "host glibc pow" if strict_out else "device pow"},
            }
            if not bottom:
                from pyani_plus_amd.engine import PinnedArena, mask_runs

                run_start, run_len = mask_runs(h_mask.numpy().view(np.uint32), int(arena.genome_start[-1]))
                pinned = PinnedArena(h_packed, run_start, run_len, np.ascontiguousarray(arena.genome_start, dtype=np.uint64))
                best, runs_e2e = None, []
                n_e2e = max(3, min(args.steps, 10))
                for it in range(n_e2e + 1):  # the first pass is a warm-up (buffers of the streamed upload)
                    arena.packed.zero_()
                    arena.mask.zero_()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _dev, sk2 = engine.sketch_streamed(pinned, args.kmer, args.scaled, arena=arena)
                    c2 = engine.pair_counts(sk2, (0, n_total), (c0, c1))
                    i2, v2 = engine.ani(c2, sk2, args.kmer, (0, n_total), (c0, c1))
                    e_ident.copy_(i2, non_blocking=True)
                    e_cov.copy_(v2, non_blocking=True)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) * 1e3
                    if it:
                        runs_e2e.append(ms)
                        best = ms if best is None else min(best, ms)
                mean_streamed = sum(runs_e2e) / len(runs_e2e)
                if not torch.equal(c2, o[2]):
                    raise SystemExit("PARITY FAILURE: streamed and resident pair counts differ")
                dev_ident, dev_cov = e_ident.numpy().copy(), e_cov.numpy().copy()
                t_e2e["streamed"] = {
                    "ms_per_step": mean_streamed, "pairs_per_s": n_total * n_total / (mean_streamed * 1e-3), "best_ms": best, "runs": len(runs_e2e),
                    "h2d_bytes": int(h_packed.numel() * 4 + 16 * len(run_start)), "mask_runs": int(len(run_start)),
                    "ani_transform": "device pow",
                    "note": "mask as runs, 64 MB chunks uploaded on a copy stream behind the hash kernel; mean of the timed runs; counts equal the resident step's",
                }
                # strict: counts -> pinned host -> libm pow on host threads -> the same two pinned f64 matrices
                e_counts = torch.empty((n_total, n_total), dtype=torch.int32).pin_memory()
                e_null = np.empty((n_total, n_total), dtype=np.uint8)
                best_s, best_pow, runs_strict, runs_pow = None, None, [], []
                for _ in range(n_e2e):
                    arena.packed.zero_()
                    arena.mask.zero_()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _dev, sk3 = engine.sketch_streamed(pinned, args.kmer, args.scaled, arena=arena)
                    c3 = engine.pair_counts(sk3, (0, n_total), (c0, c1))
                    e_counts.copy_(c3, non_blocking=True)
                    sizes3 = sk3.sizes()
                    torch.cuda.synchronize()
                    tp = time.perf_counter()
                    ani_host(e_counts.numpy().view(np.uint32), sizes3, sizes3, args.kmer, symmetric=True,
                             out=(e_ident.numpy(), e_cov.numpy(), e_null))
                    t1 = time.perf_counter()
                    runs_strict.append(t1 - t0)
                    runs_pow.append(t1 - tp)
                    if best_s is None or (t1 - t0) < best_s:
                        best_s, best_pow = t1 - t0, t1 - tp
                mean_s, mean_pow = sum(runs_strict) / len(runs_strict), sum(runs_pow) / len(runs_pow)
                # the strict matrices are the reference's numbers; the device-pow ones must sit within 1 ulp of them
                s_ident, s_cov = e_ident.numpy(), e_cov.numpy()
                nul = e_null.view(np.bool_)
                if not (np.array_equal(np.isnan(dev_ident), nul) and np.allclose(dev_ident[~nul], s_ident[~nul], rtol=2.3e-16, atol=0)
                        and np.allclose(dev_cov[~nul], s_cov[~nul], rtol=2.3e-16, atol=0)):
                    raise SystemExit("PARITY FAILURE: device-pow and host-libm ANI matrices differ by more than 1 ulp")
                if cb is not None:
                    o_ident, o_cov, o_null = cpu_ani
                    if not (np.array_equal(o_null, nul[:n_pair, :n_pair]) and np.array_equal(o_ident[~o_null], s_ident[:n_pair, :n_pair][~o_null])
                            and np.array_equal(o_cov[~o_null], s_cov[:n_pair, :n_pair][~o_null])):
                        raise SystemExit("PARITY FAILURE: strict ANI block differs from the oracle's doubles")
                n_non_null = int((~nul).sum())
                # the same transform when EVERY pair is non-NULL (a single-species set): counts forced to >= 1
                dense = np.maximum(e_counts.numpy().view(np.uint32), 1)
                tmp = (np.empty_like(s_ident), np.empty_like(s_cov), np.empty_like(e_null))
                ani_host(dense, sizes3, sizes3, args.kmer, symmetric=True, out=tmp)
                tp = time.perf_counter()
                ani_host(dense, sizes3, sizes3, args.kmer, symmetric=True, out=tmp)
                dense_pow_ms = (time.perf_counter() - tp) * 1e3
                del dense, tmp
                t_e2e["strict"] = {
                    "ms_per_step": mean_s * 1e3, "pairs_per_s": n_total * n_total / mean_s, "best_ms": best_s * 1e3, "runs": len(runs_strict),
                    "host_pow_ms": mean_pow * 1e3, "non_null_pairs": n_non_null, "d2h_bytes": int(e_counts.numel() * 4),
                    "host_pow_ms_if_all_pairs_non_null": dense_pow_ms,
                    "ms_per_step_if_all_pairs_non_null": (mean_s - mean_pow) * 1e3 + dense_pow_ms,
                    "ani_transform": "host glibc pow on host threads (pa_ani_host, one pow per ordered pair): bit-identical to the reference's doubles",
                    "over_streamed": mean_s * 1e3 / mean_streamed - 1.0,
                    "note": "mean of the timed runs; matrices equal the device-pow ones to 1 ulp" + ("; the sample block equals the oracle's doubles exactly" if cb is not None else ""),
                }
                del e_counts
            result["t_e2e"] = t_e2e
            if "strict" in t_e2e:
                # the number to quote when the question is "the reference's doubles, from packed genomes in host memory":
                # T_e2e with the bit-identical host-libm transform (SURVEY.md 8d names T_e2e as the headline clock)
                result["value_e2e_strict"] = t_e2e["strict"]["pairs_per_s"]
                result["value_e2e_strict_note"] = ("pairs/s on the T_e2e clock with the bit-identical transform (t_e2e.strict, mean of "
                                                   f"{t_e2e['strict']['runs']} runs): `value`'s step with the genomes starting in pinned host memory instead of HBM")
                if cb is not None:
                    result["value_e2e_strict_vs_cpu_baseline"] = result["value_e2e_strict"] / cb["value"]
                    result["value_vs_cpu_baseline"] = result["value"] / cb["value"]
                    # the same figures inside `cpu_baseline`, which the driver's record of this line keeps whole: the ratio to
                    # quote is the one on SURVEY.md 8(d)'s headline clock, T_e2e with the reference's own doubles
                    cb["gpu_over_cpu"] = {
                        "t_e2e_strict_pairs_per_s": result["value_e2e_strict"], "t_e2e_strict_ms_per_step": t_e2e["strict"]["ms_per_step"],
                        "t_e2e_strict_over_cpu": result["value_e2e_strict"] / cb["value"], "value_over_cpu": result["value"] / cb["value"],
                        "quote": "t_e2e_strict_over_cpu: packed genomes in pinned host memory -> the reference's doubles in host memory (PCIe-inclusive); "
                                 "value_over_cpu has the genomes resident in HBM when the clock starts, as the bench contract defines `value`",
                    }
            del h_packed, h_mask, e_ident, e_cov
        if world == 1 and not dist_path and not args.no_also and not bottom and not args.mixed_lengths:
            also = {}
            wanted = [x for x in args.also.split(",") if x]

            def extra(name, fn, *fn_args):
                """One extra run; a failure there is recorded in its own entry and never costs the headline line."""
                try:
                    also[name] = fn(*fn_args)
                except (Exception, SystemExit) as err:  # noqa: BLE001
                    also[name] = {"error": f"{type(err).__name__}: {err}"}
                torch.cuda.empty_cache()

            engine.prof_enable(True)
            # order: the runs that reuse the resident arena first, then the ones that build their own
            if "bottom" in wanted:
                extra("bottom_m", also_bottom, engine, arena, args, n_total, lengths)
            if "k51" in wanted:
                extra("long_kmer_k51", also_long_kmer, engine, arena, args, n_total, lengths)
            if "fragani" in wanted:
                extra("fragment_ani", also_fragani, engine, arena, args, n_total, lengths)
                if fresh_fragani is not None and isinstance(also.get("fragment_ani"), dict):
                    fresh_fragani["what"] = ("a child process started before this one touched the GPU: torch import, the synthetic arena, then ONE "
                                             "all-columns pa_fragani call as the first use of its workspace (first_call_seconds: what a fresh "
                                             "fastANI-hip worker pays, first-touch page mapping of the workspace included) and the same call again")
                    also["fragment_ani"]["fresh_process"] = fresh_fragani
            engine.prof_enable(False)
            del out, sk_local, sk, counts, ident, cov
            arena = None
            torch.cuda.empty_cache()
            if "fragani" in wanted and "rearranged" in wanted:
                engine.prof_enable(True)
                extra("fragment_ani_rearranged", also_fragani_rearranged, engine, args, n_total)
                engine.prof_enable(False)
            if "mixed" in wanted:
                extra("mixed_lengths", also_mixed, engine, args)
            if "n10000" in wanted:
                extra("n10000_one_gpu", also_n10000, engine, args)
            result["also"] = also
            if "config1" in wanted:  # last key of the line: the driver's record keeps the line's tail
                try:
                    result["config1_files"] = config1_files()
                except (Exception, SystemExit) as err:  # noqa: BLE001 - never costs the headline line
                    result["config1_files"] = {"error": f"{type(err).__name__}: {err}", "matrices": "not compared"}
    if rank == 0 and world > 1 and os.environ.get("PA_BENCH_NO_BASIS") != "1":
        # The same workload -- all n_total genomes -- on ONE GPU, measured here and now by rank 0 while the other ranks
        # wait at the barrier below: value / strong_basis.one_gpu_pairs_per_s is the speed-up of the N-GPU run over one
        # GPU on the SAME work.  (The per-N values of a SCALE run are not comparable with each other: the workload
        # grows with N -- 1 250 genomes per GPU -- and pairs/s grows with the genome count on one GPU already.)
        try:
            del out, sk_local, sk, counts, ident, cov
            torch.cuda.empty_cache()
            full = synth_arena_torch(engine, n_total, lengths, n_species=args.species)

            def one_gpu_step():
                if bottom:
                    sk1 = engine.sketch_bottom(full, args.kmer, args.bottom_m)
                    c1_, d1 = engine.pair_mash(sk1, args.bottom_m, (0, n_total), (0, n_total))
                    return engine.ani_mash(c1_, d1, args.kmer)
                sk1 = engine.sketch(full, args.kmer, args.scaled)
                cnt1 = engine.pair_counts(sk1)
                return engine.ani(cnt1, sk1, args.kmer)

            sec1, _ = _time_steps(torch, one_gpu_step, 2, warmup=1)
            result["strong_basis"] = {
                "genomes": n_total, "one_gpu_ms_per_step": sec1 * 1e3, "one_gpu_pairs_per_s": n_total * n_total / sec1, "steps": 2,
                "note": "the same workload on one GPU (rank 0, after the timed steps, the other ranks idle)",
            }
            result["speedup_vs_one_gpu_same_workload"] = result["value"] / result["strong_basis"]["one_gpu_pairs_per_s"]
            del full
        except (Exception, SystemExit) as err:  # noqa: BLE001 - never costs the headline line
            result["strong_basis"] = {"error": f"{type(err).__name__}: {err}"}
    if dist_path:
        dist.barrier()
        dist.destroy_process_group()
    engine.close()
    import ctypes

    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(result), flush=True)


def main():
    args = parse_args()
    if args.dry_run_plan:
        print(json.dumps(dry_run_plan(args), indent=1))
        return
    if args.fresh_fragani_child:
        fresh_fragani_child(args)
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args))
    # before this process imports torch or touches the GPU: the fragment-ANI call of a fresh process (also.fragment_ani.fresh_process)
    fresh = None
    if (args.gpus == 1 and "RANK" not in os.environ and os.environ.get("PA_BENCH_FORCE_DIST") != "1" and not args.no_also and not args.no_fresh_child
            and "fragani" in args.also.split(",") and args.sketch_mode == "scaled" and not args.mixed_lengths):
        fresh = run_fresh_fragani_child(args)
    run_rank(args, fresh)


if __name__ == "__main__":
    main()
