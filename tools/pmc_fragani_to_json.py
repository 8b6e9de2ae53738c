#!/usr/bin/env python3
"""profiles/fragani_counters.json from the summaries of the rocprofv3 counter passes on the fragment-ANI kernels
(tools/pmc_passes.sh <tag> <kernel> tools/bench_fragani.py 1000 0 interleaved 78 -> gpurun_out/<tag>_pmc/summary.txt,
copied to profiles/: the benchmark's 1 000 genomes, one batch of 2^17 query fragments per repetition).  bench.py copies
these figures into `also.fragment_ani.roofline*`, labelled as coming from these passes.

    python tools/pmc_fragani_to_json.py profiles/r04_pmc_map_segments_summary.txt profiles/r04_pmc_bucket_hits_summary.txt \
        <seed hits per bucket_hits dispatch> [workload label]
"""
import json
import re
import sys
from pathlib import Path

SIMDS, CUS, XCDS = 1024, 256, 8


def parse(path: Path) -> dict:
    out = {}
    for line in path.read_text().splitlines():
        m = re.match(r"\s+(\S+)\s+mean per dispatch\s+(\S+)", line)
        if m:
            out.setdefault(m.group(1), float(m.group(2)))
        m = re.match(r"\s+duration_ms .*: mean (\S+)", line)
        if m:
            out.setdefault("duration_ms", float(m.group(1)))
    return out


def main() -> None:
    map_file, bucket_file = Path(sys.argv[1]), Path(sys.argv[2])
    hits_per_dispatch = float(sys.argv[3]) if len(sys.argv) > 3 else None
    what = sys.argv[4] if len(sys.argv) > 4 else "tools/bench_fragani.py 1000 0 interleaved 78 (the benchmark's 1 000 genomes, one batch of 2^17 query fragments)"
    m, b = parse(map_file), parse(bucket_file)
    cycles = m["GRBM_GUI_ACTIVE"] / XCDS
    out = {
        "map_segments_kernel": {
            "source": f"rocprofv3 --pmc passes of {what} ({map_file.name}); not measured inside this run",
            "valu_busy": m["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / cycles,
            "salu_busy": m["SQ_INSTS_SALU"] / CUS / cycles,
            "valu_instructions": m["SQ_INSTS_VALU"], "salu_instructions": m["SQ_INSTS_SALU"],
            "wait_share": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
            "waves_per_simd": m["SQ_WAVE_CYCLES"] * 4 / SIMDS / cycles,  # SQ_* cycle counters are in quad-cycles
            "avg_ms_per_dispatch": m["duration_ms"],
            # the kernel's own results are a few MB per dispatch: what FETCH_SIZE / WRITE_SIZE count beyond the stretches it
            # reads is its register spill traffic (scratch memory, 72 bytes per lane in round 4)
            "fetch_bytes_per_dispatch_as_counted": m.get("FETCH_SIZE", 0.0) * 1024, "write_bytes_per_dispatch": m.get("WRITE_SIZE", 0.0) * 1024,
        },
    }
    cyc_b = b["GRBM_GUI_ACTIVE"] / XCDS
    # FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  The guide's gfx950 rule (FETCH_SIZE reports half the bytes) is
    # calibrated for wide coalesced streaming reads; this kernel reads 2- and 8-byte items scattered over short lists,
    # "other access widths are uncalibrated": both readings are given, the counted one first.
    fetch = b["FETCH_SIZE"] * 1024 * 2
    write = b["WRITE_SIZE"] * 1024
    entry = {
        "source": f"rocprofv3 --pmc passes of {what} ({bucket_file.name}); FETCH_SIZE as counted and doubled (the gfx950 rule is "
        "calibrated for wide streaming reads only); not measured inside this run",
        "fetch_bytes_per_dispatch_as_counted": fetch / 2, "write_bytes_per_dispatch": write, "avg_ms_per_dispatch": b["duration_ms"],
        "counter_gbs": (fetch / 2 + write) / (b["duration_ms"] * 1e-3) / 1e9,
        "counter_gbs_fetch_doubled": (fetch + write) / (b["duration_ms"] * 1e-3) / 1e9,
        "valu_busy": b["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / cyc_b,
        "wait_share": b["SQ_WAIT_ANY"] / b["SQ_WAVE_CYCLES"],
    }
    if hits_per_dispatch:
        entry["seed_hits_per_dispatch"] = hits_per_dispatch
        entry["algorithmic_bytes_per_hit"] = 18.0  # the posting's 2-byte genome (counting pass) + the 8-byte posting (scatter pass) + one 8-byte hit written
        entry["counter_bytes_per_hit"] = (fetch / 2 + write) / hits_per_dispatch  # FETCH_SIZE as counted (the x2 rule is calibrated for wide streaming reads only)
        entry["counter_bytes_per_hit_fetch_doubled"] = (fetch + write) / hits_per_dispatch
        entry["algorithmic_gbs"] = 18.0 * hits_per_dispatch / (b["duration_ms"] * 1e-3) / 1e9
    out["bucket_hits_kernel"] = entry
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
