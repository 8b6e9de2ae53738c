#!/usr/bin/env python3
"""profiles/fragani_counters.json from the summaries of the rocprofv3 counter passes on the fragment-ANI kernels
(tools/pmc_passes.sh <tag> <kernel> tools/bench_fragani.py 300 -> gpurun_out/<tag>_pmc/summary.txt, copied to
profiles/).  bench.py copies these figures into `also.fragment_ani.roofline*`, labelled as coming from these passes.

    python tools/pmc_fragani_to_json.py profiles/r03_pmc_map_segments_summary.txt profiles/r03_pmc_bucket_hits_summary.txt \
        <seed hits per bucket_hits dispatch>
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
    m, b = parse(map_file), parse(bucket_file)
    cycles = m["GRBM_GUI_ACTIVE"] / XCDS
    out = {
        "map_segments_kernel": {
            "source": f"rocprofv3 --pmc passes of tools/bench_fragani.py 300 ({map_file.name}); not measured inside this run",
            "valu_busy": m["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / cycles,
            "salu_busy": m["SQ_INSTS_SALU"] / CUS / cycles,
            "valu_instructions": m["SQ_INSTS_VALU"], "salu_instructions": m["SQ_INSTS_SALU"],
            "wait_share": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
            "waves_per_simd": m["SQ_WAVE_CYCLES"] / SIMDS / cycles,
            "avg_ms_per_dispatch": m["duration_ms"],
        },
    }
    cyc_b = b["GRBM_GUI_ACTIVE"] / XCDS
    fetch = b["FETCH_SIZE"] * 1024 * 2  # KiB per dispatch; x2 per the gfx950 correction (MI355X_MICROARCH.md)
    write = b["WRITE_SIZE"] * 1024
    entry = {
        "source": f"rocprofv3 --pmc passes of tools/bench_fragani.py 300 ({bucket_file.name}), FETCH_SIZE x2 per the gfx950 correction; not measured inside this run",
        "fetch_bytes_per_dispatch": fetch, "write_bytes_per_dispatch": write, "avg_ms_per_dispatch": b["duration_ms"],
        "counter_gbs": (fetch + write) / (b["duration_ms"] * 1e-3) / 1e9,
        "valu_busy": b["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / cyc_b,
        "wait_share": b["SQ_WAIT_ANY"] / b["SQ_WAVE_CYCLES"],
    }
    if hits_per_dispatch:
        entry["seed_hits_per_dispatch"] = hits_per_dispatch
        entry["algorithmic_bytes_per_hit"] = 24.0  # two reads of the 8-byte posting (count pass, scatter pass) + one 8-byte hit written
        entry["counter_bytes_per_hit"] = (fetch + write) / hits_per_dispatch
        entry["algorithmic_gbs"] = 24.0 * hits_per_dispatch / (b["duration_ms"] * 1e-3) / 1e9
    out["bucket_hits_kernel"] = entry
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
