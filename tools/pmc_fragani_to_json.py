#!/usr/bin/env python3
"""profiles/fragani_counters.json from the summaries of the rocprofv3 counter passes on the fragment-ANI kernels
(tools/pmc_passes.sh <tag> <kernel> tools/bench_fragani.py 1000 0 interleaved 78 -> gpurun_out/<tag>_pmc/summary.txt,
copied to profiles/: the benchmark's 1 000 genomes, one batch of 2^17 query fragments per repetition).  bench.py copies
these figures into `also.fragment_ani.roofline*`, labelled as coming from these passes.

    python tools/pmc_fragani_to_json.py profiles/r05_pmc_map_segments_summary.txt profiles/r05_pmc_bucket_hits_summary.txt \
        <seed hits per bucket_hits dispatch> [workload label] [profiles/r05_fragani_n1000_one_batch_trace.txt: event counts for the work model]
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


# ---- the work model of map_segments_kernel (profiles/README.md, "Work-based roofline of the mapping kernel")
# Units, from the event counters of the stats build (tools/map_stats.py, one dispatch = one batch of 2^17 fragments):
#   S segments that reach L1, H their seed hits, C candidates, W states that tie their candidate's optimum (no bound can
#   spare them; one probe where nothing was evaluated).
# Algorithmic work of a dispatch: every segment set up once (sketch loaded and bucketed, candidate set up), every hit
# ordered and scanned once, per candidate the minimizers of ONE window (2 count_windows / (w + 1) = 237 expected for
# k = 16, fragLen = 3000) plus one per further tying state ranked and entered in the bit tables once, one exact window
# evaluation per tying state.  Costs in vector instructions of a wave per unit: the kernel's own, from the static listing
# (tools/isa_lines.py on the build the counters come from) with the loop trip counts of the benchmark.
WORK_COSTS = {
    "per_segment": 320.0,   # record + sketch load 75, candidate set-up 160, result 20, L1 tail 65 (the sketch's bucket table, 80, is made once per fragment by query_sketch_kernel since the end of round 5)
    "per_hit": 11.6,        # ordered in registers (36 stages x 4 keys x ~5.5 = 800 per <= 256 hits), L1 scan 348 per 64 hits, staged: / 139 hits
    "per_entry": 2.4,       # stretch load + window ends 200, ranks 190, match bitmap 50, coarse table 320: 760 per round of 320 entries
    "per_window": 13.0,     # window mask + coarse search 410, fine pass 285 x 1.2, fold 80: 830 per pass of 64 windows
}
WINDOW_ENTRIES = 237.0


def work_model(trace_file: Path, valu_instructions: float) -> dict:
    text = trace_file.read_text()
    m = re.search(r"map stats: (\d+) segments at L1 with (\d+) hits, (\d+) candidates", text)
    w = re.search(r"work model: (\d+) minimizers in the candidates' ranges, (\d+) states tying", text)
    if not m or not w:
        return {"error": f"no event counts in {trace_file.name}"}
    seg, hits, cand = (float(x) for x in m.groups())
    range_entries, ties = (float(x) for x in w.groups())
    entries = cand * WINDOW_ENTRIES + ties
    parts = {
        "segments": seg * WORK_COSTS["per_segment"], "hits": hits * WORK_COSTS["per_hit"],
        "entries": entries * WORK_COSTS["per_entry"], "windows": ties * WORK_COSTS["per_window"],
    }
    total = sum(parts.values())
    return {
        "work_model": "vector instructions a dispatch needs with a perfect bound: each segment set up once, each seed hit ordered and scanned "
        "once, per candidate the minimizers of one window (237) plus one per further state tying the optimum ranked once, one exact window "
        "evaluation per tying state; per-unit costs are the kernel's own (static listing x loop trips), profiles/README.md",
        "algorithmic_units_per_dispatch": {"segments": seg, "seed_hits": hits, "candidates": cand, "tying_states": ties,
                                           "entries_of_one_window_per_candidate_plus_ties": entries, "minimizers_in_candidate_ranges": range_entries},
        "valu_instructions_per_unit": WORK_COSTS,
        "algorithmic_valu_instructions_per_dispatch": total, "algorithmic_valu_instructions_by_unit": parts,
        "counted_valu_instructions_per_dispatch": valu_instructions,
        "frac": total / valu_instructions if valu_instructions else None,
        "events_source": trace_file.name,
    }


def main() -> None:
    map_file, bucket_file = Path(sys.argv[1]), Path(sys.argv[2])
    hits_per_dispatch = float(sys.argv[3]) if len(sys.argv) > 3 else None
    what = sys.argv[4] if len(sys.argv) > 4 else "tools/bench_fragani.py 1000 0 interleaved 78 (the benchmark's 1 000 genomes, one batch of 2^17 query fragments)"
    trace_file = Path(sys.argv[5]) if len(sys.argv) > 5 else None
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
    if trace_file is not None:
        out["map_segments_kernel"]["work"] = work_model(trace_file, m["SQ_INSTS_VALU"])
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
