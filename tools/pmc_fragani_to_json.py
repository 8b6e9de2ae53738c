#!/usr/bin/env python3
"""profiles/fragani_counters.json from the committed measurements of the fragment-ANI kernels -- every figure of
`also.fragment_ani.roofline*` in bench.py's line follows from files under profiles/ and from nothing else:

    <tag>_pmc_map_segments_summary.txt, _pmc_map_sparse_summary.txt, _pmc_bucket_hits_summary.txt, _pmc_minimizer_summary.txt
        tools/pmc_passes.sh <tag> <kernel> tools/bench_fragani.py 1000 0 interleaved 78 (rocprofv3 --pmc, one set per run)
    <tag>_map_cut_valu.txt
        tools/map_cut_valu.sh: SQ_INSTS_VALU of map_segments_kernel cut short after each phase (enum MapCut) -- the
        instruction counter PER PHASE of the tools build
    <tag>_fragani_n1000_one_batch_trace.txt
        tools/map_stats.py 1000 78: event counts of the same batch (stats build)
    <tag>_fetch_calibration.txt
        tools/fetch_calib.sh: what FETCH_SIZE counts for posting-list-shaped reads of 2- and 8-byte items

    python tools/pmc_fragani_to_json.py r06 > profiles/fragani_counters.json

The work model of map_segments_kernel (profiles/README.md, "Work-based roofline of the mapping kernel"): the vector
instructions a dispatch NEEDS if a perfect bound spared it everything but the states that tie their candidate's optimum,
every unit priced at what the kernel itself issues for it -- MEASURED per phase (the cut runs), not read off a listing:

    fixed       everything up to and including the candidate's set-up: record, sketch, hits ordered (by counting for the
                segments whose hits are one cluster and up to two strays, by the network for the rest), the L1 scan where a
                segment needs one (a run of hits with or without strays skips it), candidate set-up           = V[cut 3]
    one round   per candidate ONE complete round on the group of the expected optimum -- the group's seed-hit bound, the
                stretch loaded, window ends, the tight bound asked once, ranks, tables, coarse search, fine passes, fold --
                = V[24] - V[3] (cut 24: the kernel ends after its first round), of which a perfect bound needs the share that
                the minimizers of ONE window (2 count_windows / (w + 1) = 237) plus one per further tying state are of the
                entries such a round ranks: share = (C x 237 + W) / (C x entries ranked per full round)

needed = V[3] + (V[24] - V[3]) x share; frac = needed / V[9].  `frac_first_group` = V[23] / V[9]: what the kernel issues when it only ever looks
at the group of the expected optimum (a perfect bound on whole groups), measured directly.  `frac_minimal_sort`: the same with EVERY segment's hits ordered at the price of the counting sort
(the network's price per segment is the start-of-round measurement, all segments through it).
"""
import json
import re
import sys
from pathlib import Path

SIMDS, CUS, XCDS = 1024, 256, 8
WINDOW_ENTRIES = 237.0  # 2 count_windows / (w + 1) for k = 16, fragLen = 3000 (count_windows 2962, w 24)
ROOT = Path(__file__).resolve().parent.parent / "profiles"


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


def parse_cuts(path: Path) -> dict:
    """cut number -> {counter: mean per dispatch} from tools/map_cut_valu.sh"""
    cuts = {}
    for line in path.read_text().splitlines():
        m = re.match(r"cut\s+(\d+):\s+(.*?)\s+\(", line)
        if m:
            vals = re.findall(r"(SQ_\w+) (\S+)", m.group(2))
            cuts[int(m.group(1))] = {k: float(v) for k, v in vals}
    return cuts


def parse_events(path: Path) -> dict:
    text = path.read_text()
    line = [x for x in text.splitlines() if "map stats:" in x][-1]

    def grab(pattern):
        m = re.search(pattern, line)
        if not m:
            raise ValueError(f"{path.name}: no {pattern!r}")
        return [float(x) for x in m.groups()]

    seg, hits, cand = grab(r"map stats: (\d+) segments at L1 with (\d+) hits, (\d+) candidates")
    rounds, entries, windows, fine = grab(r"(\d+) rounds, (\d+) stretch entries, (\d+) windows evaluated, (\d+) fine passes")
    (exact,) = grab(r"(\d+) windows with an exact value")
    (ended,) = grab(r"(\d+) rounds ended by it")
    range_entries, ties = grab(r"work model: (\d+) minimizers in the candidates' ranges, (\d+) states tying")
    one_run, one_run_hits = grab(r"(\d+) segments that are one run of hits \(no L1 scan\) with (\d+) hits")
    counted, but_one, but_two = grab(r"ordered by counting\): (\d+) segments, all but one hit: (\d+), all but two: (\d+)")
    ev = {"segments": seg, "seed_hits": hits, "candidates": cand, "rounds": rounds, "rounds_ended_by_the_tight_bound": ended,
          "full_rounds": rounds - ended, "stretch_entries_ranked": entries, "windows_evaluated": windows, "fine_passes": fine,
          "windows_with_an_exact_value": exact, "minimizers_in_candidate_ranges": range_entries, "tying_states": ties,
          "one_run_segments_no_l1_scan": one_run, "their_seed_hits": one_run_hits,
          "segments_whose_hits_are_one_cluster": counted, "one_cluster_but_one_hit": but_one, "one_cluster_but_two_hits": but_two}
    sparse = [x for x in text.splitlines() if "sparse stats:" in x]
    if sparse:
        m = re.search(r"sparse stats: (\d+) segments with a candidate, (\d+) candidates, (\d+) groups of begins evaluated, (\d+) begins, (\d+) states, (\d+) begins tying", sparse[-1])
        if m:
            ev["sparse"] = dict(zip(("segments_with_a_candidate", "candidates", "groups", "begins", "states", "begins_tying_when_folded"), (float(x) for x in m.groups())))
        m = re.search(r"(\d+) groups that hold a candidate's first or last tying begin", sparse[-1])
        if m and "sparse" in ev:
            ev["sparse"]["groups_with_the_first_or_last_tie"] = float(m.group(1))
    m = re.search(r"(\d+) segments of at most 8 hits in the sparse kernel, (\d+) of them handed on", text)
    if m:
        ev.setdefault("sparse", {})["segments"] = float(m.group(1))
        ev["sparse"]["handed_on"] = float(m.group(2))
    m = re.search(r"(\d+) fragments, (\d+) seed hits", text)
    if m:
        ev["fragments"], ev["seed_hits_of_the_batch"] = float(m.group(1)), float(m.group(2))
    return ev


def work_model(cuts: dict, ev: dict, network_valu_per_segment: float | None) -> dict:
    v = {c: d["SQ_INSTS_VALU"] for c, d in cuts.items()}
    need = {3, 9, 10, 11, 23, 24}
    if not need.issubset(v):
        return {"error": f"cut file lacks cuts {sorted(need - set(v))}"}
    c, w = ev["candidates"], ev["tying_states"]
    entries_per_round = ev["stretch_entries_ranked"] / ev["full_rounds"]
    share = min(1.0, (c * WINDOW_ENTRIES + w) / (c * entries_per_round))
    parts = {"fixed_up_to_the_candidates_set_up": v[3], "one_round_per_candidate_share_of_one_window_and_the_ties": (v[24] - v[3]) * share}
    needed = sum(parts.values())
    per_seg = lambda x: x / ev["segments"]  # noqa: E731
    out = {
        "work_model": "vector instructions a dispatch needs with a perfect bound, priced at what the kernel issues as MEASURED per phase (SQ_INSTS_VALU of the kernel "
        "cut short after each phase, tools/map_cut_valu.sh): everything up to the candidate's set-up as issued (the L1 scan only where a segment takes it) + per "
        "candidate ONE complete round on the group of the expected optimum (cut 24), scaled by the share of its ranked entries that one window (237) and the further "
        "tying states are.  profiles/README.md shows the arithmetic",
        "algorithmic_units_per_dispatch": {k: ev[k] for k in ("segments", "seed_hits", "candidates", "tying_states", "full_rounds", "stretch_entries_ranked",
                                                               "windows_evaluated", "windows_with_an_exact_value", "one_run_segments_no_l1_scan",
                                                               "segments_whose_hits_are_one_cluster", "one_cluster_but_one_hit", "one_cluster_but_two_hits")},
        "valu_instructions_per_phase_per_segment": {
            "record_and_sketch": per_seg(v[10]), "hits_ordered_and_staged": per_seg(v[11] - v[10]),
            "sketch_table": per_seg(v[1] - v[11]) if 1 in v else None, "l1": per_seg(v[2] - v[1]) if 1 in v and 2 in v else None,
            "candidate_set_up": per_seg(v[3] - v[2]) if 2 in v else None,
            "first_round_bound_stretch_ranks_bitmap": per_seg(v[5] - v[3]) if 5 in v else None,
            "first_round_items_and_coarse_table": per_seg(v[7] - v[5]) if 5 in v and 7 in v else None,
            "first_round_window_masks_and_coarse_search": per_seg(v[8] - v[7]) if 7 in v and 8 in v else None,
            "first_round_fine_passes_and_fold": per_seg(v[24] - v[8]) if 8 in v else None,
            "rest_of_the_first_group": per_seg(v[23] - v[24]), "other_groups": per_seg(v[9] - v[23]), "whole_kernel": per_seg(v[9])},
        "entries_ranked_per_full_round": entries_per_round, "entries_of_one_window": WINDOW_ENTRIES, "share_of_a_round_needed": share,
        "algorithmic_valu_instructions_per_dispatch": needed, "algorithmic_valu_instructions_by_unit": parts,
        "counted_valu_instructions_per_dispatch": v[9], "frac": needed / v[9], "frac_first_group": v[23] / v[9],
        "frac_first_group_what": "vector instructions of the kernel when it only ever looks at the group of 64 begins around the expected optimum (cut 23) / instructions issued",
        "source": "profiles/<tag>_map_cut_valu.txt (tools build) and profiles/<tag>_fragani_n1000_one_batch_trace.txt (stats build): one batch of 2^17 query fragments against the 1 000-genome index",
    }
    if network_valu_per_segment:
        # the counting sort's price from the mix: (V11 - V10) = counted x a + networked x b, b = the network's measured price
        counted = ev["segments_whose_hits_are_one_cluster"] + ev["one_cluster_but_one_hit"] + ev["one_cluster_but_two_hits"]
        networked = max(ev["segments"] - counted, 0.0)
        a = max(((v[11] - v[10]) - networked * network_valu_per_segment) / counted, 0.0) if counted else None
        if a is not None:
            minimal_fixed = v[3] - (v[11] - v[10]) + ev["segments"] * a
            out["counting_sort_valu_per_segment"] = a
            out["network_sort_valu_per_segment"] = network_valu_per_segment
            out["frac_minimal_sort"] = (needed - v[3] + minimal_fixed) / (v[9] - (v[11] - v[10]) + ev["segments"] * a)
            out["frac_minimal_sort_note"] = ("needed and issued with EVERY segment's hits ordered at the counting sort's price (the network's price per segment: the "
                                             "all-network build measured at the start of the round, profiles/r06_map_cut_valu_start_of_round.txt)")
    return out


def busy(m: dict) -> dict:
    cycles = m["GRBM_GUI_ACTIVE"] / XCDS
    return {"valu_busy": m["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / cycles, "salu_busy": m["SQ_INSTS_SALU"] / CUS / cycles if "SQ_INSTS_SALU" in m else None,
            "valu_instructions": m["SQ_INSTS_VALU"], "salu_instructions": m.get("SQ_INSTS_SALU"), "lds_instructions": m.get("SQ_INSTS_LDS"),
            "wait_share": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in m else None,
            "waves_per_simd": m["SQ_WAVE_CYCLES"] * 4 / SIMDS / cycles,  # SQ_* cycle counters are in quad-cycles
            "avg_ms_per_dispatch": m["duration_ms"],
            "fetch_bytes_per_dispatch_as_counted": m.get("FETCH_SIZE", 0.0) * 1024, "write_bytes_per_dispatch": m.get("WRITE_SIZE", 0.0) * 1024}


def parse_calibration(path: Path) -> dict:
    out = {}
    for line in path.read_text().splitlines():
        m = re.match(r"(calib_\S+(?: \S+)?): bytes asked (\S+), .* = (\S+), FETCH_SIZE counted (\S+) bytes -> counted / asked = (\S+), counted / line bytes = (\S+)", line)
        if m:
            out[m.group(1).rstrip(":")] = {"bytes_asked": float(m.group(2).rstrip(",")), "line_bytes": float(m.group(3).rstrip(",")),
                                           "counted": float(m.group(4)), "counted_over_asked": float(m.group(5).rstrip(",")), "counted_over_line_bytes": float(m.group(6))}
    return out


def build(tag: str, root: Path = ROOT) -> dict:
    what = "tools/bench_fragani.py 1000 0 interleaved 78 (the benchmark's 1 000 genomes, one batch of 2^17 query fragments)"
    ev = parse_events(root / f"{tag}_fragani_n1000_one_batch_trace.txt")
    cuts = parse_cuts(root / f"{tag}_map_cut_valu.txt")
    start = root / "r06_map_cut_valu_start_of_round.txt"
    network = None
    if start.is_file():
        sc = parse_cuts(start)
        network = (sc[11]["SQ_INSTS_VALU"] - sc[10]["SQ_INSTS_VALU"]) / sc[10]["SQ_WAVES"]
    m = parse(root / f"{tag}_pmc_map_segments_summary.txt")
    out = {"map_segments_kernel": {"source": f"rocprofv3 --pmc passes of {what} ({tag}_pmc_map_segments_summary.txt); not measured inside this run", **busy(m)}}
    out["map_segments_kernel"]["valu_instructions_per_segment"] = m["SQ_INSTS_VALU"] / ev["segments"]
    out["map_segments_kernel"]["work"] = work_model(cuts, ev, network)
    sp_file = root / f"{tag}_pmc_map_sparse_summary.txt"
    if sp_file.is_file():
        sp = parse(sp_file)
        entry = {"source": f"rocprofv3 --pmc passes of {what} ({sp_file.name}); event counts of the stats build; not measured inside this run", **busy(sp)}
        se = ev.get("sparse", {})
        if se.get("segments"):
            entry["segments_per_dispatch"] = se["segments"]
            entry["valu_instructions_per_segment"] = sp["SQ_INSTS_VALU"] / se["segments"]
        if se.get("states"):
            # work-based, in groups of 64 begins (the unit the kernel loads a stretch and builds its bit masks for): of the states that
            # tie a candidate's best only the first and the last decide the mapping's position, so what any order of evaluation has to
            # load is the one or two groups that hold them; the kernel also evaluates the groups it cannot rule out beforehand
            entry["events_per_dispatch"] = se
            entry["valu_instructions_per_state_evaluated"] = sp["SQ_INSTS_VALU"] / se["states"]
            entry["states_per_begin"] = se["states"] / se["begins"]
            entry["groups_per_segment"] = se["groups"] / se["segments_with_a_candidate"]
            if se.get("groups_with_the_first_or_last_tie"):
                entry["frac"] = se["groups_with_the_first_or_last_tie"] / se["groups"]
                entry["frac_what"] = ("work-based, in groups of 64 begins: groups that hold a candidate's first or last tying begin (the two states that decide the mapping's "
                                      "position: what any order of evaluation has to load) / groups evaluated; the kernel's instructions are proportional to the groups it "
                                      "evaluates (valu_busy is the pipe's share)")
        out["map_sparse_kernel"] = entry
    mi_file = root / f"{tag}_pmc_minimizer_summary.txt"
    if mi_file.is_file():
        mi = parse(mi_file)
        entry = {"source": f"rocprofv3 --pmc passes of {what} ({mi_file.name}); not measured inside this run", **busy(mi)}
        # 1 000 x 5 Mb: tiles of 1 920 own positions (2 048 hashed: 128 of look-back), two MurmurHash3_x64_128 of 16 bytes per position
        positions = 1000 * 5_000_064
        entry["arena_positions"] = positions
        entry["valu_instructions_per_position"] = mi["SQ_INSTS_VALU"] * 64 / positions
        # needed: the two hashes of every position, at the kernel's own 126 vector instructions per position and lane for them
        # (static listing, lines of the hashing loop: 1 008 per eight positions), once per position (no look-back, nothing else)
        entry["hash_valu_instructions_per_position"] = 126.0
        entry["frac"] = 126.0 / entry["valu_instructions_per_position"]
        entry["frac_what"] = ("work-based: vector instructions of the two MurmurHash3 per position (126 per position: the kernel's own hashing loop) / vector instructions "
                              "issued per position (look-back positions hashed twice, winnowing, contig bookkeeping, the chained scan); valu_busy is the pipe's share")
        entry["algorithmic_gbs"] = (positions / 4 + 12 * 4.0e8) / (mi["duration_ms"] * 1e-3) / 1e9
        entry["algorithmic_bytes_note"] = "2 bits per position read + three 4-byte arrays written per minimizer (4*10^8 of them): nowhere near an HBM roof"
        out["minimizer_kernel"] = entry
    b = parse(root / f"{tag}_pmc_bucket_hits_summary.txt")
    hits = ev.get("seed_hits_of_the_batch")
    cal_file = root / f"{tag}_fetch_calibration.txt"
    cal = parse_calibration(cal_file) if cal_file.is_file() else {}
    fetch_counted = b["FETCH_SIZE"] * 1024
    write = b["WRITE_SIZE"] * 1024
    entry = {"source": f"rocprofv3 --pmc passes of {what} ({tag}_pmc_bucket_hits_summary.txt); FETCH_SIZE corrected by the factors measured for this access pattern "
                       f"({cal_file.name}: tools/fetch_calib); not measured inside this run", **busy(b)}
    if hits:
        entry["seed_hits_per_dispatch"] = hits
        entry["algorithmic_bytes_per_hit"] = 18.0  # the posting's 2-byte genome (counting pass) + the 8-byte posting (scatter pass) + one 8-byte hit written
        entry["algorithmic_gbs"] = 18.0 * hits / (b["duration_ms"] * 1e-3) / 1e9
        entry["fetch_bytes_per_hit_as_counted"] = fetch_counted / hits
        k16, k64 = cal.get("calib_runs<unsigned short>"), cal.get("calib_runs<unsigned long>")
        if k16 and k64:
            # The kernel asks for 2 + 8 bytes per hit in runs like the calibration's; the counter reports f16 x 2 + f64 x 8 counted bytes
            # per hit for them: the factor that turns counted bytes into 64-byte lines actually moved is the calibration's own
            # (counted -> line bytes), weighted by what each pass contributes to the count.
            c16, c64 = 2.0 * k16["counted_over_asked"], 8.0 * k64["counted_over_asked"]  # counted bytes per hit the two passes are expected to give
            lines16, lines64 = c16 / k16["counted_over_line_bytes"], c64 / k64["counted_over_line_bytes"]  # 64-byte lines x 64 per hit behind them
            factor = (lines16 + lines64) / (c16 + c64)
            entry["fetch_calibration"] = {"counted_over_asked_2_byte_runs": k16["counted_over_asked"], "counted_over_asked_8_byte_runs": k64["counted_over_asked"],
                                          "counted_over_line_bytes_2_byte_runs": k16["counted_over_line_bytes"], "counted_over_line_bytes_8_byte_runs": k64["counted_over_line_bytes"],
                                          "expected_counted_bytes_per_hit": c16 + c64, "line_bytes_per_counted_byte": factor}
            entry["fetch_bytes_per_hit_calibrated"] = fetch_counted / hits * factor
            entry["counter_bytes_per_hit"] = (fetch_counted * factor + write) / hits
            entry["counter_gbs"] = (fetch_counted * factor + write) / (b["duration_ms"] * 1e-3) / 1e9
            entry["traffic_over_algorithmic"] = entry["counter_bytes_per_hit"] / 18.0
        else:
            entry["counter_bytes_per_hit"] = (fetch_counted + write) / hits
            entry["counter_gbs"] = (fetch_counted + write) / (b["duration_ms"] * 1e-3) / 1e9
        entry["write_bytes_per_hit"] = write / hits
    out["bucket_hits_kernel"] = entry
    out["tag"] = tag
    return out


def main() -> None:
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    json.dump(build(tag), sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
