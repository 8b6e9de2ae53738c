"""profiles/hash_counters.json from the counter passes of tools/pmc_passes.sh (k-mer hash kernel).

    python tools/pmc_to_json.py gpurun_out/<tag>_pmc profiles/hash_counters.json <label>
Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE (KiB) is doubled on gfx950 for wide
coalesced streaming reads, WRITE_SIZE (KiB) is exact; SQ_* cycle counters are in quad-cycles; GRBM_GUI_ACTIVE is the
sum over the 8 XCDs.  bench.py copies these figures into its JSON line and says where they come from.
"""
import json
import re
import sys
from pathlib import Path

src, dst, label = Path(sys.argv[1]), Path(sys.argv[2]), sys.argv[3]
text = (src / "summary.txt").read_text()
vals, dur = {}, {}
section = None
for line in text.splitlines():
    if line.startswith("== "):
        section = line[3:].strip()
    m = re.match(r"\s+(\w+)\s+mean per dispatch ([0-9.e+]+)", line)
    if m:
        vals.setdefault(m.group(1), []).append(float(m.group(2)))
    m = re.match(r"\s+duration_ms .*: mean ([0-9.]+)", line)
    if m and section:
        dur[section] = float(m.group(1))
mean = lambda k: sum(vals[k]) / len(vals[k])
simds = 1024
cycles = mean("GRBM_GUI_ACTIVE") / 8.0
clock = cycles / (dur["sq_valu"] * 1e-3) / 1e9
out = {
    "label": label,
    "kernel": "kmer_hash_kernel<31,true>, 1000 x 5 Mb (tools/pmc_hash.py), profiled dispatch",
    "traffic_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes ({label}); FETCH_SIZE x2 per the gfx950 correction; not measured inside this run",
    "hbm_bytes_per_launch": mean("FETCH_SIZE") * 1024 * 2 + mean("WRITE_SIZE") * 1024,
    "fetch_size_kib_raw": mean("FETCH_SIZE"),
    "write_size_kib": mean("WRITE_SIZE"),
    "effective_clock_ghz": clock,
    "clock_source": "GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of the profiled dispatch (MI355X_MICROARCH.md, DVFS give-back)",
    "valu_busy_pct": mean("SQ_ACTIVE_INST_VALU") * 4 / simds / cycles * 100.0,
    "valu_busy_note": "standard derived metric VALUBusy = SQ_ACTIVE_INST_VALU x 4 / SIMDs / (GRBM_GUI_ACTIVE / 8); it counts 4 cycles per "
    "instruction, the plain VOP2 ops issue in ~2.5, hence > 100",
    "sq": {k: mean(k) for k in sorted(vals) if k.startswith("SQ_")},
    "grbm_gui_active": mean("GRBM_GUI_ACTIVE"),
    "kernel_ms_by_pass": dur,
    "waves_per_simd_avg": mean("SQ_WAVE_CYCLES") * 4 / simds / cycles,
    "valu_instr_per_wave_window": mean("SQ_INSTS_VALU") / (1000 * 5_000_064 / 64.0),
    "valu_instr_per_wave_window_note": "SQ_INSTS_VALU / (arena positions / 64); tools/pmc_hash.py hashes 1000 genomes of 5 000 064 arena positions",
}
dst.write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
