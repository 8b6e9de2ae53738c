#!/bin/bash
# rocprofv3 counter passes, one --pmc set per run (never combined with runtime/sys traces).
# Usage on the GPU box (from the repo root):
#   bash tools/pmc_passes.sh <tag> <kernel-name filter> <python script> [script args...]
# e.g. bash tools/pmc_passes.sh r02_hash kmer_hash tools/pmc_hash.py 1000
set -u
TAG=${1:-r02}
FILTER=${2:-kmer_hash}
shift 2
if [ $# -eq 0 ]; then set -- tools/pmc_hash.py 1000; fi
ROOT=$(pwd)
SCRIPT=$ROOT/$1
shift
OUT=$ROOT/gpurun_out/${TAG}_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ -f "$OUT/../counters_available.txt" ] || rocprofv3 -L > "$OUT/../counters_available.txt" 2>&1
run_pass() {  # name, counters...
  local name=$1; shift
  # counters for the kernel under study only: with every dispatch of the run counted (the generator of the synthetic genomes
  # alone launches thousands) the profiler segfaults once counter instances x dispatches pass some size -- at 1 000
  # genomes every TCC counter and any two SQ counters did, one SQ counter or GRBM_GUI_ACTIVE did not (round 4)
  rocprofv3 --pmc "$@" --kernel-include-regex "$FILTER" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "$SCRIPT" "${ARGS[@]}" > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
ARGS=("$@")
run_pass sq_valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run_pass sq_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
cd "$ROOT" && python3 tools/pmc_summary.py "$OUT" "$FILTER" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# the raw per-dispatch CSVs are large (gpurun brings back 64 MiB at most): keep the summary and the logs
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
