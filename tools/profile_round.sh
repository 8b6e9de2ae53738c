#!/bin/bash
# The rocprofv3 evidence of a round, gathered in one gpurun call:  bash tools/profile_round.sh <tag>
# (kernel-trace statistics of the bench command and of the fragment-ANI run, counter passes of the two dominant
# kernels).  Results land under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -u
TAG=${1:-r02}
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_stats_bench" -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-also > "$ROOT/gpurun_out/${TAG}_stats_bench.json" 2> "$ROOT/gpurun_out/${TAG}_stats_bench.err"
echo "bench stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_stats_fragani" -- python3 "$ROOT/tools/bench_fragani.py" 1000 0 > "$ROOT/gpurun_out/${TAG}_stats_fragani.log" 2>&1
echo "fragani stats rc=$?"
cd "$ROOT"
bash tools/pmc_passes.sh ${TAG}_hash kmer_hash tools/pmc_hash.py 1000
# the fragment-ANI kernels at the benchmark's 1 000 genomes, one batch of 2^17 query fragments (78 query genomes) per repetition
bash tools/pmc_passes.sh ${TAG}_fragmap "map_segments_kernel<320u, true>" tools/bench_fragani.py 1000 0 interleaved 78
bash tools/pmc_passes.sh ${TAG}_bucket bucket_hits tools/bench_fragani.py 1000 0 interleaved 78
# seed hits per bucket_hits dispatch of that run (the denominator of its bytes per hit)
PA_FRAGANI_TRACE=1 python3 tools/bench_fragani.py 1000 0 interleaved 78 2>&1 | grep "seed hits" | head -1 > gpurun_out/${TAG}_fragani1000_onebatch_trace.txt
for d in gpurun_out/${TAG}_stats_bench gpurun_out/${TAG}_stats_fragani; do
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    # our own kernels only (the synthetic-genome generator is torch plumbing), then drop the raw traces
    { head -1 "$f"; grep -v "at::native\|rocclr\|hiprand" "$f" | tail -n +2; } > "$d.kernel_stats.csv"
    echo "== $d.kernel_stats.csv"; cut -c1-60,200- "$d.kernel_stats.csv" | head -14
  fi
  rm -rf "$d"
done
