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
bash tools/pmc_passes.sh ${TAG}_sparse "map_sparse_kernel" tools/bench_fragani.py 1000 0 interleaved 78
bash tools/pmc_passes.sh ${TAG}_bucket bucket_hits tools/bench_fragani.py 1000 0 interleaved 78
bash tools/pmc_passes.sh ${TAG}_minimizer "minimizer_kernel" tools/bench_fragani.py 1000 0 interleaved 78
bash tools/pmc_passes.sh ${TAG}_postings "postings_kernel" tools/bench_fragani.py 1000 0 interleaved 78
bash tools/pmc_passes.sh ${TAG}_rs_scatter "rs_scatter_kernel" tools/bench_fragani.py 1000 0 interleaved 78
# vector instructions of the general mapping kernel per phase (the kernel cut short after each phase under --pmc)
bash tools/map_cut_valu.sh ${TAG} > /dev/null
# FETCH_SIZE calibrated for the seeding kernel's access pattern
bash tools/fetch_calib.sh ${TAG} > /dev/null
# the set with indels, rearrangements, repeat families and contigs: event counts of one batch, and the whole run
PA_SYNTH=rearranged python3 tools/map_stats.py 1000 78 2>&1 | grep "pa_fragani:" > gpurun_out/${TAG}_fragani1000_rearranged_onebatch_trace.txt
PA_SYNTH=rearranged python3 tools/bench_fragani.py 1000 > gpurun_out/${TAG}_fragani1000_rearranged.log 2>&1
# seed hits per bucket_hits dispatch of that run (the denominator of its bytes per hit) and the mapping kernels' event counts
# (the switches live in the tools build: tools/map_stats.py loads libpyani_hip_stats.so, `make -C pyani_plus_amd/csrc stats`)
python3 tools/map_stats.py 1000 78 2>&1 | grep "pa_fragani:" > gpurun_out/${TAG}_fragani1000_onebatch_trace.txt
# where the mapping kernel's time goes: the kernel cut short after each phase (tools build)
python3 tools/map_cut.py 1000 9,9,10,11,1,2,3,4,9,23,9 > gpurun_out/${TAG}_map_cut.txt 2>&1
# the bench line itself, un-profiled
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
for d in gpurun_out/${TAG}_stats_bench gpurun_out/${TAG}_stats_fragani; do
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    # our own kernels only (the synthetic-genome generator is torch plumbing), then drop the raw traces
    { head -1 "$f"; grep -v "at::native\|rocclr\|hiprand" "$f" | tail -n +2; } > "$d.kernel_stats.csv"
    echo "== $d.kernel_stats.csv"; cut -c1-60,200- "$d.kernel_stats.csv" | head -14
  fi
  rm -rf "$d"
done
