#!/bin/bash
# What tools/profile_round.sh <tag> left under gpurun_out/ copied to the names profiles/ keeps:  bash tools/copy_round_profiles.sh <tag>
# (then: python tools/pmc_to_json.py gpurun_out/<tag>_hash_pmc profiles/hash_counters.json "<label>";
#        python tools/pmc_fragani_to_json.py <tag> > profiles/fragani_counters.json; python tools/gen_profiles_readme_<tag>.py)
set -e
T=${1:-r06}
G=gpurun_out
P=profiles
cp $G/${T}_bench.json $P/${T}_bench_n1000_result.json
cp $G/${T}_stats_bench.kernel_stats.csv $P/${T}_bench_n1000_kernel_stats.csv
cp $G/${T}_stats_fragani.kernel_stats.csv $P/${T}_fragani_n1000_kernel_stats.csv
cp $G/${T}_fragani1000_onebatch_trace.txt $P/${T}_fragani_n1000_one_batch_trace.txt
cp $G/${T}_fragani1000_rearranged_onebatch_trace.txt $P/${T}_fragani_n1000_rearranged_one_batch_trace.txt
cp $G/${T}_fragani1000_rearranged.log $P/${T}_fragani_n1000_rearranged_run.txt
cp $G/${T}_map_cut_valu.txt $P/${T}_map_cut_valu.txt
cp $G/${T}_map_cut.txt $P/${T}_map_segments_phase_cuts.txt
cp $G/${T}_fetch_calibration.txt $P/${T}_fetch_calibration.txt
for pair in hash:kmer_hash fragmap:map_segments sparse:map_sparse bucket:bucket_hits minimizer:minimizer postings:postings rs_scatter:rs_scatter; do
  cp $G/${T}_${pair%%:*}_pmc/summary.txt $P/${T}_pmc_${pair##*:}_summary.txt
done
git status --short $P | head -30
