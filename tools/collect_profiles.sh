# developer tool (CPU box): copy the summaries of a tools/experiments/profile_all.sh run from gpurun_out/ into profiles/
# usage: TAG=r03b bash tools/collect_profiles.sh
set -e
TAG=${TAG:-r03b}
for sfx in "" _bf16 _train; do
  O=gpurun_out/$TAG$sfx
  cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${TAG}${sfx}_kernel_stats.csv
  cp $O/pmc.json profiles/${TAG}${sfx}_pmc.json
  cp $O/bench_under_rocprof.json profiles/${TAG}${sfx}_bench_under_rocprof.json
  cp $O/pmc_traffic$sfx.json profiles/pmc_traffic$sfx.json
done
ls -la profiles | grep $TAG
