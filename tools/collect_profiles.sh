#!/bin/bash
# After `gpurun -- bash tools/profile_all.sh <tag>`: copy the per-workload summaries and rocprofv3 kernel-stats
# tables from gpurun_out/ (scratch) into profiles/ (tracked) and rebuild the tables bench.py prices against.
#   tools/collect_profiles.sh <tag>
set -eu
TAG=$1
cd "$(dirname "$0")/.."
FILES=""
for d in gpurun_out/prof_${TAG}_*; do
  name=${d#gpurun_out/prof_${TAG}_}
  cp "$d/summary.json" "profiles/${TAG}_${name}.json"
  FILES="$FILES profiles/${TAG}_${name}.json"
  ks=$(find "$d/trace" -name "*kernel_stats.csv" | head -1)      # rocprofv3 --kernel-trace --stats summary of the bench command
  [ -n "$ks" ] && cp "$ks" "profiles/${TAG}_${name}_kernel_stats.csv"
done
python3 tools/update_profile_tables.py $FILES      # (only the summaries just copied: other records share the tag's prefix)
