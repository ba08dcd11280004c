#!/bin/bash
# collect_profiles.sh TAG - copy the summaries of a tools/profile_round.sh run from the scratch directory
# (gpurun_out/prof_<TAG>/, which is what comes back from the GPU box) into profiles/ (tracked), and rebuild
# the PMC summaries from the raw counter files.  Runs anywhere (no GPU needed).
set -eo pipefail
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
cd "$ROOT"
cp "$(find "$OUT/c2" -name 'c2_kernel_stats.csv' | head -1)" "profiles/${TAG}_C2_fused_kernel_stats.csv"
python3 tools/traffic_from_pmc.py "$(find "$OUT/pmc_fetch" -name 'f_counter_collection.csv' | head -1)" \
    "$(find "$OUT/pmc_write" -name 'w_counter_collection.csv' | head -1)" 6000000 1000 f32 "${TAG}_C2_fused"
cp "$(find "$OUT/c2sq" -name 'c2sq_kernel_stats.csv' | head -1)" "profiles/${TAG}_C2sq_kernel_stats.csv"
python3 tools/lds_from_pmc.py "$(find "$OUT/pmc_lds" -name 'l_counter_collection.csv' | head -1)" "${TAG}_C2sq"
f=$(find "$OUT/pmc_lds_c4" -name 'l_counter_collection.csv' 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/lds_from_pmc.py "$f" "${TAG}_C4shard"
for cfg in C3 C4shard C5shard; do
    f=$(find "$OUT/$cfg" -name 'x_kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" "profiles/${TAG}_${cfg}_kernel_stats.csv"
done
for cfg in c2 c2sq C3 C4shard C5shard; do
    # the bench line each profiled run printed (under the profiler: a few per cent slower than a plain run)
    grep -h '^{"metric"' "$OUT/$cfg.log" > "profiles/${TAG}_bench_under_rocprof_${cfg}.json" || true
done
python3 tools/trace_timed.py "$(find "$OUT/c2" -name c2_kernel_trace.csv | head -1)" 20 "" "profiles/${TAG}_bench_under_rocprof_c2.json" > "profiles/${TAG}_C2_fused_timed_launches.txt"
ls profiles | grep "^${TAG}_"
