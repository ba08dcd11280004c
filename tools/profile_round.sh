#!/bin/bash
# profile_round.sh TAG - the rocprofv3 runs behind profiles/ (run on the GPU box through gpurun):
#   1. kernel trace + stats of the default bench (C2, single-sweep path)
#   2. FETCH_SIZE / WRITE_SIZE PMC passes of the same command (separate passes, no trace domains
#      other than --kernel-trace) -> tools/traffic_from_pmc.py -> profiles/traffic_latest.json
#   3. kernel trace + stats of the sorted path (C2sq: sort + PAV z-step, two sweeps)
#   4. LDS PMC pass of the sorted path (bank conflicts / LDS activity of the sort and PAV kernels)
# Raw output goes to gpurun_out/prof_<TAG>/ (scratch); the summaries are copied to profiles/.
set -eo pipefail
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$ROOT/profiles"
export TMPDIR=/tmp
cd "$ROOT"
B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-c1"   # the driver's command (its extra records left out)

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c2" -o c2 -- python3 $B > "$OUT/c2.log" 2>&1
cp "$(find "$OUT/c2" -name 'c2_kernel_stats.csv' | head -1)" "profiles/${TAG}_C2_fused_kernel_stats.csv"
echo "[1/5] C2 kernel stats done"

rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o f -- python3 $B --steps 5 > "$OUT/f.log" 2>&1
echo "[2/5] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o w -- python3 $B --steps 5 > "$OUT/w.log" 2>&1
python3 tools/traffic_from_pmc.py "$(find "$OUT/pmc_fetch" -name 'f_counter_collection.csv' | head -1)" \
    "$(find "$OUT/pmc_write" -name 'w_counter_collection.csv' | head -1)" 6000000 1000 f32 "${TAG}_C2_fused"
echo "[3/5] WRITE_SIZE pass + traffic summary done"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c2sq" -o c2sq -- python3 $B --config C2sq > "$OUT/c2sq.log" 2>&1
cp "$(find "$OUT/c2sq" -name 'c2sq_kernel_stats.csv' | head -1)" "profiles/${TAG}_C2sq_kernel_stats.csv"
echo "[4/5] C2sq kernel stats done"

rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d "$OUT/pmc_lds" -o l -- python3 $B --config C2sq --steps 5 > "$OUT/l.log" 2>&1
python3 tools/lds_from_pmc.py "$(find "$OUT/pmc_lds" -name 'l_counter_collection.csv' | head -1)" "${TAG}_C2sq"
echo "[5/5] LDS PMC pass done"

# the same counters where the z-step runs sort + PAV in EVERY iteration (EHRM: 32-bit radix sort, both PAV kernels)
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d "$OUT/pmc_lds_c4" -o l -- python3 $B --config C4shard --steps 5 > "$OUT/l4.log" 2>&1
python3 tools/lds_from_pmc.py "$(find "$OUT/pmc_lds_c4" -name 'l_counter_collection.csv' | head -1)" "${TAG}_C4shard"
echo "[5b] LDS PMC pass of C4shard done"

# kernel stats of the other BASELINE configurations that fit one GPU (no PMC passes)
for cfg in C3 C4shard C5shard; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$cfg" -o x -- python3 $B --config $cfg --steps 10 > "$OUT/$cfg.log" 2>&1
    cp "$(find "$OUT/$cfg" -name 'x_kernel_stats.csv' | head -1)" "profiles/${TAG}_${cfg}_kernel_stats.csv"
    echo "[+] $cfg kernel stats done"
done
