#!/bin/bash
# bench_all.sh OUTDIR - the committed bench lines of a round: plain `python bench.py` for every single-GPU
# configuration (C2 with the CPU baseline and the C1 sub-record, the others with their F* / gap record).
set -eo pipefail
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/bench_all}
mkdir -p "$OUT"
python bench.py --steps 20 --warmup 5 > "$OUT/C2_driver.json" 2> "$OUT/C2_driver.err"; echo "C2 as the driver runs it (--steps 20 --warmup 5) $(cut -c90-130 $OUT/C2_driver.json)"
python bench.py > "$OUT/C2.json" 2> "$OUT/C2.err"; echo "C2 $(cut -c90-130 $OUT/C2.json)"
python bench.py --storage f64 --no-cpu-baseline --no-c1 > "$OUT/C2_f64.json" 2> "$OUT/C2_f64.err"; echo "C2_f64 $(cut -c90-130 $OUT/C2_f64.json)"
for c in C2hinge C2l2 C2smooth C5shard C2sq C3 C4shard; do
    python bench.py --config $c --no-cpu-baseline --no-c1 > "$OUT/$c.json" 2> "$OUT/$c.err"; echo "$c $(cut -c90-130 $OUT/$c.json)"
done
python bench.py --no-cpu-baseline --no-c1 --no-gap > "$OUT/C2_again.json" 2> /dev/null; echo "C2 again $(cut -c90-130 $OUT/C2_again.json)"
