#!/bin/bash
# The multi-GPU driver's own per-iteration cost at one rank's share of C2 on 8 GPUs (750 000 rows), measured on ONE GPU:
#   tools/ab_driver.sh OUT_DIR "VAR=VAL" ...     ("-" = defaults); bench.py --sharded-driver, 200 timed iterations, interleaved twice
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    f=$out/drv_s${i}_r${rep}
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    env $envs timeout -k 10 300 python bench.py --sharded-driver --rows 750000 --steps 200 --no-gap --no-cpu-baseline --no-c1 > $f.json 2> $f.err || { echo "FAILED: $setting"; tail -3 $f.err; exit 1; }
    python - $f.json "$setting" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = j["roofline"]["kernels"]["sweep_erm"]
print("%-28s %8.1f it/s  %.4f ms/iter  sweep %.4f ms  outside the sweep %.1f us  collectives/iter %.1f" % (
    sys.argv[2], j["value"], j["ms_per_step"], k["avg_ms"], (j["ms_per_step"] - k["avg_ms"]) * 1e3,
    j["config"]["collectives_per_iteration"]))
PY
  done
done
