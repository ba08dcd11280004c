#!/bin/bash
# Interleaved comparison of several environment settings on ONE box:
#   tools/ab_multi.sh OUT_DIR CONFIG "VAR=VAL VAR2=VAL" "VAR=VAL" ...      ("-" = defaults)
# AB_ARGS="--storage f64" adds bench.py arguments to every run
# two rounds over all settings, bench.py --steps 50 each; one summary line per run
out=$1; cfg=$2; shift 2
mkdir -p $out
for rep in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    f=$out/${cfg}_s${i}_r${rep}
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    env $envs timeout -k 10 300 python bench.py --config $cfg $AB_ARGS --no-gap --no-cpu-baseline --no-c1 --steps 50 > $f.json 2> $f.err || { echo "FAILED: $setting"; tail -3 $f.err; exit 1; }
    python - $f.json $cfg "$setting" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = j["roofline"]["kernels"]
print("%-6s %-52s %7.2f it/s  %.3f ms/iter  steady %.3f ms  gemvt %.3f  sweep %.3f  (first %.3f last %.3f)" % (
    sys.argv[2], sys.argv[3], j["value"], j["ms_per_step"], j["roofline"]["steady_state"]["ms_per_step_median"],
    k["gemvt"]["avg_ms"], max(k["gemv"]["avg_ms"], k["sweep_erm"]["avg_ms"]),
    j["roofline"]["kernel_ms_first"], j["roofline"]["kernel_ms_last"]))
PY
  done
done
