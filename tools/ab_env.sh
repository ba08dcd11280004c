#!/bin/bash
# Generic interleaved A/B of one environment switch on ONE box:  tools/ab_env.sh OUT_DIR VAR=VALUE config [config ...]
#   "new" = the variable unset (defaults), "old" = VAR=VALUE (e.g. RBL_GEMVT_SWEEP=0, RBL_WSTEP_PERSIST=0)
out=$1; kv=$2; shift 2
mkdir -p $out
for cfg in "$@"; do
  for rep in 1 2; do
    for mode in new old; do
      if [ $mode = old ]; then export "$kv"; else unset "${kv%%=*}"; fi
      timeout -k 10 300 python bench.py --config $cfg --no-gap --no-cpu-baseline --no-c1 --steps 50 \
          > $out/${cfg}_${mode}_r${rep}.json 2> $out/${cfg}_${mode}_r${rep}.err || exit 1
      python - $out/${cfg}_${mode}_r${rep}.json $cfg $mode "$kv" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = j["roofline"]["kernels"]
print("%-9s %-3s (%s)  %.2f it/s  %.3f ms/iter  steady %.3f ms  gemvt %.3f ms  sweep %.3f ms  inner %d" % (
    sys.argv[2], sys.argv[3], sys.argv[4] if sys.argv[3] == "old" else "default", j["value"], j["ms_per_step"],
    j["roofline"]["steady_state"]["ms_per_step_median"], k["gemvt"]["avg_ms"], max(k["gemv"]["avg_ms"], k["sweep_erm"]["avg_ms"]),
    j["config"]["inner_iters_last"]))
PY
    done
  done
  unset "${kv%%=*}"
done
