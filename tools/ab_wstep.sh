#!/bin/bash
# A/B of the one-launch persistent w-step (RBL_WSTEP_PERSIST, default on) against the batched launches, interleaved
# on ONE box (boxes differ by +-3 %):  tools/ab_wstep.sh [out_dir] [configs...]
out=${1:-gpurun_out/ab_wstep}; shift
cfgs=${@:-C2l2 C2smooth C2sq C4shard}
mkdir -p $out
for cfg in $cfgs; do
  for rep in 1 2; do
    for mode in 1 0; do
      RBL_WSTEP_PERSIST=$mode timeout -k 10 300 python bench.py --config $cfg --no-gap --no-cpu-baseline --no-c1 --steps 50 \
          > $out/${cfg}_p${mode}_r${rep}.json 2> $out/${cfg}_p${mode}_r${rep}.err || exit 1
      python - $out/${cfg}_p${mode}_r${rep}.json $cfg $mode <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-9s persist=%s  %.2f it/s  %.3f ms/iter  inner %d  steady %.3f ms" % (sys.argv[2], sys.argv[3], j["value"], j["ms_per_step"],
      j["config"]["inner_iters_last"], j["roofline"]["steady_state"]["ms_per_step_median"]))
PY
    done
  done
done
