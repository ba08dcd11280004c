#!/bin/bash
# A/B of round 3's z-step pieces against round 2's forms, interleaved on ONE box:  tools/ab_zstep.sh [out_dir] [configs...]
#   new = defaults;  old = RBL_PAV_NO_SEQ=1 RBL_PAV_UPPER_PERSIST=0 RBL_EHRM_SPEC=-1 RBL_SORT32=0
out=${1:-gpurun_out/ab_zstep}; shift
cfgs=${@:-C4shard}
mkdir -p $out
for cfg in $cfgs; do
  for rep in 1 2; do
    for mode in new old; do
      if [ $mode = old ]; then export RBL_PAV_NO_SEQ=1 RBL_PAV_UPPER_PERSIST=0 RBL_EHRM_SPEC=-1 RBL_SORT32=0; else unset RBL_PAV_NO_SEQ RBL_PAV_UPPER_PERSIST RBL_EHRM_SPEC RBL_SORT32; fi
      [ $cfg = C2sq_sort ] && export RBL_NO_ZBAND=1 || unset RBL_NO_ZBAND
      c=$cfg; [ $cfg = C2sq_sort ] && c=C2sq
      timeout -k 10 300 python bench.py --config $c --no-gap --no-cpu-baseline --no-c1 --steps 50 --phase-times \
          > $out/${cfg}_${mode}_r${rep}.json 2> $out/${cfg}_${mode}_r${rep}.err || exit 1
      python - $out/${cfg}_${mode}_r${rep}.json $cfg $mode <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = j["config"]["phase_ms_last"]
print("%-9s %-3s  %.2f it/s  %.3f ms/iter  z %.3f q %.3f w %.3f v %.3f ms" % (sys.argv[2], sys.argv[3], j["value"], j["ms_per_step"],
      p["z"], p["q"], p["w"], p["v"]))
PY
    done
  done
done
