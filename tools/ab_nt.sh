#!/bin/bash
# Interleaved A/B on one box: cache policy of the streaming loads of D (sweep_erm.hip: RBL_D_AUX; sweep.hip:
# RBL_D_STREAM).  Builds the variants next to each other and alternates them (a fresh box runs its first minute
# a few per cent slower, so back-to-back blocks of one variant mislead).  tools/ab_nt.sh CONFIG "flagsA" "flagsB" ...
set -eo pipefail
cd "$(dirname "$0")/.."
CFG=$1; shift
LIB=admm-for-rank-based-loss_amd/csrc/librbl.so
i=0
for flags in "$@"; do
    touch admm-for-rank-based-loss_amd/csrc/sweep_erm.hip admm-for-rank-based-loss_amd/csrc/sweep.hip
    RBL_HIPCC_FLAGS="$flags" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
    cp $LIB /tmp/librbl_$i.so; i=$((i+1))
done
B="bench.py --no-cpu-baseline --no-gap --no-c1 --steps 60 --warmup 5 --config $CFG"
for rep in 1 2 3 4; do
    j=0
    for flags in "$@"; do
        cp /tmp/librbl_$j.so $LIB
        printf "%-40s " "$flags"; python $B 2>/dev/null | cut -c90-130
        j=$((j+1))
    done
done
