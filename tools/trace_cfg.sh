#!/bin/bash
# trace_cfg.sh CONFIG TAG [extra bench args] - kernel trace + stats of one bench configuration and the kernel timeline
# of one steady-state iteration (tools/kernel_timeline.py); raw output under gpurun_out/prof_<TAG>_<CONFIG>/
set -eo pipefail
CFG=$1; TAG=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_${CFG}
mkdir -p "$OUT" "$ROOT/profiles"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o x -- python3 bench.py --config $CFG --steps 12 --warmup 3 --no-cpu-baseline --no-gap --no-c1 "$@" > "$OUT/bench.log" 2>&1
cp "$(find "$OUT" -name 'x_kernel_stats.csv' | head -1)" "profiles/${TAG}_${CFG}_kernel_stats.csv"
python3 tools/kernel_timeline.py "$(find "$OUT" -name 'x_kernel_trace.csv' | head -1)" > "$OUT/timeline.txt"
tail -1 "$OUT/bench.log" | head -c 400; echo
