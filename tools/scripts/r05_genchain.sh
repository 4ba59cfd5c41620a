#!/bin/bash
# Round 5: the exact LM on a generated chain — timings (tools/genchain_lm.py) and the per-kernel trace of one solve per config.
# Run on the GPU box from the repo root: bash tools/scripts/r05_genchain.sh [times] [trace]
set -e
OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
what=${@:-times trace}
for w in $what; do
  case $w in
    times)
      timeout -k 10 800 python tools/genchain_lm.py --config 1 2 3 2>&1 | grep -v amdgpu > $OUT/genchain_lm.log
      timeout -k 10 500 python tools/genchain_lm.py --config 1 2 3 --chain division 2>&1 | grep -v amdgpu > $OUT/genchain_division.log
      for c in division flex divfree; do timeout -k 10 300 python tools/genchain_lm.py --config 1 2 3 --chain $c --no-cg --phases 2>&1 | grep -v amdgpu; done > $OUT/genchain_final_phases.log
      for c in division flex divfree; do timeout -k 10 300 python tools/genchain_lm.py --config 1 2 3 --chain $c --no-cg --deterministic 2>&1 | grep -v amdgpu; done > $OUT/genchain_deterministic.log ;;
    trace)
      cd /tmp && export TMPDIR=/tmp
      for spec in "1 flex c1" "3 flex c3" "3 division division_c3"; do
        set -- $spec
        rm -rf /tmp/gtrace_$3
        timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gtrace_$3 -o t -- python3 $OLDPWD/tools/genchain_lm.py --config $1 --chain $2 --trace > $OUT/genchain_trace_$3.log 2>&1 < /dev/null
        f=$(find /tmp/gtrace_$3 -name "*kernel_stats.csv" | head -1)
        cp "$f" $OUT/genchain_$3_kernel_stats.csv
      done
      cd $OLDPWD ;;
  esac
done
