#!/bin/bash
# Round 5: the exact LM on a generated chain — timings (tools/genchain_lm.py) and the per-kernel trace of one solve per config.
# Run on the GPU box from the repo root: bash tools/scripts/r05_genchain.sh [times] [trace]
set -e
OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
what=${@:-times trace}
for w in $what; do
  case $w in
    times) timeout -k 10 800 python tools/genchain_lm.py --config 1 2 3 > $OUT/genchain_lm.log 2>&1 ;;
    trace)
      cd /tmp && export TMPDIR=/tmp
      for c in 1 3; do
        rm -rf /tmp/gtrace_$c
        timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gtrace_$c -o t -- python3 $OLDPWD/tools/genchain_lm.py --config $c --trace > $OUT/genchain_trace_c$c.log 2>&1 < /dev/null
        f=$(find /tmp/gtrace_$c -name "*kernel_stats.csv" | head -1)
        cp "$f" $OUT/genchain_c${c}_kernel_stats.csv
      done
      cd $OLDPWD ;;
  esac
done
