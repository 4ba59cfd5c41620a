# Round 3: what bounds the small launch?  PMC passes (one counter group per run, no tracing beside them) of the one-launch
# step on rank 0's shard of an 8-way split of rig-32 (N = 124 801) and on ring-8 (config 2, N = 102 400).
# usage: bash tools/scripts/r03_small_pmc.sh        -> gpurun_out/r03/pmc_small_*.json
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r03
mkdir -p $O
run() {   # tag, counters..., then -- bench args
  local tag=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf $O/pmc_$tag
  rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-normal-probe "$@" > $O/pmc_$tag.log 2>&1 < /dev/null
  python3 $R/tools/pmc_summary.py $O/pmc_$tag ba_eval > $O/pmc_small_$tag.json
  echo "== $tag"; cat $O/pmc_small_$tag.json
}
for cfg in w8 c2; do
  if [ $cfg = w8 ]; then ARGS="--config 3 --emulate-world 8"; else ARGS="--config 2"; fi
  run ${cfg}_fetch FETCH_SIZE -- $ARGS
  run ${cfg}_write WRITE_SIZE -- $ARGS
  run ${cfg}_sq_a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -- $ARGS
  run ${cfg}_sq_b SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES -- $ARGS
done
