# Round 3: what does the matrix-free J v wait for?  PMC passes (one counter group per run) on tools/matfree_bench.py jv 0 template.
# usage: bash tools/scripts/r03_matfree_pmc.sh        -> gpurun_out/r03/pmc_matfree_*.json
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r03
mkdir -p $O
run() {   # tag, counters...
  local tag=$1; shift
  rm -rf $O/pmc_mf_$tag
  rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_mf_$tag -- python3 $R/tools/matfree_bench.py jv 0 template > $O/pmc_mf_$tag.log 2>&1 < /dev/null
  python3 $R/tools/pmc_summary.py $O/pmc_mf_$tag ba_matfree > $O/pmc_matfree_$tag.json
  echo "== $tag"; cat $O/pmc_matfree_$tag.json
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
# (a TA_* pass hung rocprofv3 on this pool and was dropped)
run sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ
