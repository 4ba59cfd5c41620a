#!/bin/bash
# Round 5: what does the matrix-free J^T (J v) wait for?  Two PMC passes (one counter group per run) on tools/matfree_bench.py jtjv 0 template.
# usage (GPU box, repo root): bash tools/scripts/r05_matfree_pmc.sh   -> gpurun_out/r05/matfree_jtjv_sq_{a,b}.json
set -e
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # tag, counters...
  local tag=$1; shift
  rm -rf /tmp/pmc_mf_$tag
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_mf_$tag -- python3 $R/tools/matfree_bench.py jtjv 0 template > $O/pmc_mf_$tag.log 2>&1 < /dev/null
  python3 $R/tools/pmc_summary.py /tmp/pmc_mf_$tag ba_matfree > $O/matfree_jtjv_sq_$tag.json
  echo "== $tag"; cat $O/matfree_jtjv_sq_$tag.json
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD
run b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU
