# Round 5, triangulation (row f4): what is the 50 us?  (1) lanes-per-point sweep of the register kernel, (2) SQ counter passes.
# usage: bash tools/scripts/r05_tri.sh   -> gpurun_out/r05/tri_*.log, tri_sq_counters_*.json
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
: > $O/tri_sweep.log
for lanes in 1 2 4 8; do
  for var in 1 3; do
    echo "== PCS_TRI_LANES=$lanes PCS_TRI_VARIANT=$var" >> $O/tri_sweep.log
    PCS_TRI_LANES=$lanes PCS_TRI_VARIANT=$var timeout -k 10 200 python tools/tri_bench.py 2>&1 | grep "device in\|max rel diff" >> $O/tri_sweep.log
  done
done
cd /tmp && export TMPDIR=/tmp
run() {   # tag, counters...
  local tag=$1; shift
  rm -rf $O/pmc_tri_$tag
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_tri_$tag -- python3 $R/tools/tri_bench.py > $O/pmc_tri_$tag.log 2>&1 < /dev/null
  python3 $R/tools/pmc_summary.py $O/pmc_tri_$tag triangulate_reg > $O/tri_sq_counters_$tag.json 2>/dev/null < /dev/null
  rm -rf $O/pmc_tri_$tag $O/pmc_tri_$tag.log
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_LDS
cat $O/tri_sweep.log $O/tri_sq_counters_a.json $O/tri_sq_counters_b.json
