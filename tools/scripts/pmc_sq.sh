set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/pmc3
for chain in template self free; do
  cfg=3; [ $chain != template ] && cfg=4
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $O/${chain}_a -- python3 $R/bench.py --config $cfg --chain $chain --steps 10 --warmup 3 --no-cpu-baseline > $O.${chain}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${chain}_b -- python3 $R/bench.py --config $cfg --chain $chain --steps 10 --warmup 3 --no-cpu-baseline > $O.${chain}_b.log 2>&1
  python3 $R/tools/pmc_summary.py $O/${chain}_a > $R/gpurun_out/pmc3_${chain}_a.json
  python3 $R/tools/pmc_summary.py $O/${chain}_b > $R/gpurun_out/pmc3_${chain}_b.json
done
