# SQ counters of ba_normal_kernel on rig-32 (template chain); usage: bash tools/scripts/pmc_normal.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/pmc_nrm
rm -rf $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $O/a -- python3 $R/tools/normal_bench.py template --only-default > $O.a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/b -- python3 $R/tools/normal_bench.py template --only-default > $O.b.log 2>&1
python3 $R/tools/pmc_summary.py $O/a ba_normal > $R/gpurun_out/pmc_nrm_a.json
python3 $R/tools/pmc_summary.py $O/b ba_normal > $R/gpurun_out/pmc_nrm_b.json
