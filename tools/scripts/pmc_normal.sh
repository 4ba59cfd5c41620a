# SQ counters of ba_normal_mfma_kernel on rig-32 (template chain); usage: bash tools/scripts/pmc_normal.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/pmc_nrm
rm -rf $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/a -- python3 $R/tools/normal_bench.py template --only-default > $O.a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/b -- python3 $R/tools/normal_bench.py template --only-default > $O.b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT --output-format csv -d $O/c -- python3 $R/tools/normal_bench.py template --only-default > $O.c.log 2>&1 || echo "pass c failed (counter names?)"
python3 $R/tools/pmc_summary.py $O/a ba_normal > $R/gpurun_out/pmc_nrm_a.json
python3 $R/tools/pmc_summary.py $O/b ba_normal > $R/gpurun_out/pmc_nrm_b.json
python3 $R/tools/pmc_summary.py $O/c ba_normal > $R/gpurun_out/pmc_nrm_c.json || true
