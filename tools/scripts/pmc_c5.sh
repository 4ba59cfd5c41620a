# Why does the FP64-arithmetic / FP32-bytes kernel slow down between 4e6 and 1e7 detections?  usage: bash tools/scripts/pmc_c5.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/pmc_c5
rm -rf $O; mkdir -p $O
for sc in 0.4 1.0; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/a_$sc -- python3 $R/tools/sweep.py --config 5 --dtype mixed --variants 6 --wgs 0 --rounds 2 --scale $sc --tag _pmc > $O/a_$sc.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum --output-format csv -d $O/b_$sc -- python3 $R/tools/sweep.py --config 5 --dtype mixed --variants 6 --wgs 0 --rounds 2 --scale $sc --tag _pmc > $O/b_$sc.log 2>&1 || echo "pass b failed"
  python3 $R/tools/pmc_summary.py $O/a_$sc ba_eval > $O/sum_a_$sc.json
  python3 $R/tools/pmc_summary.py $O/b_$sc ba_eval > $O/sum_b_$sc.json || true
done
cat $O/sum_a_0.4.json $O/sum_a_1.0.json $O/sum_b_0.4.json $O/sum_b_1.0.json
