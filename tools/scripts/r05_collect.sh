# Collect the round-5 measurements (one MI355X).  usage: bash tools/scripts/r05_collect.sh [part ...]   -> gpurun_out/r05/final/
# parts: bench stats pmc lm far tri normal gloo scaling (default: all)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05/final
mkdir -p $O
cd $R
PARTS="${@:-bench stats pmc lm far tri normal gloo scaling}"
say() { echo "[r05_collect] $*"; }
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has bench; then
say "bench config 3 (default run)"; timeout -k 10 300 python bench.py > $O/bench_N1.json 2> $O/bench_N1.err < /dev/null
say "bench config 2"; timeout -k 10 200 python bench.py --config 2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_N1_config2_ring8.json 2> $O/bench_c2.err < /dev/null
say "bench config 4"; timeout -k 10 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_N1_config4_self.json 2> $O/bench_c4.err < /dev/null
say "bench config 5 f32"; timeout -k 10 400 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_N1_config5_f32.json 2> $O/bench_c5.err < /dev/null
fi
if has scaling; then
say "scaling projection"; bash tools/scaling_projection.sh > $O/scaling_projection_one_gpu.log 2>&1 < /dev/null
fi
if has stats; then
say "rocprofv3 kernel stats of the bench command"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_bench
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-normal-probe > $O/prof_bench.log 2>&1 < /dev/null
f=$(find $O/prof_bench -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/bench_N1_kernel_stats.csv
rm -rf $O/prof_bench
cd $R
fi
if has pmc; then
say "PMC traffic of the headline kernel (separate passes per counter)"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_c3_$c
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_c3_$c -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-normal-probe > $O/pmc_c3_$c.log 2>&1 < /dev/null
done
mkdir -p $O/pmc_c3; rm -rf $O/pmc_c3/*; mv $O/pmc_c3_FETCH_SIZE $O/pmc_c3/fetch; mv $O/pmc_c3_WRITE_SIZE $O/pmc_c3/write
python3 $R/tools/pmc_summary.py $O/pmc_c3 ba_eval > $O/pmc_traffic_c3.json 2>/dev/null < /dev/null
rm -rf $O/pmc_c3 $O/pmc_c3_*.log
cd $R
fi
if has lm; then
say "device LM: the four forms (fused x deterministic), then the kernel timeline of one trial per form"
for cfg in "3 template" "2 template" "4 self" "1 template"; do
  set -- $cfg
  timeout -k 10 300 python tools/lm_modes.py --config $1 --chain $2 2>&1 < /dev/null | grep -v amdgpu >> $O/lm_modes.log
done
cd /tmp && export TMPDIR=/tmp
for spec in "3 template rig32" "4 self rig32_self" "2 template ring8" "1 template config1"; do
  set -- $spec
  for form in 10 11 00; do
    rm -rf $O/prof_lm
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lm -- python3 $R/tools/lm_modes.py --config $1 --chain $2 --trace --forms $form > $O/prof_lm.log 2>&1 < /dev/null
    f=$(find $O/prof_lm -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python3 $R/tools/lm_trace.py "$f" > $O/lm_trace_$3_form$form.log
    if [ "$form" = "10" ] && [ "$3" = "rig32" ]; then f=$(find $O/prof_lm -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/lm_rig32_kernel_stats.csv; fi
    rm -rf $O/prof_lm
  done
done
cd $R
fi
if has far; then
say "far starts: evaluations per damping policy"
timeout -k 10 900 python tools/lm_far_start.py 2>&1 < /dev/null | grep -v amdgpu > $O/lm_far_start.log
fi
if has tri; then
say "triangulation: lanes sweep + SQ counters"; bash tools/scripts/r05_tri.sh > /dev/null 2>&1
cp $R/gpurun_out/r05/tri_sweep.log $R/gpurun_out/r05/tri_sq_counters_a.json $R/gpurun_out/r05/tri_sq_counters_b.json $O/ 2>/dev/null
fi
if has normal; then
say "normal equations: SQ counters of the round-5 kernel (atomics and deterministic)"
cd /tmp && export TMPDIR=/tmp
for det in 0 1; do
for tag in a b; do
  if [ $tag = a ]; then ctr="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; else ctr="SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; fi
  rm -rf $O/pmc_n
  PCS_NORMAL_DET=$det timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_n -- python3 $R/tools/normal_quick.py > $O/pmc_n.log 2>&1 < /dev/null
  python3 $R/tools/pmc_summary.py $O/pmc_n normal > $O/normal_kernel_sq_counters_det${det}_$tag.json 2>/dev/null < /dev/null
  rm -rf $O/pmc_n $O/pmc_n.log
done
done
cd $R
fi
if has gloo; then
say "bench.py --gpus N without a launcher (gloo rehearsal on one GPU: the ranks share the card)"
for n in 2 4; do
  PCS_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus $n --steps 50 --warmup 5 > $O/bench_N${n}_selflaunch_gloo_one_gpu.json 2> $O/bench_N${n}_gloo.err < /dev/null
done
fi
say "done"; ls $O
