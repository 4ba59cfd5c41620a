# Collect the round-3 measurements (one MI355X).  usage: bash tools/scripts/r03_collect.sh [part ...]   -> gpurun_out/r03/final/
# parts: bench scaling stats pmc normal lm gloo (default: all)
R=/root/repo
O=$R/gpurun_out/r03/final
mkdir -p $O
cd $R
PARTS="${@:-bench scaling stats pmc normal lm gloo}"
say() { echo "[r03_collect] $*"; }
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has bench; then
say "bench config 3 (default run)"; timeout -k 10 300 python bench.py > $O/bench_N1.json 2> $O/bench_N1.err < /dev/null
say "bench config 2"; timeout -k 10 200 python bench.py --config 2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_N1_config2_ring8.json 2> $O/bench_c2.err < /dev/null
say "bench config 4"; timeout -k 10 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_N1_config4_self.json 2> $O/bench_c4.err < /dev/null
say "bench config 5 f32"; timeout -k 10 400 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_N1_config5_f32.json 2> $O/bench_c5.err < /dev/null
say "bench config 5 mixed"; timeout -k 10 400 python bench.py --config 5 --dtype mixed --steps 20 --warmup 3 --no-cpu-baseline --no-normal-probe > $O/bench_N1_config5_mixed.json 2> $O/bench_c5m.err < /dev/null
say "bench config 5 f32, Jacobian streamed to host"; timeout -k 10 400 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline --no-normal-probe --stream-to-host > $O/bench_N1_config5_f32_stream_to_host.json 2> $O/bench_c5s.err < /dev/null
fi
if has scaling; then
say "scaling projection"; bash tools/scaling_projection.sh > $O/scaling_projection_one_gpu.log 2>&1 < /dev/null
say "small steps A/B"; timeout -k 10 300 python tools/small_step.py --config 3 --worlds 8,4,2,1 --variants 6 --lazy 0,1 --tag _final 2>&1 < /dev/null | grep -v amdgpu > $O/small_step_c3.log
timeout -k 10 200 python tools/small_step.py --config 2 --worlds 1 --variants 6 --lazy 0,1 --tag _final 2>&1 < /dev/null | grep -v amdgpu > $O/small_step_c2.log
fi
if has stats; then
say "rocprofv3 kernel stats of the bench command"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_bench
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-normal-probe > $O/prof_bench.log 2>&1 < /dev/null
f=$(find $O/prof_bench -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/bench_N1_kernel_stats.csv
rm -rf $O/prof_bench_c2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench_c2 -- python3 $R/bench.py --config 2 --steps 200 --warmup 20 --no-cpu-baseline --no-normal-probe > $O/prof_bench_c2.log 2>&1 < /dev/null
f=$(find $O/prof_bench_c2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/bench_N1_config2_kernel_stats.csv
cd $R
fi
if has pmc; then
say "PMC traffic (separate passes per counter)"
cd /tmp && export TMPDIR=/tmp
for tag in c3 c2 c4 c5mixed; do
  case $tag in c3) A="";; c2) A="--config 2";; c4) A="--config 4";; c5mixed) A="--config 5 --dtype mixed --steps 10 --warmup 2";; esac
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${tag}_$c
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${tag}_$c -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-normal-probe $A > $O/pmc_${tag}_$c.log 2>&1 < /dev/null
  done
  mkdir -p $O/pmc_$tag; rm -rf $O/pmc_$tag/*; mv $O/pmc_${tag}_FETCH_SIZE $O/pmc_$tag/fetch; mv $O/pmc_${tag}_WRITE_SIZE $O/pmc_$tag/write
  python3 $R/tools/pmc_summary.py $O/pmc_$tag ba_eval > $O/pmc_traffic_$tag.json 2>/dev/null < /dev/null
done
cd $R
fi
if has normal; then
say "normal equations: dense vs blocked build"; timeout -k 10 300 python tools/blocked_bench.py 2>&1 < /dev/null | grep -v amdgpu > $O/blocked_bench.log
say "normal equations: SQ counters of the shipped kernel (three passes)"
cd /tmp && export TMPDIR=/tmp
P=$O/pmc_nrm; rm -rf $P
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $P/a -- python3 $R/tools/normal_bench.py template --only-default > $P.a.log 2>&1 < /dev/null
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $P/b -- python3 $R/tools/normal_bench.py template --only-default > $P.b.log 2>&1 < /dev/null
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT --output-format csv -d $P/c -- python3 $R/tools/normal_bench.py template --only-default > $P.c.log 2>&1 < /dev/null || echo "pass c failed"
for x in a b c; do python3 $R/tools/pmc_summary.py $P/$x ba_normal > $O/normal_kernel_sq_counters_$x.json 2>/dev/null < /dev/null; done
say "matrix-free products: SQ counters"
P=$O/pmc_mf; rm -rf $P
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $P/a -- python3 $R/tools/matfree_bench.py > $P.a.log 2>&1 < /dev/null
python3 $R/tools/pmc_summary.py $P/a ba_matfree > $O/matfree_sq_counters.json 2>/dev/null < /dev/null
cd $R
timeout -k 10 300 python tools/matfree_bench.py 2>&1 < /dev/null | grep -v amdgpu > $O/matfree_bench.log
fi
if has lm; then
say "device LM: phases + rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_lm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lm -- python3 $R/tools/lm_profile.py --config 3 > $O/prof_lm.log 2>&1 < /dev/null
f=$(find $O/prof_lm -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/lm_rig32_kernel_stats.csv
cd $R
timeout -k 10 300 python tools/lm_profile.py --config 3 2>&1 < /dev/null | grep -v amdgpu > $O/lm_profile_rig32.log
timeout -k 10 300 python tools/lm_profile.py --config 4 --chain self --reps 5 2>&1 < /dev/null | grep -v amdgpu > $O/lm_profile_rig32_self.log
timeout -k 10 300 python tools/lm_profile.py --config 2 2>&1 < /dev/null | grep -v amdgpu > $O/lm_profile_ring8.log
fi
if has gloo; then
say "gloo rehearsal of bench.py --gpus N on one GPU (ranks share the card; the box admits at most 6 GPU processes INCLUDING the launcher: a 6-rank rehearsal was killed by its process guard)"
for n in 2 4; do
  PCS_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --steps 50 --warmup 5 > $O/bench_N${n}_gloo_rehearsal_one_gpu.json 2> $O/bench_N${n}_gloo.err < /dev/null
done
fi
say "done"; ls $O
