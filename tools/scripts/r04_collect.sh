# Collect the round-4 measurements (one MI355X).  usage: bash tools/scripts/r04_collect.sh [part ...]   -> gpurun_out/r04/final/
# parts: bench stats pmc lm chol tri gen gloo scaling (default: all)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04/final
mkdir -p $O
cd $R
PARTS="${@:-bench stats pmc lm chol tri gen gloo scaling}"
say() { echo "[r04_collect] $*"; }
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
say "device LM: phases, kernel stats and the timeline of one trial"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_lm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lm -- python3 $R/tools/lm_profile.py --config 3 > $O/prof_lm.log 2>&1 < /dev/null
f=$(find $O/prof_lm -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/lm_rig32_kernel_stats.csv
f=$(find $O/prof_lm -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python3 $R/tools/lm_trace.py "$f" > $O/lm_trace_rig32.log
rm -rf $O/prof_lm
rm -rf $O/prof_lm4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lm4 -- python3 $R/tools/lm_profile.py --config 4 --chain self --reps 5 > $O/prof_lm4.log 2>&1 < /dev/null
f=$(find $O/prof_lm4 -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python3 $R/tools/lm_trace.py "$f" > $O/lm_trace_rig32_self.log
rm -rf $O/prof_lm4
cd $R
timeout -k 10 300 python tools/lm_profile.py --config 3 2>&1 < /dev/null | grep -v amdgpu > $O/lm_profile_rig32.log
timeout -k 10 300 python tools/lm_profile.py --config 4 --chain self --reps 5 2>&1 < /dev/null | grep -v amdgpu > $O/lm_profile_rig32_self.log
timeout -k 10 300 python tools/lm_profile.py --config 2 2>&1 < /dev/null | grep -v amdgpu > $O/lm_profile_ring8.log
fi
if has chol; then
say "one-launch Cholesky: per-column trace and timings"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pycamset_amd/csrc -o tools/probes/chol_persist_probe tools/probes/chol_persist_probe.hip > /dev/null 2>&1
CP_SHOW_TRACE=1 timeout -k 5 100 tools/probes/chol_persist_probe 480 > $O/chol_persist_trace_480.log 2>&1
CP_SHOW_TRACE=1 timeout -k 5 100 tools/probes/chol_persist_probe 1680 > $O/chol_persist_trace_1680.log 2>&1
timeout -k 5 100 tools/probes/chol_persist_probe 1 31 33 97 480 1003 1680 > $O/chol_persist_times.log 2>&1
timeout -k 10 100 python tools/syrk_bench.py 2>&1 < /dev/null | grep -v amdgpu > $O/syrk_bench.log
fi
if has tri; then
say "triangulation"; bash tools/scripts/r04_tri.sh > /dev/null 2>&1
cp $R/gpurun_out/r04/tri_bench_r03_kernel.log $R/gpurun_out/r04/tri_bench.log $R/gpurun_out/r04/pmc_traffic_triangulate.json $O/ 2>/dev/null
fi
if has gen; then
say "generated chain: one launch against two"; timeout -k 10 300 python tools/genchain_forms.py 2>&1 < /dev/null | grep -v amdgpu > $O/genchain_forms.log
fi
if has gloo; then
say "bench.py --gpus N without a launcher (gloo rehearsal on one GPU: the ranks share the card)"
for n in 2 4; do
  PCS_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus $n --steps 50 --warmup 5 > $O/bench_N${n}_selflaunch_gloo_one_gpu.json 2> $O/bench_N${n}_gloo.err < /dev/null
done
fi
say "done"; ls $O
