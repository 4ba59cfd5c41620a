# Collect the round-2 measurements (one MI355X).  usage: bash tools/scripts/r02_collect.sh   -> gpurun_out/r02/final/
R=/root/repo
O=$R/gpurun_out/r02/final
mkdir -p $O
cd $R
say() { echo "[r02_collect] $*"; }
say "bench config 3 (default run)"; timeout -k 10 300 python bench.py > $O/bench_N1.json 2> $O/bench_N1.err < /dev/null
say "bench config 4"; timeout -k 10 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_N1_config4_self.json 2> $O/bench_c4.err < /dev/null
say "bench config 5 f32"; timeout -k 10 400 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_N1_config5_f32.json 2> $O/bench_c5.err < /dev/null
say "bench config 5 mixed"; timeout -k 10 400 python bench.py --config 5 --dtype mixed --steps 20 --warmup 3 --no-cpu-baseline --no-normal-probe > $O/bench_N1_config5_mixed.json 2> $O/bench_c5m.err < /dev/null
say "bench config 2"; timeout -k 10 200 python bench.py --config 2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_N1_config2_ring8.json 2> $O/bench_c2.err < /dev/null
say "scaling projection"; bash tools/scaling_projection.sh > $O/scaling_projection_one_gpu.log 2>&1 < /dev/null
say "sweeps"
for args in "--config 3" "--config 4 --chain self" "--config 4 --chain free" "--config 3 --dtype f32" "--config 3 --dtype mixed" "--config 3 --shuffle" "--config 3 --mode resid --variants 4" "--config 2"; do
  timeout -k 10 200 python tools/sweep.py $args --wgs 0 --rounds 7 --tag _final 2>&1 < /dev/null | grep -v amdgpu >> $O/sweeps_default_geometry.log
done
say "normal equations"; timeout -k 10 300 python tools/normal_bench.py --only-default > $O/normal_bench.log 2>&1 < /dev/null
say "compact"; timeout -k 10 200 python tools/compact_bench.py > $O/compact_bench.log 2>&1 < /dev/null
say "rocprofv3 kernel stats of the bench command"
timeout -k 10 300 bash tools/scripts/prof_stats.sh bench_N1_final $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-normal-probe > $O/prof_stats.log 2>&1 < /dev/null
cp $R/gpurun_out/r02/bench_N1_final_kernel_stats.csv $O/bench_N1_kernel_stats.csv 2>/dev/null
say "PMC traffic (separate passes)"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-normal-probe > $O/pmc_$c.log 2>&1 < /dev/null
done
python3 $R/tools/pmc_summary.py $O ba_eval > $O/bench_N1_pmc_summary.json 2>/dev/null < /dev/null
say "done"; ls $O
