# Normal-equations measurements of round 2 (one MI355X).  usage: bash tools/scripts/r02_normal.sh  -> gpurun_out/r02/normal/
R=/root/repo
O=$R/gpurun_out/r02/normal
mkdir -p $O
cd $R
echo "[r02_normal] normal_bench"; timeout -k 10 300 python tools/normal_bench.py --only-default > $O/normal_bench.log 2>&1 < /dev/null
echo "[r02_normal] phases, template"; timeout -k 10 200 python tools/normal_quick.py template 0,2,8,16,32,64,24,88 0 > $O/normal_phases_template.log 2>&1 < /dev/null
echo "[r02_normal] passes + phases, self"
timeout -k 10 300 python tools/normal_quick.py self 0,1536,1552,1544,1538,1280,1296,1288,1282,768,784,776,770 0 > $O/normal_phases_self.log 2>&1 < /dev/null
echo "[r02_normal] walk variant of the pose-point pass"; timeout -k 10 200 python tools/normal_quick.py self 0,768 0 --walk > $O/normal_self_walk.log 2>&1 < /dev/null
echo "[r02_normal] free"; timeout -k 10 200 python tools/normal_quick.py free 0,1536,1280 0 > $O/normal_phases_free.log 2>&1 < /dev/null
for chain in template self free; do
  echo "[r02_normal] rocprofv3 stats $chain"
  timeout -k 10 300 bash tools/scripts/prof_stats.sh normal_$chain $R/tools/normal_quick.py $chain 0 0 > $O/prof_$chain.log 2>&1 < /dev/null
  cp $R/gpurun_out/r02/normal_${chain}_kernel_stats.csv $O/ 2>/dev/null
done
echo "[r02_normal] done"; ls $O
