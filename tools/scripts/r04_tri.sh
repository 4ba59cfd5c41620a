# triangulation (row f4): round-3 kernel vs round-4 kernel on one box + HBM traffic of the new one.  -> gpurun_out/r04/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
PCS_TRI_VARIANT=0 timeout -k 10 200 python tools/tri_bench.py 2>&1 | grep -v amdgpu | head -9 > $O/tri_bench_r03_kernel.log
timeout -k 10 200 python tools/tri_bench.py 2>&1 | grep -v amdgpu | head -9 > $O/tri_bench.log
PCS_TRI_NO_SORT=1 timeout -k 10 200 python tools/tri_bench.py 2>&1 | grep "device in" | sed 's/^/  [points in table order] /' >> $O/tri_bench.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_tri_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_tri_$c -- python3 $R/tools/tri_bench.py > $O/pmc_tri_$c.log 2>&1 < /dev/null
done
mkdir -p $O/pmc_tri; rm -rf $O/pmc_tri/*; mv $O/pmc_tri_FETCH_SIZE $O/pmc_tri/fetch; mv $O/pmc_tri_WRITE_SIZE $O/pmc_tri/write
python3 $R/tools/pmc_summary.py $O/pmc_tri triangulate_reg > $O/pmc_traffic_triangulate.json 2>/dev/null < /dev/null
rm -rf $O/pmc_tri $O/pmc_tri_*.log
cat $O/tri_bench_r03_kernel.log $O/tri_bench.log $O/pmc_traffic_triangulate.json
