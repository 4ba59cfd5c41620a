# HBM traffic of the fused kernel by PMC counters, separate passes (MI355X_MICROARCH.md HBM section).
# usage: bash tools/scripts/pmc_traffic.sh <tag> <bench.py args...>     e.g.  c4_self --config 4 --chain self
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/pmc_traffic_$TAG
mkdir -p $R/gpurun_out; rm -rf $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-normal-probe "$@" > $O.fetch.log 2>&1 < /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-normal-probe "$@" > $O.write.log 2>&1 < /dev/null
python3 $R/tools/pmc_summary.py $O ba_eval > $R/gpurun_out/pmc_traffic_$TAG.json
cat $R/gpurun_out/pmc_traffic_$TAG.json
