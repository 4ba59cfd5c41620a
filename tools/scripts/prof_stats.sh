# rocprofv3 kernel stats of one command; usage: bash tools/scripts/prof_stats.sh <tag> <python script + args...>
# writes gpurun_out/r02/<tag>_kernel_stats.csv and prints its head
set -e
TAG=$1; shift
R=/root/repo
mkdir -p $R/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/prof_$TAG -- python3 "$@" > $R/gpurun_out/r02/prof_$TAG.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/r02/prof_$TAG -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/r02/${TAG}_kernel_stats.csv; cut -c1-220 "$f" | head -14; else echo "no stats file"; tail -5 $R/gpurun_out/r02/prof_$TAG.log; fi
