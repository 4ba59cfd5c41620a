# A/B two builds of libpcs_hip.so on one box: bash gpurun_ab/ab.sh "<command>" [rounds]
cmd="$1"; n=${2:-3}
for i in $(seq $n); do
  for v in old new; do
    cp gpurun_ab/$v.so pycamset_amd/libpcs_hip.so
    echo "== $v"; eval "$cmd" 2>&1 | grep -v amdgpu | tail -2
  done
done
cp gpurun_ab/new.so pycamset_amd/libpcs_hip.so
