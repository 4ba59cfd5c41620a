set -e
cp pycamset_amd/libpcs_hip.so ab/libpcs_new.so
one() { python bench.py --no-cpu-baseline --no-normal-probe "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  kernel %.2f us  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['frac']))"; }
for rep in 1 2 3; do
  for which in new old; do
    cp ab/libpcs_$which.so pycamset_amd/libpcs_hip.so
    echo "$which T: $(one)"; echo "$which S: $(one --config 4 --chain self)"; echo "$which F: $(one --config 4 --chain free)"; echo "$which T f32: $(one --dtype f32)"
  done
done
cp ab/libpcs_new.so pycamset_amd/libpcs_hip.so
