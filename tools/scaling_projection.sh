# One-GPU projection of the strong-scaling curve: step time of rank 0's shard of a K-way split (bench.py --emulate-world K).
for k in 1 2 4 8; do
  timeout -k 10 120 python bench.py --emulate-world $k --no-cpu-baseline --no-normal-probe "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('world %d' % d['config'].get('emulated_world', 1), 'N', d['config']['detections_total'], 'ms/step %.4f' % d['ms_per_step'], 'kernel %.4f (isolated %.4f, min %.4f)' % (r['kernel_ms'], r['kernel_ms_isolated'], r['kernel_ms_isolated_min']), 'prep %.4f' % r['slab_prep_ms'], 'frac %.3f' % r['frac'])"
done
