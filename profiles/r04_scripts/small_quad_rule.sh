#!/bin/bash
# the kernel-family rule of the small fp32 batches, end to end on the cfg 3 schedule (a tag pose every 14th tick) and on cfg 2's cadence (every tick)
for B in 1024 4096; do for q in default 3 1 0; do for r in 1 2; do
if [ $q = default ]; then unset QLE_QUAD; else export QLE_QUAD=$q; fi
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --batch-per-gpu $B --seq-ticks 140 --steps 2800 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cfg3 schedule B=$B fp32 QLE_QUAD=$q: ticks/s %.4g'%d['value'], 'us/step %.3f'%(d['ms_per_step']*1e3), r['kernel'], 'us %.3f'%r['avg_launch_us'])"
done; done; done
unset QLE_QUAD
for q in default 3 1 0; do
if [ $q = default ]; then unset QLE_QUAD; else export QLE_QUAD=$q; fi
timeout -k 10 200 python bench.py --no-cpu-baseline --workload cfg2 --dtype f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cfg2 cadence B=4096 fp32 QLE_QUAD=$q: ticks/s %.4g'%d['value'], 'us/step %.3f'%(d['ms_per_step']*1e3), r['kernel'], 'us %.3f'%r['avg_launch_us'])"
done
