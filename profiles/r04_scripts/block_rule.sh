for B in 65536 131072; do for b in 256 128 64; do for r in 1 2; do
QLE_BLOCK=$b timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --batch-per-gpu $B --seq-ticks 140 --steps 2800 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('B=$B fp32 QLE_BLOCK=$b: ticks/s %.4g'%d['value'], 'us/step %.3f'%(d['ms_per_step']*1e3), r['kernel'], 'us %.3f'%r['avg_launch_us'])"
done; done; done
