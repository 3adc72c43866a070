for st in 0 0 1 2 3 4 6; do
  QLE_STAGGER=$st timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --batch-per-gpu 131072 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('stagger $st x 8128 cycles: ticks/s %.4g'%d['value'], 'k_predict us %.3f'%r['avg_launch_us'])"
done
