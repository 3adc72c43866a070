#!/bin/bash
# A/B of two in-tree builds of the engine on one box, alternating: bash profiles/r04_scripts/ab_lib.sh <libA.so> <libB.so> <rounds> <bench args...>
A=$1; B=$2; N=$3; shift 3
for r in $(seq $N); do
  for lib in $A $B; do
    QLE_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 'ticks/s %.4g'%d['value'], r['kernel'], 'us %.3f'%r['avg_launch_us'], 'step us %.3f'%(d['ms_per_step']*1e3))"
  done
done
