#!/bin/bash
# One tick launched in chunks of QLE_CHUNK filters (each chunk under the cache policy of a chunk-sized state) against the whole batch at once.
# usage: chunk_sweep.sh OUTDIR
O=$1; mkdir -p $O
for B in 131072 262144 524288; do
  for C in 0 65536 131072; do
    [ $C -ge $B ] && continue
    tag=b${B}_c${C}
    QLE_CHUNK=$C timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --batch-per-gpu $B --seq-ticks 140 --steps 1400 --kernel-steps 500 > $O/$tag.json 2> $O/$tag.err
    python3 -c "
import json; d=json.load(open('$O/$tag.json')); r=d['roofline']
print('B=%d chunk=%d: %.3e ticks/s  %.2f us/step  predict tick %.2f us  %.0f GB/s (%.3f of 8 TB/s)  policy %s' % ($B, $C, d['value'], d['ms_per_step']*1e3, r['avg_launch_us'], r['achieved'], r['frac'], r['state_policy']))" | tee -a $O/summary.txt
  done
done
