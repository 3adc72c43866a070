set -e
mkdir -p gpurun_out/r4/passes
out=gpurun_out/r4/passes/sweep.txt
: > $out
for cfg in "65536 1" "65536 2" "131072 1" "131072 2" "262144 1" "262144 2" "262144 4" "524288 1" "524288 2" "524288 4" "524288 8" "1048576 1" "1048576 4"; do
  set -- $cfg
  QLE_PASSES=$2 timeout -k 10 300 python bench.py --batch-per-gpu $1 --no-cpu-baseline --no-extras --seq-ticks 140 --steps 1400 --kernel-steps 500 > gpurun_out/r4/passes/b$1_p$2.json 2> gpurun_out/r4/passes/b$1_p$2.err
  python - $1 $2 >> $out <<PY
import json,sys
d=json.loads(open('gpurun_out/r4/passes/b%s_p%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[1],sys.argv[2],'ticks/s %.4g'%d['value'],'k_predict us %.2f'%r['avg_launch_us'],'frac %.3f'%r['frac'],'nonfinite',d.get('nonfinite_filters'))
PY
  tail -1 $out
done
