# usage: bash profiles/run_sweep.sh <tag> [batches...]   (run on the GPU box from the repo root)
tag=$1; shift
Bs=${@:-65536 131072 262144 524288 1048576 2097152}
mkdir -p gpurun_out/sweep
for B in $Bs; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --batch-per-gpu $B --seq-ticks 140 --steps 1400 --predict-only-steps 500 > gpurun_out/sweep/${tag}_b$B.json 2> gpurun_out/sweep/${tag}_b$B.err
  python -c "import json; d=json.load(open('gpurun_out/sweep/${tag}_b$B.json')); print('$tag', $B, '%.3e'%d['value'], '%.1f us/step'%(d['ms_per_step']*1e3), 'predict GB/s %.0f'%d['roofline']['achieved'], 'mixed %.0f'%d['roofline']['mixed_achieved'])"
done
