set -e
mkdir -p gpurun_out/r2
L=gpurun_out/r2/base.log
: > $L
for a in "65536 f32" "262144 f32" "1048576 f32" "65536 f64" "4096 f64" "4096 f32"; do
  timeout -k 10 200 python profiles/time_kernels.py $a >> $L 2>&1
done
for nt in auto 0 1 2; do
  if [ $nt = auto ]; then unset QLE_NT; else export QLE_NT=$nt; fi
  echo "cfg3mr NT=$nt" >> $L
  timeout -k 10 200 python bench.py --workload cfg3mr --no-cpu-baseline --steps 1400 --predict-only-steps 500 >> $L 2>gpurun_out/r2/base_mr_$nt.err
done
unset QLE_NT
echo "cfg2" >> $L
timeout -k 10 200 python bench.py --workload cfg2 --no-cpu-baseline >> $L 2>gpurun_out/r2/base_cfg2.err
echo done >> $L
