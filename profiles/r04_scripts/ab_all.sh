set -e
mkdir -p gpurun_out/r4/ab
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/ab/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -2 gpurun_out/r4/ab/tests.log; [ $rc -eq 0 ]
A=quadrotor_landing_amd/libqle_base.so; B=quadrotor_landing_amd/libqle_ekf.so
for w in "--workload cfg3mr" "--workload cfg2" "--workload hardware" "--workload rotors" "--dtype f64" "--workload cfg3mr --dtype f64" "--batch-per-gpu 131072" "--batch-per-gpu 32768" "--batch-per-gpu 2097152 --steps 280"; do
  echo "== $w"; bash profiles/r04_scripts/ab_lib.sh $A $B 2 $w
done 2>&1 | tee gpurun_out/r4/ab/ab_all.log
