set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for q in 0 3; do
  rm -rf gpurun_out/r2/pmc_q$q
  QLE_QUAD=$q QLE_ROWS_MAX=0 QLE_TIME_N=40 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d gpurun_out/r2/pmc_q$q -o pmc -- python3 profiles/time_kernels.py 65536 f32 > gpurun_out/r2/pmc_q$q.log 2>&1
  python3 profiles/summarize.py gpurun_out/r2/pmc_q$q gpurun_out/r2/pmc_q$q.md "QLE_QUAD=$q 65536 f32"
done
