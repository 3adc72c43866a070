set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
rocprofv3 -L > gpurun_out/r2/counters_list.txt 2>&1 || true
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rm -rf gpurun_out/r2/pmc5_$tag
  QLE_QUAD=0 QLE_ROWS_MAX=0 QLE_TIME_N=40 rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/r2/pmc5_$tag -o pmc -- python3 profiles/time_kernels.py 65536 f32 > gpurun_out/r2/pmc5_$tag.log 2>&1 || echo "set failed: $set"
  python3 profiles/summarize.py gpurun_out/r2/pmc5_$tag gpurun_out/r2/pmc5_$tag.md "$set" || true
done
