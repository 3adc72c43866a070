# valu_issue3 compiled with and without the SLP vectoriser: is the short-loop rate of section 3 packed math? (r02_tuning.md section 10)
mkdir -p gpurun_out/s2 && cd profiles/micro
for fl in "" "-fno-slp-vectorize"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $fl -o /tmp/vi3 valu_issue3.hip 2>/dev/null && echo "== flags: '$fl'" && timeout -k 10 120 /tmp/vi3
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/s2/issue_rate.log
