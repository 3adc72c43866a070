#!/bin/bash
# est_bias = false: compact records (64 words per direction) against full records (136): launch times at 65 536 ... 1 048 576 filters, and
# the HBM/fabric byte counters (separate FETCH_SIZE / WRITE_SIZE passes) of the compact predict tick at 1 048 576 filters (HBM-served)
# and at 65 536 filters
export QLE_HEAD_SHA=${1:-unknown}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/${2:-compact}; mkdir -p $O
export QLE_TIME_EST_BIAS=0
: > $O/times.jsonl
QLE_COMPACT=0 timeout -k 10 200 python3 profiles/time_kernels.py 65536 f32 warmup > /dev/null 2>> $O/err.log   # first process on a fresh box: clocks still ramping
for B in 65536 262144 1048576; do for c in 0 1; do
  QLE_COMPACT=$c timeout -k 10 200 python3 profiles/time_kernels.py $B f32 compact$c >> $O/times.jsonl 2>> $O/err.log
done; done
QLE_COMPACT=0 timeout -k 10 200 python3 profiles/time_kernels.py 65536 f64 compact0 >> $O/times.jsonl 2>> $O/err.log
QLE_COMPACT=1 timeout -k 10 200 python3 profiles/time_kernels.py 65536 f64 compact1 >> $O/times.jsonl 2>> $O/err.log
cat $O/times.jsonl
export QLE_COMPACT=1 QLE_TIME_N=100
for B in 65536 1048576; do
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch_$B -o p -- python3 profiles/time_kernels.py $B f32 pmc > $O/pmc_fetch_$B.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write_$B -o p -- python3 profiles/time_kernels.py $B f32 pmc > $O/pmc_write_$B.log 2>&1
  python3 profiles/summarize.py --traffic $O/pmc_fetch_$B $O/pmc_write_$B nobias:$B:f32:predict 'k_predict<float, false, ., false, true>' $O/traffic_compact.json
  python3 profiles/summarize.py --traffic $O/pmc_fetch_$B $O/pmc_write_$B nobias:$B:f32:step 'k_step<float, true, false, false, ., true>' $O/traffic_compact.json
done
cat $O/traffic_compact.json
