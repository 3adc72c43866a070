#!/bin/bash
# Round-3 evidence, regenerated in one go on the GPU box from the committed head:
#     gpurun --timeout 1150 -- "bash profiles/r03_profile.sh $(git rev-parse --short HEAD)"
# FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, --kernel-trace only) at 65 536 filters, at 2 097 152 filters, for cfg 2 and for the
# multirate predict tick (-> traffic.json, read by the bench lines that follow), bench lines of every workload, rocprofv3 kernel stats
# of the headline command, of the HBM-resident batch, of cfg 2, of the multirate workload and of fp64, the batch sweep, the per-kernel
# launch times, the per-wave timelines of the diagnostic build, the pk_fma issue microbenchmark, the ticks after a multirate correction,
# compact records against full records, the accuracy table.
# Raw CSVs stay under gpurun_out/ (scratch); the summaries go to profiles/ via gpurun_out/r3/profiles_out/ (copied back by hand).
export QLE_HEAD_SHA=${1:-unknown}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/profiles_out
R=gpurun_out/r3/prof
rm -rf $O $R; mkdir -p $O $R
py=python3
step() { echo "== $*"; }

step HBM traffic counters
pmc() {  # tag counter args...
  tag=$1; ctr=$2; shift 2
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $R/pmc_$tag -o p -- $py bench.py --no-cpu-baseline --no-extras "$@" > $R/pmc_$tag.log 2>&1
  $py profiles/summarize.py $R/pmc_$tag $O/r03_pmc_$tag.md "--pmc $ctr -- bench.py $*"
}
pmc fetch_b65536 FETCH_SIZE --steps 280 --kernel-steps 200
pmc write_b65536 WRITE_SIZE --steps 280 --kernel-steps 200
$py profiles/summarize.py --traffic $R/pmc_fetch_b65536 $R/pmc_write_b65536 cfg3:65536:f32:predict 'k_predict<float, false, 2, false,' $O/traffic.json
$py profiles/summarize.py --traffic $R/pmc_fetch_b65536 $R/pmc_write_b65536 cfg3:65536:f32:step 'k_step<float' $O/traffic.json
pmc fetch_b2097152 FETCH_SIZE --batch-per-gpu 2097152 --steps 56 --warmup 14 --kernel-steps 40
pmc write_b2097152 WRITE_SIZE --batch-per-gpu 2097152 --steps 56 --warmup 14 --kernel-steps 40
$py profiles/summarize.py --traffic $R/pmc_fetch_b2097152 $R/pmc_write_b2097152 cfg3:2097152:f32:predict 'k_predict<float, false, 3, false,' $O/traffic.json
pmc fetch_cfg2 FETCH_SIZE --workload cfg2 --steps 200 --kernel-steps 200
pmc write_cfg2 WRITE_SIZE --workload cfg2 --steps 200 --kernel-steps 200
$py profiles/summarize.py --traffic $R/pmc_fetch_cfg2 $R/pmc_write_cfg2 cfg2:4096:f64:step 'kw_tick<double' $O/traffic.json
pmc fetch_mr FETCH_SIZE --workload cfg3mr --steps 280 --kernel-steps 200
pmc write_mr WRITE_SIZE --workload cfg3mr --steps 280 --kernel-steps 200
$py profiles/summarize.py --traffic $R/pmc_fetch_mr $R/pmc_write_mr cfg3mr:65536:f32:predict 'k_predict<float, false, 2, true,' $O/traffic.json
$py profiles/summarize.py --traffic $R/pmc_fetch_mr $R/pmc_write_mr cfg3mr:65536:f32:step 'k_step_mr<float' $O/traffic.json
cp $O/traffic.json profiles/traffic.json   # the bench lines below read it (roofline.traffic)

step bench lines
timeout -k 10 600 $py bench.py > $O/r03_bench.json 2> $R/bench.err || tail -5 $R/bench.err
timeout -k 10 300 $py bench.py --steps 20 --warmup 5 > $O/r03_bench_driver_args.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg2 > $O/r03_bench_cfg2.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg4 --steps 1400 > $O/r03_bench_cfg4_1gpu.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg5 --steps 1400 > $O/r03_bench_cfg5_1gpu.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg5 --batch-per-gpu 32768 --steps 1400 > $O/r03_bench_cfg5_shard.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg4 --batch-per-gpu 131072 --steps 1400 > $O/r03_bench_cfg4_shard.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg3mr --steps 1400 > $O/r03_bench_multirate.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --steps 20000 --warmup 6000 --no-cpu-baseline --no-extras > $O/r03_bench_long.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg3mr --dtype f64 --steps 560 --no-cpu-baseline --no-extras > $O/r03_bench_multirate_f64.json 2>> $R/bench.err

step kernel stats
stats() {  # tag label args...
  tag=$1; label=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_$tag -o s -- $py bench.py "$@" > $R/stats_$tag.log 2>&1
  $py profiles/summarize.py $R/stats_$tag $O/r03_kernel_stats$( [ "$tag" = main ] || echo _$tag ).md "$label"
}
stats main "bench.py --no-cpu-baseline --no-extras --steps 1400 (cfg3, 65 536 fp32 filters)" --no-cpu-baseline --no-extras --steps 1400
stats b2097152 "bench.py --batch-per-gpu 2097152 --no-cpu-baseline --no-extras --steps 280 --kernel-steps 200 (HBM-resident batch, 1.15 GB of state)" --batch-per-gpu 2097152 --no-cpu-baseline --no-extras --steps 280 --warmup 14 --kernel-steps 200
stats cfg2 "bench.py --workload cfg2 (4 096 fp64 filters, update on every tick)" --workload cfg2 --no-cpu-baseline
stats multirate "bench.py --workload cfg3mr --steps 1400" --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400
stats f64 "bench.py --dtype f64 --steps 1400 (cfg3 schedule, 65 536 fp64 filters)" --dtype f64 --no-cpu-baseline --no-extras --steps 1400
stats multirate_f64 "bench.py --workload cfg3mr --dtype f64 --steps 560" --workload cfg3mr --dtype f64 --no-cpu-baseline --no-extras --steps 560

step SQ counters of the multirate correcting tick and of the predict tick
pmc sq_mr "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" --workload cfg3mr --steps 280 --kernel-steps 200
pmc sq_b65536 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" --steps 280 --kernel-steps 200

step batch sweep
echo "# batch sweep, fp32, cfg3 schedule (bench.py --no-cpu-baseline --no-extras --batch-per-gpu B --seq-ticks 140 --steps 1400 --kernel-steps 500), head $QLE_HEAD_SHA" > $O/r03_sweep.md
echo "" >> $O/r03_sweep.md
echo "| filters | state MiB | served by | ticks/s | us/step | k_predict us | GB/s | frac of 8 TB/s | frac of 6.29 TB/s copy | mixed GB/s |" >> $O/r03_sweep.md
echo "|---|---|---|---|---|---|---|---|---|---|" >> $O/r03_sweep.md
for B in 4096 16384 32768 65536 131072 262144 524288 1048576 2097152 4194304; do
  timeout -k 10 200 $py bench.py --no-cpu-baseline --no-extras --batch-per-gpu $B --seq-ticks 140 --steps 1400 --kernel-steps 500 > $R/sweep_$B.json 2> $R/sweep_$B.err
  $py -c "
import json; d=json.load(open('$R/sweep_$B.json')); r=d['roofline']
print('| %d | %.0f | %s | %.3e | %.2f | %.2f | %.0f | %.3f | %.3f | %.0f |' % ($B, $B*576/2**20, r['served_by'], d['value'], d['ms_per_step']*1e3, r['avg_launch_us'], r['achieved'], r['frac'], r['frac_of_measured_copy'], r['mixed_achieved']))" >> $O/r03_sweep.md
done
cat $O/r03_sweep.md

step per-kernel launch times
rm -f $O/r03_kernel_times.jsonl
for spec in "16384 f32" "65536 f32" "131072 f32" "262144 f32" "1048576 f32" "65536 f64" "4096 f64"; do
  set -- $spec
  QLE_QUAD=0 timeout -k 10 120 $py profiles/time_kernels.py $1 $2 lanes >> $O/r03_kernel_times.jsonl 2>> $R/times.err
done
for spec in "1024 f64" "4096 f64" "4096 f32" "16384 f64" "16384 f32"; do
  set -- $spec
  timeout -k 10 120 $py profiles/time_kernels.py $1 $2 default >> $O/r03_kernel_times.jsonl 2>> $R/times.err
done
QLE_QUAD=0 QLE_TIME_DIRECT=0 timeout -k 10 120 $py profiles/time_kernels.py 65536 f32 lanes-conventional >> $O/r03_kernel_times.jsonl 2>> $R/times.err
QLE_QUAD=0 QLE_TIME_DIRECT=0 timeout -k 10 120 $py profiles/time_kernels.py 65536 f64 lanes-conventional >> $O/r03_kernel_times.jsonl 2>> $R/times.err
cat $O/r03_kernel_times.jsonl | cut -c1-200

step "per-wave timelines, diagnostic build, and the pk_fma / store-path microbenchmarks"
for mb in pk_issue store_path; do /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o profiles/micro/$mb profiles/micro/$mb.hip; done
QLE_LIB=$PWD/quadrotor_landing_amd/libqle_dbg.so timeout -k 10 200 $py profiles/r03_scripts/mr_timeline.py 65536 12 > $O/r03_mr_timeline.log 2>&1
QLE_LIB=$PWD/quadrotor_landing_amd/libqle_dbg.so timeout -k 10 200 $py profiles/r03_scripts/kw_timeline.py 4096 f64 > $O/r03_kw_timeline.log 2>&1
timeout -k 10 120 profiles/micro/pk_issue > $O/r03_pk_issue.log 2>&1
timeout -k 10 60 profiles/micro/store_path > $O/r03_store_path.log 2>&1
tail -3 $O/r03_mr_timeline.log $O/r03_kw_timeline.log

step "what the ticks after a multirate correction cost (durations by distance from the correcting tick, from the kernel trace above)"
$py profiles/r03_scripts/after_step.py $R/stats_multirate k_step_mr > $O/r03_after_step.md 2>&1; cat $O/r03_after_step.md

step "compact records (est_bias = false): launch times against full records, FETCH_SIZE / WRITE_SIZE of the compact ticks"
bash profiles/r03_scripts/compact.sh $QLE_HEAD_SHA compact_final > $R/compact.log 2>&1
cp gpurun_out/r3/compact_final/times.jsonl $O/r03_compact_times.jsonl
cp gpurun_out/r3/compact_final/traffic_compact.json $O/r03_compact_traffic.json
cut -c1-220 $O/r03_compact_times.jsonl

step "small batches: workgroup size of the lane kernels, kernel family by batch size"
bash profiles/r03_scripts/small_block.sh > $O/r03_small_block.log 2>&1
bash profiles/r03_scripts/small_family.sh > $O/r03_small_family.log 2>&1
tail -4 $O/r03_small_family.log

step accuracy
timeout -k 10 300 $py profiles/measure_accuracy.py > $O/r03_accuracy.md 2> $R/acc.err || tail -5 $R/acc.err
ls -la $O
