#!/bin/bash
# Round-4 evidence, regenerated in one go on the GPU box from the committed head:
#     gpurun --timeout 1150 -- "bash profiles/r04_profile.sh $(git rev-parse --short HEAD)"
# (round 4 adds: the shipped-cadence workloads rotors / hardware, fp64 multirate timeline, DRAM-destination counters at the cached and the
# HBM-resident size, L2 hit / fabric counters at 65 536 against 131 072 filters and the SQ picture of a 32 768-filter shard, chunked launches)
# FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, --kernel-trace only) at 65 536 filters, at 2 097 152 filters, for cfg 2 and for the
# multirate predict tick (-> traffic.json, read by the bench lines that follow), bench lines of every workload, rocprofv3 kernel stats
# of the headline command, of the HBM-resident batch, of cfg 2, of the multirate workload and of fp64, the batch sweep, the per-kernel
# launch times, the per-wave timelines of the diagnostic build, the pk_fma issue microbenchmark, the ticks after a multirate correction,
# compact records against full records, the accuracy table.
# Raw CSVs stay under gpurun_out/ (scratch); the summaries go to profiles/ via gpurun_out/r4/profiles_out/ (copied back by hand).
export QLE_HEAD_SHA=${1:-unknown}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/profiles_out
R=gpurun_out/r4/prof
rm -rf $O $R; mkdir -p $O $R
py=python3
step() { echo "== $*"; }

step HBM traffic counters
pmc() {  # tag counter args...
  tag=$1; ctr=$2; shift 2
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $R/pmc_$tag -o p -- $py bench.py --no-cpu-baseline --no-extras "$@" > $R/pmc_$tag.log 2>&1
  $py profiles/summarize.py $R/pmc_$tag $O/r04_pmc_$tag.md "--pmc $ctr -- bench.py $*"
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
pmc fetch_mr_f64 FETCH_SIZE --workload cfg3mr --dtype f64 --steps 280 --kernel-steps 100
pmc write_mr_f64 WRITE_SIZE --workload cfg3mr --dtype f64 --steps 280 --kernel-steps 100
$py profiles/summarize.py --traffic $R/pmc_fetch_mr_f64 $R/pmc_write_mr_f64 cfg3mr:65536:f64:step 'k_step_mr<double' $O/traffic.json
pmc fetch_hw FETCH_SIZE --workload hardware --steps 140 --kernel-steps 100
pmc write_hw WRITE_SIZE --workload hardware --steps 140 --kernel-steps 100
$py profiles/summarize.py --traffic $R/pmc_fetch_hw $R/pmc_write_hw hardware:65536:f32:step 'k_step_mr<float' $O/traffic.json
cp $O/traffic.json profiles/traffic.json   # the bench lines below read it (roofline.traffic)

step "can a counter separate Infinity-Cache hits from HBM?  requests by destination at the cached size and at the HBM-resident size"
pmc dram_b65536 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" --steps 280 --kernel-steps 200
pmc dram_b2097152 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" --batch-per-gpu 2097152 --steps 56 --warmup 14 --kernel-steps 40

step "the plateau: L2 hit rate and fabric request levels at 65 536 against 131 072 filters; the SQ picture of a 32 768-filter shard"
for B in 65536 131072; do
  pmc tcc_b$B "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" --batch-per-gpu $B --steps 280 --kernel-steps 200
  pmc tccl_b$B "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum" --batch-per-gpu $B --steps 280 --kernel-steps 200
done
for B in 32768 65536; do
  pmc sq_b$B "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" --batch-per-gpu $B --steps 280 --kernel-steps 200
done

step bench lines
timeout -k 10 600 $py bench.py > $O/r04_bench.json 2> $R/bench.err || tail -5 $R/bench.err
timeout -k 10 300 $py bench.py --steps 20 --warmup 5 > $O/r04_bench_driver_args.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg2 > $O/r04_bench_cfg2.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg4 --steps 1400 > $O/r04_bench_cfg4_1gpu.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg5 --steps 1400 > $O/r04_bench_cfg5_1gpu.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg5 --batch-per-gpu 32768 --steps 1400 > $O/r04_bench_cfg5_shard.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg4 --batch-per-gpu 131072 --steps 1400 > $O/r04_bench_cfg4_shard.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg3mr --steps 1400 > $O/r04_bench_multirate.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --steps 20000 --warmup 6000 --no-cpu-baseline --no-extras > $O/r04_bench_long.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg3mr --dtype f64 --steps 560 --no-cpu-baseline --no-extras > $O/r04_bench_multirate_f64.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload rotors --no-cpu-baseline --no-extras > $O/r04_bench_rotors.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload rotors --dtype f64 --no-cpu-baseline --no-extras > $O/r04_bench_rotors_f64.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload hardware --no-cpu-baseline --no-extras > $O/r04_bench_hardware.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload hardware --dtype f64 --no-cpu-baseline --no-extras > $O/r04_bench_hardware_f64.json 2>> $R/bench.err

step "soak runs on the final kernels (long timed regions: no non-finite filter, the short runs' rates)"
timeout -k 10 300 $py bench.py --steps 504000 --warmup 0 --no-cpu-baseline --no-extras > $O/r04_bench_soak.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg3mr --steps 140000 --warmup 0 --no-cpu-baseline --no-extras > $O/r04_bench_soak_multirate.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg3mr --dtype f64 --steps 140000 --warmup 0 --no-cpu-baseline --no-extras > $O/r04_bench_soak_multirate_f64.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload hardware --steps 70000 --warmup 0 --no-cpu-baseline --no-extras > $O/r04_bench_soak_hardware.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload hardware --dtype f64 --steps 42000 --warmup 0 --no-cpu-baseline --no-extras > $O/r04_bench_soak_hardware_f64.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload rotors --steps 280000 --warmup 0 --no-cpu-baseline --no-extras > $O/r04_bench_soak_rotors.json 2>> $R/bench.err
for f in $O/r04_bench_soak*.json; do $py -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], '%.3e ticks/s' % d['value'], 'steps', d['steps'], 'nonfinite', d['nonfinite_filters'], 'rmse', d.get('rmse_vs_truth',{}).get('position_m'))"; done

step kernel stats
stats() {  # tag label args...
  tag=$1; label=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_$tag -o s -- $py bench.py "$@" > $R/stats_$tag.log 2>&1
  $py profiles/summarize.py $R/stats_$tag $O/r04_kernel_stats$( [ "$tag" = main ] || echo _$tag ).md "$label"
}
stats main "bench.py --no-cpu-baseline --no-extras --steps 1400 (cfg3, 65 536 fp32 filters)" --no-cpu-baseline --no-extras --steps 1400
stats b2097152 "bench.py --batch-per-gpu 2097152 --no-cpu-baseline --no-extras --steps 280 --kernel-steps 200 (HBM-resident batch, 1.15 GB of state)" --batch-per-gpu 2097152 --no-cpu-baseline --no-extras --steps 280 --warmup 14 --kernel-steps 200
stats cfg2 "bench.py --workload cfg2 (4 096 fp64 filters, update on every tick)" --workload cfg2 --no-cpu-baseline
stats multirate "bench.py --workload cfg3mr --steps 1400" --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400
stats f64 "bench.py --dtype f64 --steps 1400 (cfg3 schedule, 65 536 fp64 filters)" --dtype f64 --no-cpu-baseline --no-extras --steps 1400
stats multirate_f64 "bench.py --workload cfg3mr --dtype f64 --steps 560" --workload cfg3mr --dtype f64 --no-cpu-baseline --no-extras --steps 560
stats rotors "bench.py --workload rotors (relative_pose_EKF_rotors.yaml as shipped, 65 536 fp32 filters)" --workload rotors --no-cpu-baseline --no-extras
stats hardware "bench.py --workload hardware (relative_pose_EKF_hardware.yaml as shipped, 65 536 fp32 filters)" --workload hardware --no-cpu-baseline --no-extras
stats hardware_f64 "bench.py --workload hardware --dtype f64 (65 536 fp64 filters)" --workload hardware --dtype f64 --no-cpu-baseline --no-extras --steps 700

step SQ counters of the multirate correcting tick and of the predict tick
pmc sq_mr "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" --workload cfg3mr --steps 280 --kernel-steps 200

step batch sweep
echo "# batch sweep, fp32, cfg3 schedule (bench.py --no-cpu-baseline --no-extras --batch-per-gpu B --seq-ticks 140 --steps 1400 --kernel-steps 500), head $QLE_HEAD_SHA" > $O/r04_sweep.md
echo "" >> $O/r04_sweep.md
echo "| filters | state MiB | served by | ticks/s | us/step | k_predict us | GB/s | frac of 8 TB/s | frac of 6.29 TB/s copy | mixed GB/s |" >> $O/r04_sweep.md
echo "|---|---|---|---|---|---|---|---|---|---|" >> $O/r04_sweep.md
for B in 4096 16384 32768 65536 131072 262144 524288 1048576 2097152 4194304; do
  timeout -k 10 200 $py bench.py --no-cpu-baseline --no-extras --batch-per-gpu $B --seq-ticks 140 --steps 1400 --kernel-steps 500 > $R/sweep_$B.json 2> $R/sweep_$B.err
  $py -c "
import json; d=json.load(open('$R/sweep_$B.json')); r=d['roofline']
print('| %d | %.0f | %s | %.3e | %.2f | %.2f | %.0f | %.3f | %.3f | %.0f |' % ($B, $B*576/2**20, r['served_by'], d['value'], d['ms_per_step']*1e3, r['avg_launch_us'], r['achieved'], r['frac'], r['frac_of_measured_copy'], r['mixed_achieved']))" >> $O/r04_sweep.md
done
cat $O/r04_sweep.md

step per-kernel launch times
rm -f $O/r04_kernel_times.jsonl
for spec in "16384 f32" "65536 f32" "131072 f32" "262144 f32" "1048576 f32" "65536 f64" "4096 f64"; do
  set -- $spec
  QLE_QUAD=0 timeout -k 10 120 $py profiles/time_kernels.py $1 $2 lanes >> $O/r04_kernel_times.jsonl 2>> $R/times.err
done
for spec in "1024 f64" "4096 f64" "4096 f32" "16384 f64" "16384 f32"; do
  set -- $spec
  timeout -k 10 120 $py profiles/time_kernels.py $1 $2 default >> $O/r04_kernel_times.jsonl 2>> $R/times.err
done
QLE_QUAD=0 QLE_TIME_DIRECT=0 timeout -k 10 120 $py profiles/time_kernels.py 65536 f32 lanes-conventional >> $O/r04_kernel_times.jsonl 2>> $R/times.err
QLE_QUAD=0 QLE_TIME_DIRECT=0 timeout -k 10 120 $py profiles/time_kernels.py 65536 f64 lanes-conventional >> $O/r04_kernel_times.jsonl 2>> $R/times.err
cat $O/r04_kernel_times.jsonl | cut -c1-200

step "per-wave timelines, diagnostic build, and the pk_fma / store-path microbenchmarks"
for mb in pk_issue store_path wave_load_split; do /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o profiles/micro/$mb profiles/micro/$mb.hip; done
QLE_LIB=$PWD/quadrotor_landing_amd/libqle_dbg.so timeout -k 10 200 $py profiles/r03_scripts/mr_timeline.py 65536 12 > $O/r04_mr_timeline.log 2>&1
QLE_LIB=$PWD/quadrotor_landing_amd/libqle_dbg.so timeout -k 10 200 $py profiles/r03_scripts/mr_timeline.py 65536 12 f64 > $O/r04_mr_timeline_f64.log 2>&1
QLE_LIB=$PWD/quadrotor_landing_amd/libqle_dbg.so timeout -k 10 200 $py profiles/r03_scripts/kw_timeline.py 4096 f64 > $O/r04_kw_timeline.log 2>&1
timeout -k 10 120 profiles/micro/pk_issue > $O/r04_pk_issue.log 2>&1
timeout -k 10 60 profiles/micro/store_path > $O/r04_store_path.log 2>&1
timeout -k 10 60 profiles/micro/wave_load_split > $O/r04_wave_load_split.log 2>&1
tail -3 $O/r04_mr_timeline.log $O/r04_kw_timeline.log

step "what the ticks after a multirate correction cost (durations by distance from the correcting tick, from the kernel trace above)"
$py profiles/r03_scripts/after_step.py $R/stats_multirate k_step_mr > $O/r04_after_step.md 2>&1; cat $O/r04_after_step.md

step "compact records (est_bias = false): launch times against full records, FETCH_SIZE / WRITE_SIZE of the compact ticks"
bash profiles/r03_scripts/compact.sh $QLE_HEAD_SHA compact_final > $R/compact.log 2>&1
cp gpurun_out/r3/compact_final/times.jsonl $O/r04_compact_times.jsonl
cp gpurun_out/r3/compact_final/traffic_compact.json $O/r04_compact_traffic.json
cut -c1-220 $O/r04_compact_times.jsonl

step "small batches: workgroup size of the lane kernels, kernel family by batch size"
bash profiles/r03_scripts/small_block.sh > $O/r04_small_block.log 2>&1
bash profiles/r03_scripts/small_family.sh > $O/r04_small_family.log 2>&1
tail -4 $O/r04_small_family.log

step "one tick launched in chunks against the whole batch at once"
bash profiles/r04_scripts/chunk_sweep.sh $R/chunk > /dev/null 2>&1; cp $R/chunk/summary.txt $O/r04_chunk_sweep.txt; cat $O/r04_chunk_sweep.txt

step accuracy
timeout -k 10 300 $py profiles/measure_accuracy.py > $O/r04_accuracy.md 2> $R/acc.err || tail -5 $R/acc.err
ls -la $O
