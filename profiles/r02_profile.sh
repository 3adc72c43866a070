#!/bin/bash
# Round-2 evidence, regenerated in one go on the GPU box from the committed head:
#     gpurun --timeout 1150 -- "bash profiles/r02_profile.sh $(git rev-parse --short HEAD)"
# FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, --kernel-trace only) at 65 536 filters, at 2 097 152 filters and for cfg 2
# (-> traffic.json, read by the bench lines that follow), bench lines of every workload, rocprofv3 kernel stats of the headline
# command, SQ / TCC passes at 65 536 and 262 144 filters, the batch sweep, the accuracy table.
# Raw CSVs stay under gpurun_out/ (scratch); the summaries go to profiles/ via gpurun_out/r2/profiles_out/ (copied back by hand).
export QLE_HEAD_SHA=${1:-unknown}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2/profiles_out
R=gpurun_out/r2/prof
rm -rf $O $R; mkdir -p $O $R
py=python3
step() { echo "== $*"; }

step HBM traffic counters
pmc() {  # tag counter args...
  tag=$1; ctr=$2; shift 2
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $R/pmc_$tag -o p -- $py bench.py --no-cpu-baseline --no-extras "$@" > $R/pmc_$tag.log 2>&1
  $py profiles/summarize.py $R/pmc_$tag $O/r02_pmc_$tag.md "--pmc $ctr -- bench.py $*"
}
pmc fetch_b65536 FETCH_SIZE --steps 280 --kernel-steps 200
pmc write_b65536 WRITE_SIZE --steps 280 --kernel-steps 200
$py profiles/summarize.py --traffic $R/pmc_fetch_b65536 $R/pmc_write_b65536 cfg3:65536:f32:predict 'k_predict<float, false, 2, false>' $O/traffic.json
$py profiles/summarize.py --traffic $R/pmc_fetch_b65536 $R/pmc_write_b65536 cfg3:65536:f32:step 'k_step<float' $O/traffic.json
pmc fetch_b2097152 FETCH_SIZE --batch-per-gpu 2097152 --steps 56 --warmup 14 --kernel-steps 40
pmc write_b2097152 WRITE_SIZE --batch-per-gpu 2097152 --steps 56 --warmup 14 --kernel-steps 40
$py profiles/summarize.py --traffic $R/pmc_fetch_b2097152 $R/pmc_write_b2097152 cfg3:2097152:f32:predict 'k_predict<float, false, 3, false>' $O/traffic.json
pmc fetch_cfg2 FETCH_SIZE --workload cfg2 --steps 200 --kernel-steps 200
pmc write_cfg2 WRITE_SIZE --workload cfg2 --steps 200 --kernel-steps 200
$py profiles/summarize.py --traffic $R/pmc_fetch_cfg2 $R/pmc_write_cfg2 cfg2:4096:f64:step 'kw_tick<double' $O/traffic.json

cp $O/traffic.json profiles/traffic.json   # the bench lines below read it (roofline.traffic)

step bench lines
timeout -k 10 600 $py bench.py > $O/r02_bench.json 2> $R/bench.err || tail -5 $R/bench.err
timeout -k 10 300 $py bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/r02_bench_driver_args.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg2 > $O/r02_bench_cfg2.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg4 --steps 1400 > $O/r02_bench_cfg4_1gpu.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg5 --steps 1400 > $O/r02_bench_cfg5_1gpu.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg5 --batch-per-gpu 32768 --steps 1400 > $O/r02_bench_cfg5_shard.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --workload cfg3mr --steps 1400 > $O/r02_bench_multirate.json 2>> $R/bench.err
timeout -k 10 300 $py bench.py --steps 20000 --warmup 6000 --no-cpu-baseline --no-extras > $O/r02_bench_long.json 2>> $R/bench.err

step kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -o s -- $py bench.py --no-cpu-baseline --no-extras --steps 1400 > $R/stats.log 2>&1
$py profiles/summarize.py $R/stats $O/r02_kernel_stats.md "bench.py --no-cpu-baseline --no-extras --steps 1400 (cfg3, 65 536 fp32 filters)"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_cfg2 -o s -- $py bench.py --workload cfg2 --no-cpu-baseline > $R/stats_cfg2.log 2>&1
$py profiles/summarize.py $R/stats_cfg2 $O/r02_kernel_stats_cfg2.md "bench.py --workload cfg2 (4 096 fp64 filters, update on every tick)"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_mr -o s -- $py bench.py --workload cfg3mr --no-cpu-baseline --steps 1400 > $R/stats_mr.log 2>&1
$py profiles/summarize.py $R/stats_mr $O/r02_kernel_stats_multirate.md "bench.py --workload cfg3mr --steps 1400"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_f64 -o s -- $py bench.py --dtype f64 --no-cpu-baseline --no-extras --steps 1400 > $R/stats_f64.log 2>&1
$py profiles/summarize.py $R/stats_f64 $O/r02_kernel_stats_f64.md "bench.py --dtype f64 --steps 1400 (cfg3 schedule, 65 536 fp64 filters)"

step SQ and TCC counters, 65536 vs 262144 filters
for B in 65536 262144; do
  pmc sq_b$B "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" --batch-per-gpu $B --steps 280 --kernel-steps 200
  pmc tcc_b$B "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" --batch-per-gpu $B --steps 280 --kernel-steps 200
  pmc lvl_b$B "SQ_LEVEL_WAVES SQ_WAVES GRBM_GUI_ACTIVE" --batch-per-gpu $B --steps 280 --kernel-steps 200
done

step batch sweep
echo "# batch sweep, fp32, cfg3 schedule (bench.py --no-cpu-baseline --no-extras --batch-per-gpu B --seq-ticks 140 --steps 1400 --kernel-steps 500), head $QLE_HEAD_SHA" > $O/r02_sweep.md
echo "" >> $O/r02_sweep.md
echo "| filters | state MiB | served by | ticks/s | us/step | k_predict us | GB/s | frac of 8 TB/s | frac of 6.29 TB/s copy | mixed GB/s |" >> $O/r02_sweep.md
echo "|---|---|---|---|---|---|---|---|---|---|" >> $O/r02_sweep.md
for B in 16384 32768 65536 131072 262144 524288 1048576 2097152 4194304; do
  timeout -k 10 200 $py bench.py --no-cpu-baseline --no-extras --batch-per-gpu $B --seq-ticks 140 --steps 1400 --kernel-steps 500 > $R/sweep_$B.json 2> $R/sweep_$B.err
  $py -c "
import json; d=json.load(open('$R/sweep_$B.json')); r=d['roofline']
print('| %d | %.0f | %s | %.3e | %.2f | %.2f | %.0f | %.3f | %.3f | %.0f |' % ($B, $B*576/2**20, r['served_by'], d['value'], d['ms_per_step']*1e3, r['avg_launch_us'], r['achieved'], r['frac'], r['frac_of_measured_copy'], r['mixed_achieved']))" >> $O/r02_sweep.md
done
cat $O/r02_sweep.md

step accuracy
timeout -k 10 300 $py profiles/measure_accuracy.py > $O/r02_accuracy.md 2> $R/acc.err || tail -5 $R/acc.err
ls -la $O
