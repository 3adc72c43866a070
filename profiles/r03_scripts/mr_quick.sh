#!/bin/bash
# kernel stats of the multirate workload only (one rocprofv3 run) + the ticks after the correcting tick
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/${1:-mrq}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mr -o s -- python3 bench.py --workload cfg3mr ${2:-} --no-cpu-baseline --no-extras --steps 1400 > $O/bench.json 2> $O/bench.err
python3 profiles/summarize.py $O/stats_mr $O/kernel_stats_multirate.md "bench.py --workload cfg3mr --steps 1400" > /dev/null
sed -n 7,12p $O/kernel_stats_multirate.md | cut -c1-160 | grep -v synth
python3 profiles/r03_scripts/after_step.py $O/stats_mr k_step_mr | sed -n 5,9p
