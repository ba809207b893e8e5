#!/bin/bash
# the bench part of tools/final_profile.sh alone (bench line + per-kernel stats + step trace of the same command)
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/fb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fb -o b -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sequence --no-strong > $O/bench_prof.json 2> $O/bench_prof.err
cp $(find /tmp/fb -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/trace_step.py $(find /tmp/fb -name "*kernel_trace.csv" | head -1) corr_dm_encode 20 > $O/step_trace.txt 2>&1
python $R/tools/trace_step.py $(find /tmp/fb -name "*kernel_trace.csv" | head -1) corr_dm_encode 21 > $O/step_trace_b.txt 2>&1
cd $R; tail -c 300 $O/bench.err; tail -c 400 $O/bench.json
