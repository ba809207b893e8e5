#!/bin/bash
# end-of-round evidence run (through gpurun): bench line, per-kernel stats of the same command, render pass
# stats, BA solve sizes.  Outputs land in gpurun_out/final/.
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/fb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fb -o b -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
cp $(find /tmp/fb -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf /tmp/fr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fr -o r -- python $R/tools/prof_render.py > /dev/null 2>&1
cp $(find /tmp/fr -name "*kernel_stats.csv" | head -1) $O/render_kernel_stats.csv
for K in 14 26 50; do
  rm -rf /tmp/fs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fs -o s -- python $R/tools/prof_ba_sizes.py $K 2>&1 | grep "K=" >> $O/ba_sizes.txt
  python $R/tools/show_stats.py $(find /tmp/fs -name "*kernel_stats.csv" | head -1) 80 | grep -E "ba_|chol" >> $O/ba_sizes.txt
done
cd $R; python tools/render_shard_time.py 2>&1 | grep world > $O/render_shards.txt
tail -c 600 $O/bench.json
