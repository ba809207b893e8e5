#!/bin/bash
# end-of-round evidence run (through gpurun): bench line, per-kernel stats of the same command, step trace, render pass stats,
# PMC passes of the correlation lookups and of the decoders, training iteration and low-memory global BA stats, BA solve sizes.
# Outputs land in gpurun_out/final/; tools/collect_profiles.sh copies them into profiles/ with the round prefix.
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/fb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fb -o b -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --g8-only > $O/bench_prof.json 2> $O/bench_prof.err
cp $(find /tmp/fb -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/trace_step.py $(find /tmp/fb -name "*kernel_trace.csv" | head -1) corr_dm_encode 20 > $O/step_trace.txt 2>&1
rm -rf /tmp/fr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fr -o r -- python $R/tools/prof_render.py > /dev/null 2>&1
cp $(find /tmp/fr -name "*kernel_stats.csv" | head -1) $O/render_kernel_stats.csv
rm -rf /tmp/ft; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ft -o t -- python $R/tools/prof_train.py > $O/train.log 2>&1
cp $(find /tmp/ft -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
rm -rf /tmp/fl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl -o l -- python $R/tools/time_backend.py 128 512 > $O/lowmem.log 2>&1
cp $(find /tmp/fl -name "*kernel_stats.csv" | head -1) $O/lowmem_kernel_stats.csv
for K in 14 26 50; do
  rm -rf /tmp/fs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fs -o s -- python $R/tools/prof_ba_sizes.py $K 2>&1 | grep "K=" >> $O/ba_sizes.txt
  python $R/tools/show_stats.py $(find /tmp/fs -name "*kernel_stats.csv" | head -1) 80 | grep -E "ba_|chol" >> $O/ba_sizes.txt
done
cd $R
python tools/render_shard_time.py 2>&1 | grep world > $O/render_shards.txt
bash tools/pmc_corr.sh > $O/pmc_corr.log 2>&1; cp gpurun_out/pmc_corr/summary.json $O/pmc_corr.json
bash tools/pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma.txt $O/pmc_mfma.txt
bash tools/pmc_kernel.sh mlp_ tools/prof_render.py > $O/pmc_render.txt 2>&1
bash tools/pmc_kernel.sh knn_query tools/prof_render.py >> $O/pmc_render.txt 2>&1
bash tools/pmc_passes.sh > $O/pmc_passes.log 2>&1; cp gpurun_out/pmc_kernels.json $O/pmc_kernels.json 2>/dev/null
bash tools/conv_pp_timeline.sh > /dev/null 2>&1
tail -c 600 $O/bench.json
