#!/bin/bash
# gpurun_out/final/* (tools/final_profile.sh) -> profiles/<round>_*   usage: tools/collect_profiles.sh r03
P=${1:-r03}; O=gpurun_out/final
cp $O/bench.json profiles/${P}_final_bench.json
cp $O/bench_kernel_stats.csv profiles/${P}_final_bench_kernel_stats.csv
cp $O/step_trace.txt profiles/${P}_step_trace.txt
cp $O/render_kernel_stats.csv profiles/${P}_render_kernel_stats.csv
cp $O/train_kernel_stats.csv profiles/${P}_train_kernel_stats.csv
cp $O/lowmem_kernel_stats.csv profiles/${P}_lowmem_kernel_stats.csv
cp $O/lowmem.log profiles/${P}_lowmem_times.txt
cp $O/ba_sizes.txt profiles/${P}_ba_solve_sizes.txt
cp $O/render_shards.txt profiles/${P}_render_shard_projection.txt
cp $O/pmc_corr.json profiles/${P}_pmc_corr.json
cp $O/pmc_mfma.txt profiles/${P}_pmc_mfma.txt
cp $O/pmc_render.txt profiles/${P}_pmc_render.txt
cp $O/pmc_kernels.json profiles/${P}_pmc_kernels.json 2>/dev/null
ls -la profiles/${P}_*
