#!/bin/bash
# render parity tests + per-kernel stats of four full-frame passes (run through gpurun)
python -m pytest tests/test_gpu_render.py -x -q 2>&1 | tail -3
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/tools/prof_render.py > /tmp/log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/render_stats.csv
python $R/tools/show_stats.py $f 6
