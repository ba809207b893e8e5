#!/bin/bash
# R2 as the PRODUCT performs it (the 128-byte feature rows of the 8 neighbours are pulled inside the decoder
# kernels, not by a stand-alone gather): mlp.hip built with -DEXP_GATHER_ONLY keeps the ids / weights / feature-row / position
# loads and their interpolation of mlp_geo_v4 and mlp_nb_v4 and removes the networks; rocprofv3 times them on four full-frame
# passes of the bench workload (614,400 samples per launch) next to the unchanged search launch.
#   -> gpurun_out/knn_gather_phase.json  (copied to profiles/<round>_knn_gather_phase.json, read by bench.py)
R=$PWD; export TMPDIR=/tmp; export GLORIE_EXTRA_HIPFLAGS_ONLY=mlp.hip
mkdir -p gpurun_out
GLORIE_EXTRA_HIPFLAGS="-DEXP_GATHER_ONLY" python glorie_slam_amd/build.py > /dev/null 2>&1 || exit 1
(cd /tmp && rm -rf /tmp/gp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o r -- python $R/tools/prof_render.py > /tmp/gp.log 2>&1)
f=$(find /tmp/gp -name "*kernel_stats.csv" | head -1)
python - "$f" > gpurun_out/knn_gather_phase.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pick = lambda key: next(float(r["AverageNs"]) / 1e6 for r in rows if key in r["Name"])
out = {"samples_per_launch": 614400, "build": "-DEXP_GATHER_ONLY (csrc/mlp.hip)",
       "geo_gather_ms": pick("mlp_geo_v4"), "nb_gather_ms": pick("mlp_nb_v4"), "search_ms": pick("knn_query_kernel"),
       "calls": {r["Name"][:40]: int(r["Calls"]) for r in rows[:6]}}
json.dump(out, sys.stdout, indent=1)
PY
GLORIE_EXTRA_HIPFLAGS="" python glorie_slam_amd/build.py > /dev/null 2>&1
cat gpurun_out/knn_gather_phase.json
