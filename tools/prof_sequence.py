"""Host profile of the long synthetic sequence (pipeline.synthetic_long_runner, 100 frames): where a kept keyframe's ~60 ms go
on the host; run under rocprofv3 --kernel-trace --stats for the device side.   python tools/prof_sequence.py [frames]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glorie_slam_amd.pipeline import synthetic_long_runner  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
run, lc, frames = synthetic_long_runner(dev, n_frames=n, map_iters=20)
if os.environ.get("TRACK_GRAPHS") == "0":          # A/B: the frontend's updates always eager
    run.frontend.graph.use_graphs = False
if os.environ.get("TRIM_GB"):
    run.frontend.graph.capture_trim_bytes = int(float(os.environ["TRIM_GB"]) * 2 ** 30)
intr = lc["intrinsics"]
it = iter(frames())
pr = cProfile.Profile()
t0 = time.perf_counter()
for k, (ts, im) in enumerate(it):
    if k == 30:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pr.enable()
    run.track(ts, im, intr)
    run.map_pending()
torch.cuda.synchronize()
pr.disable()
t2 = time.perf_counter()
print(f"frames 30..{n}: {(t2 - t1) / (n - 30) * 1e3:.1f} ms per frame (host wall incl. device waits)")
import numpy as np
tr, kp = np.array(run.timing["track_ms"]), np.array(run.timing["kept"])
print(f"track ms per kept keyframe p50 {np.percentile(tr[kp][8:], 50):.1f} mean {tr[kp][8:].mean():.1f} p95 {np.percentile(tr[kp][8:], 95):.1f}; "
      f"mapping iteration {np.mean(run.timing['map_iter_ms'][2:]):.2f} ms; allocator: {torch.cuda.memory_stats()['num_device_alloc']} device allocs, "
      f"{torch.cuda.memory_reserved() / 2**30:.1f} GB reserved")
st = pstats.Stats(pr).sort_stats("cumulative")
st.print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
st.print_callers("method 'to' of")
st.print_callees("render_train.py:195")
st.print_callees("pipeline.py:116")
