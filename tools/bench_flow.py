"""flow_encoder[0] (7x7, 4 -> 128) stand-alone at the bench shape: fp32-map kernel against the padded-fp16 kernel.
python tools/bench_flow.py  (FLOW_N = maps)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glorie_slam_amd import update_ops as U

dev = torch.device("cuda:0")
n, h, w = int(os.environ.get('FLOW_N', '36')), 60, 80
torch.manual_seed(0)
flow = torch.randn(n, h, w, 4, device=dev)
wgt = torch.randn(128, 4, 7, 7, device=dev) / 14
bias = torch.randn(128, device=dev)
wp = U.pack_flow_conv7(wgt)
out = torch.empty((n, 128, h, w), dtype=torch.float16, device=dev).contiguous(memory_format=torch.channels_last)
out2 = torch.empty_like(out)
pf = U.PaddedFlow(n, h, w, dev)
U.flow_pad(flow, pf)
ref = torch.relu(torch.nn.functional.conv2d(flow.permute(0, 3, 1, 2).half().float(), wgt.half().float(), bias, padding=3))


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


t0 = timed(lambda: U.flow_conv7(flow, wp, bias, out))
t1 = timed(lambda: U.flow_conv7_padded(pf, wp, bias, out2))
t2 = timed(lambda: U.flow_pad(flow, pf))
print("fp32 map %.1f us (err %.4f)   padded fp16 %.1f us (err %.4f, equal to fp32-map kernel: %s)   pad pass %.1f us" % (
    t0, float((out.float() - ref).abs().max()), t1, float((out2.float() - ref).abs().max()), bool(torch.equal(out, out2)), t2))
