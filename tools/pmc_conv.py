"""Workload for PMC passes on the GRU gate convolution alone (bench shapes).
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out -o c -- python tools/pmc_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
launch, _ = bench.gru_gate_conv_workload(dev, 36, 60, 80)
for _ in range(5):
    launch()
torch.cuda.synchronize()
