"""interleaved A/B of the 128 x 512 ping-pong tile (conv_ppw_kernel) on the update operator's 128-channel layers at G8:
    python tools/bench_ppw.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_conv  # noqa: E402

if __name__ == "__main__":
    bench_conv.ppw_ab(int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 36)
