#!/bin/bash
# timing ablations of conv8_kernel on the 448->256 layer (results are wrong with any bit set):
# 1 = no DMA after the second tile, 2 = no fragment reads, 4 = no MFMAs, 8 = vmcnt(0) instead of counted waits,
# 16 = every DMA re-reads tile (chunk 0, centre tap), 32 = no pixel units, 64 = no weight units
for d in ${@:-0 1 2 4 8 3 5 6 7}; do
  echo -n "dbg=$d: "; GLORIE_CONV8_DBG=$d timeout 100 python tools/bench_conv.py 2>&1 | grep -e "448-> 256" | sed 's/.*igemm//'
done
