for v in 0 1 2; do echo "== variant $v"; GLORIE_CONV_VARIANT=$v timeout 300 python tools/bench_conv.py 2>&1 | grep "igemm"; done
