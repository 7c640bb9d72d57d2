#!/bin/bash
# A/B of GEMM kernel builds: tools/_var/libstattn_gv*.so swapped in turn under tools/probes/gemm_ab.py (TFLOP/s per shape)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PKG=$ROOT/video-description-with-spatial-temporal-attention_amd
VAR=$ROOT/tools/_var
rounds=${1:-2}
cp $PKG/libstattn.so $VAR/_product.so
for r in $(seq $rounds); do
    for so in $VAR/libstattn_gv*.so; do
        n=$(basename $so .so); n=${n#libstattn_}
        cp $so $PKG/libstattn.so
        for t in 22 11; do echo "== $n $(STATTN_GEMM_TILE=$t python $ROOT/tools/probes/gemm_ab.py 2>&1 | tail -1)"; done
    done
done
cp $VAR/_product.so $PKG/libstattn.so
