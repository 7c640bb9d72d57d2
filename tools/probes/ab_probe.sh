#!/bin/bash
# A/B of whole-library builds on one box: tools/_var/libstattn_<name>.so are swapped in turn under `leg_probe.py c1|c5` (N rounds).
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PKG=$ROOT/video-description-with-spatial-temporal-attention_amd
VAR=$ROOT/tools/_var
which=${1:-c1}; rounds=${2:-2}
cp $PKG/libstattn.so $VAR/_product.so
for r in $(seq $rounds); do
    for so in $VAR/libstattn_*.so; do
        n=$(basename $so .so); n=${n#libstattn_}
        cp $so $PKG/libstattn.so
        echo "== $n: $(python $ROOT/tools/probes/leg_probe.py $which 2>/dev/null | tr '\n' ' ' | cut -c1-420)"
    done
done
cp $VAR/_product.so $PKG/libstattn.so
