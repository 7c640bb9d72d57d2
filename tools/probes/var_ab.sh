#!/bin/bash
# A/B of whole-library builds on one box: every tools/_var/libstattn_<name>.so is swapped in in turn under <command> (its last lines are shown)
# usage: tools/probes/var_ab.sh <rounds> <tail lines> <command...>
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PKG=$ROOT/video-description-with-spatial-temporal-attention_amd
VAR=$ROOT/tools/_var
rounds=$1; lines=$2; shift 2
cp $PKG/libstattn.so $VAR/_product.so
for r in $(seq $rounds); do
    for so in $VAR/libstattn_*.so; do
        n=$(basename $so .so); n=${n#libstattn_}
        cp $so $PKG/libstattn.so
        echo "== $n"; "$@" 2>&1 | grep -v amdgpu.ids | tail -$lines
    done
done
cp $VAR/_product.so $PKG/libstattn.so
