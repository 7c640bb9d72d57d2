#!/bin/bash
# A/B builds of spatial_shared_cols_kernel at the evaluation shape (T = 28, K = 8, beam 5): usage
#   tools/probes/cols_probe.sh build name1:"-DSTATTN_COLS_ABL=1" ...   (here: variants of attn.o linked into tools/_var/libstattn_<name>.so)
#   tools/probes/cols_probe.sh run [nvid ...]                          (on the GPU box; the update runs as its own launch; restores the product library)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PKG=$ROOT/video-description-with-spatial-temporal-attention_amd
CS=$PKG/csrc
VAR=$ROOT/tools/_var
if [ "$1" = build ]; then
    shift; rm -rf $VAR; mkdir -p $VAR
    make -C $CS -j8 >/dev/null
    OBJS=$(cd $CS && ls *.o | grep -v '^attn.o$' | sed "s|^|$CS/|")
    for spec in "$@"; do
        name=${spec%%:*}; flags=${spec#*:}
        hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-variable -mllvm -amdgpu-mfma-vgpr-form $flags -c $CS/attn.hip -o $VAR/attn_$name.o
        hipcc --offload-arch=gfx950 -shared -fPIC -o $VAR/libstattn_$name.so $OBJS $VAR/attn_$name.o -ldl
        rm -f $VAR/attn_$name.o
        echo "built $name ($flags)"
    done
else
    shift || true
    cp $PKG/libstattn.so $VAR/_product.so
    export PROBE_T=28 PROBE_K=8 STATTN_NO_UPDATE_RIDER=1 STATTN_SHARED_MIN=1 STATTN_SHARED_COLS=1
    for so in $VAR/libstattn_*.so; do
        n=$(basename $so .so); n=${n#libstattn_}
        cp $so $PKG/libstattn.so
        echo "== $n"
        python $ROOT/tools/probes/shared_rounds_probe.py "${@:-51}" 2>/dev/null | cut -c1-100
    done
    cp $VAR/_product.so $PKG/libstattn.so
fi
