#!/bin/bash
# A/B builds of spatial_shared_kernel (configs[4] beam search): usage
#   tools/probes/shared_probe.sh build name1:"-DFOO=1" name2:"-DBAR=2" ...   (here: variants of attn.o linked into tools/_var/libstattn_<name>.so)
#   tools/probes/shared_probe.sh run                                           (on the GPU box: leg_probe c5 under every variant; restores the product library)
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
    cp $PKG/libstattn.so $VAR/_product.so
    for so in $VAR/libstattn_*.so; do
        n=$(basename $so .so); n=${n#libstattn_}
        cp $so $PKG/libstattn.so
        echo "== $n: $(python $ROOT/tools/probes/leg_probe.py c5 2>/dev/null | grep -E 'value|spatial' | tr '\n' ' ')"
    done
    cp $VAR/_product.so $PKG/libstattn.so
fi
