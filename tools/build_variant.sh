#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>" [file.hip ...]
# An alternate libstattn with some kernels rebuilt under extra flags (A/B of a compile-time switch on the same box):
#   tools/_var/libstattn_<name>.so = the product objects, except the listed .hip files (default: every file whose kernels use the wave
#   reductions of devmath.h) recompiled with the flags.  Run something under it with tools/with_variant.sh <name> <command...>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/video-description-with-spatial-temporal-attention_amd/csrc
n=$1; flags=$2; shift 2
files=${@:-attn.hip bwd.hip misc.hip panel.hip panelw.hip beam.hip skinny.hip}
make -C $C -j8 >/dev/null
mkdir -p $ROOT/tools/_var/$n
skip=""
for f in $files; do
    b=${f%.hip}
    extra=""; case $b in attn|bwd) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
    ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra $flags -c $C/$f -o $ROOT/tools/_var/$n/$b.o ) &
    skip="$skip $b.o"
done
wait
objs=$(cd $C && for o in *.o; do case " $skip " in *" $o "*) ;; *) echo $C/$o;; esac; done)
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_var/libstattn_$n.so $objs $ROOT/tools/_var/$n/*.o -ldl
rm -rf $ROOT/tools/_var/$n
echo "built tools/_var/libstattn_$n.so ($flags: $files)"
