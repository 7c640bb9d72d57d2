#!/bin/bash
# tools/_var/libstattn_tools.so: the product sources compiled with -DSTATTN_TOOL_SWITCHES, i.e. the same kernels (the macro touches host
# code only: csrc/switches.h) with launchers that listen to the tool switches (STATTN_GEMM_TILE, STATTN_SHARED_MIN, STATTN_PW_R ...).
# The sweep / A-B tools of this directory need it:   tools/with_variant.sh tools python tools/gemm_bf16_sweep.py
# The product library ignores every variable that is not on the list in csrc/switches.h.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/video-description-with-spatial-temporal-attention_amd/csrc
O=$ROOT/tools/_var/tools_obj; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DSTATTN_TOOL_SWITCHES"
pids=""
for f in $C/*.hip $C/*.cpp; do
    b=$(basename $f); b=${b%.*}
    extra=""; case $b in attn|bwd) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
    ( hipcc $F $extra -c $f -o $O/$b.o ) &
    pids="$pids $!"
    if [ $(jobs -r | wc -l) -ge 8 ]; then wait -n; fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_var/libstattn_tools.so $O/*.o -ldl
rm -rf $O
echo "built tools/_var/libstattn_tools.so"
