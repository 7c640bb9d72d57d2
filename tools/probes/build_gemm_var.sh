#!/bin/bash
# usage: tools/probes/build_gemm_var.sh <name> <extra hipcc flags...>  -> tools/_var/libstattn_<name>.so ($SRC.hip (default gemm) rebuilt with the flags)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/video-description-with-spatial-temporal-attention_amd/csrc
n=$1; shift
SRC=${SRC:-gemm}
mkdir -p $ROOT/tools/_var
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $C/$SRC.hip -o $ROOT/tools/_var/${SRC}_$n.o || exit 1
objs=$(cd $C && ls *.o | grep -v "^$SRC.o\$" | sed "s|^|$C/|")
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_var/libstattn_$n.so $objs $ROOT/tools/_var/${SRC}_$n.o -ldl
