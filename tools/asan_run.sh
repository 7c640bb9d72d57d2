#!/bin/bash
# HIP AddressSanitizer run of a command on the GPU box.  The instrumented library is built in the build
# container (4 minutes on 8 cores; a scratch copy of csrc, `make ASAN=1`: the kernels of ASAN_OBJS device + host instrumented,
# gfx950:xnack+) and travels as libstattn_asan.so; here it is swapped in for the product library, "$@" runs with HSA_XNACK=1 and the
# ASan runtime preloaded, and the product library is restored.
#   build:  tools/asan_run.sh --build          (container)
#   run:    tools/asan_run.sh python -m pytest tests/test_gpu_parity.py -k beam -x -q        (GPU box)
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/video-description-with-spatial-temporal-attention_amd"
if [ "${1:-}" = "--build" ]; then
    rm -rf /tmp/asan_tree && mkdir -p /tmp/asan_tree/pkg && cp -r "$PKG/csrc" /tmp/asan_tree/pkg/csrc && cp -r "$ROOT/include" /tmp/asan_tree/include
    make -C /tmp/asan_tree/pkg/csrc clean >/dev/null 2>&1
    make -C /tmp/asan_tree/pkg/csrc -j8 ASAN=1 > /tmp/asan_build.log 2>&1 || { grep -n "rror" /tmp/asan_build.log | head; echo "ASAN BUILD FAILED"; exit 3; }
    cp /tmp/asan_tree/pkg/libstattn.so "$PKG/libstattn_asan.so" && echo "built $PKG/libstattn_asan.so"
    exit 0
fi
[ -f "$PKG/libstattn_asan.so" ] || { echo "no libstattn_asan.so: run tools/asan_run.sh --build in the container first"; exit 3; }
cp "$PKG/libstattn.so" /tmp/libstattn_product.so
cp "$PKG/libstattn_asan.so" "$PKG/libstattn.so"
RT="$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)"
echo "asan runtime: $RT"
HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 LD_PRELOAD="$RT" LD_LIBRARY_PATH="$(dirname "$RT"):/opt/rocm/lib:${LD_LIBRARY_PATH:-}" "$@"
rc=$?
cp /tmp/libstattn_product.so "$PKG/libstattn.so"
exit $rc
