// Red-zone scan (STATTN_DBG_REDZONE=1, handle.h DevBuf): one workgroup per canary region, every byte compared with the canary value;
// the lowest (region, offset) that differs is kept.  Debugging aid, never on a timed path.
#include "kernels.h"

namespace stattn {

namespace {

// *bad = min over damaged regions of (region index << 40 | first damaged byte offset << 8 | the byte found there); ~0 on entry = clean
__global__ __launch_bounds__(256) void redzone_scan_kernel(const RedzoneRegion* __restrict__ regs, int canary, unsigned long long* __restrict__ bad) {
    const RedzoneRegion r = regs[blockIdx.x];
    __shared__ unsigned long long first;
    if (threadIdx.x == 0) first = ~0ull;
    __syncthreads();
    for (size_t i = threadIdx.x; i < r.nbytes; i += 256) {
        const unsigned v = r.p[i];
        if (v != (unsigned)canary) { atomicMin(&first, ((unsigned long long)i << 8) | v); break; }     // (a thread's first hit is its lowest)
    }
    __syncthreads();
    if (threadIdx.x == 0 && first != ~0ull) atomicMin(bad, ((unsigned long long)blockIdx.x << 40) | (first & 0xffffffffffull));
}

}  // namespace

hipError_t launch_redzone_scan(hipStream_t s, const RedzoneRegion* regs, int nregs, int canary, unsigned long long* bad) {
    if (nregs <= 0) return hipSuccess;
    hipLaunchKernelGGL(redzone_scan_kernel, dim3(nregs), dim3(256), 0, s, regs, canary, bad);
    return hipGetLastError();
}

}  // namespace stattn
