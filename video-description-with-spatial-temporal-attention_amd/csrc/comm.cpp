// Data-parallel gradient exchange inside libstattn.so: one RCCL communicator per handle (= per rank = per GPU),
// sum all-reduce of the flat gradient buffer over xGMI.  The reference is single-process;
// this is the "partition the caption batch over the 8 GPUs of one node" part of the path.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the ROCm collective library; a process that already
// carries one -- e.g. through torch -- shares it): single-GPU users never load it, and a box without it fails
// loudly in stattn_comm_init, nowhere else.
//
// Overlap (stattn_handle::comm_overlap): stattn_backward hands regions of the gradient buffer to
// comm_reduce_range() as soon as they are final -- the readout gradients before the reverse scan starts, the
// decoder / projection / embedding regions while the remaining weight-gradient GEMMs still run.  Each region is
// reduced on a private stream behind an event recorded on the compute stream; stattn_allreduce_grads() reduces
// whatever has not been handed over and makes the compute stream wait for all of it.
#include "handle.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err, path;
};

// Resolution order: (1) an RCCL this process ALREADY carries (RTLD_NOLOAD: torch ships its own librccl.so under
// torch/lib and a second copy of the library in one process would mean two sets of IPC / topology state), (2) the
// librccl that sits NEXT TO the HIP runtime this library is bound to (torch/lib or the ROCm install: RCCL opens
// "libhsa-runtime64.so" by name, and an RCCL from another directory than the running HIP / HSA pair brings a second,
// uninitialised HSA runtime into the process -- ncclCommInitRank then reports "no ROCm-capable device"), (3) the
// loader path, (4) the ROCm install.  The file that was taken is reported by stattn_comm_library_path() and printed
// once to stderr by the first stattn_comm_init, so a scaling log shows which RCCL ran.
Rccl* rccl() {
    static Rccl r;
    if (r.lib || !r.err.empty()) return &r;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (r.lib) break;
    }
    if (!r.lib) {
        Dl_info hip{};
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &hip) && hip.dli_fname) {
            std::string dir(hip.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char* n : {"librccl.so", "librccl.so.1"}) {
                    r.lib = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL);
                    if (r.lib) break;
                }
            }
        }
    }
    for (const char* n : names) {
        if (r.lib) break;
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.lib) { r.err = std::string("cannot load librccl.so: ") + dlerror(); return &r; }
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.err.empty()) { dlclose(r.lib); r.lib = nullptr; return &r; }
    Dl_info info{};
    r.path = (dladdr(reinterpret_cast<void*>(r.AllReduce), &info) && info.dli_fname) ? info.dli_fname : "(unknown)";
    return &r;
}

#define NCCLCHK(h, expr)                                                                              \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess)                                                                        \
            return fail(h, STATTN_EHIP, "%s failed: %s", #expr, rccl()->GetErrorString(r_));          \
    } while (0)

}  // namespace

namespace stattn_detail {

// Called by stattn_backward: gradients in [off, off + n) of the flat buffer are final on the compute stream.
// With a multi-rank communicator and overlap enabled the range is summed over ranks on the side stream; ranges
// must be handed over at most once per backward.  No-op otherwise.
int comm_reduce_range(stattn_handle* h, size_t off, size_t n) {
    // (comm_overlap == 2: also with a single rank -- how the event / side-stream path is exercised on a one-GPU box)
    if (!h->comm || !h->comm_overlap || n == 0 || (h->comm_nranks < 2 && h->comm_overlap != 2)) return STATTN_OK;
    hipEvent_t ready = h->comm_ready[h->comm_regions & 7];         // an event of its own per region of a pass (five per pass)
    ++h->comm_regions;
    HIPCHK(h, hipEventRecord(ready, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->comm_stream, ready, 0));
    NCCLCHK(h, rccl()->AllReduce(h->d_grads + off, h->d_grads + off, n, ncclFloat32, ncclSum,
                                 static_cast<ncclComm_t>(h->comm), h->comm_stream));
    h->comm_covered += n;
    return STATTN_OK;
}

// A new backward pass is about to rewrite the gradient buffer.  If the previous pass handed regions to the side stream
// and nobody called stattn_allreduce_grads since, those collectives may still be running in place on the same buffer:
// the compute stream waits for them first.
int comm_backward_begins(stattn_handle* h) {
    if (comm_pending(h)) {
        HIPCHK(h, hipEventRecord(h->comm_done, h->comm_stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->comm_done, 0));
    }
    h->comm_covered = 0; h->comm_regions = 0; h->grads_reduced = false; h->comm_timed = false;
    return STATTN_OK;
}

void comm_release(stattn_handle* h) {
    if (h->comm) { (void)rccl()->CommDestroy(static_cast<ncclComm_t>(h->comm)); h->comm = nullptr; }
    if (h->comm_stream) { (void)hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr; }
    for (hipEvent_t& e : h->comm_ready) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (hipEvent_t* e : {&h->comm_done, &h->comm_t0, &h->comm_t1}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    h->comm_covered = 0; h->comm_regions = 0; h->comm_timed = false;
    h->comm_nranks = 1; h->comm_rank = 0;
}

}  // namespace stattn_detail

extern "C" {

int stattn_comm_unique_id(void* id_out) {
    if (!id_out) return STATTN_EINVAL;
    Rccl* r = rccl();
    if (!r->lib) { g_create_error = r->err; return STATTN_EHIP; }
    ncclUniqueId id;
    ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(e); return STATTN_EHIP; }
    static_assert(sizeof(id) == STATTN_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof id);
    return STATTN_OK;
}

int stattn_comm_init(stattn_handle* h, int rank, int nranks, const void* id_bytes) {
    if (!h || !id_bytes || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, STATTN_EINVAL, "comm_init: bad argument");
    if (h->comm) return fail(h, STATTN_ESTATE, "comm_init: this handle already has a communicator");
    Rccl* r = rccl();
    if (!r->lib) return fail(h, STATTN_EHIP, "comm_init: %s", r->err.c_str());
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    // stream and events first: a handle never holds a communicator without them
    auto make = [&]() -> int {
        HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
        for (hipEvent_t& e : h->comm_ready) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->comm_done, hipEventDisableTiming));
        HIPCHK(h, hipEventCreate(&h->comm_t0));
        HIPCHK(h, hipEventCreate(&h->comm_t1));
        return STATTN_OK;
    };
    if (int rc = make()) { comm_release(h); return rc; }
    ncclComm_t c = nullptr;
    const ncclResult_t e = r->CommInitRank(&c, nranks, id, rank);
    if (e != ncclSuccess) {
        comm_release(h);
        return fail(h, STATTN_EHIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, r->GetErrorString(e));
    }
    h->comm = c; h->comm_rank = rank; h->comm_nranks = nranks;
    static const char* noov = sw_product("STATTN_COMM_NO_OVERLAP");
    h->comm_overlap = noov ? 0 : 1;
    static bool said = false;
    if (!said && rank == 0) { said = true; fprintf(stderr, "stattn: RCCL from %s, %d rank(s)\n", r->path.c_str(), nranks); }
    return STATTN_OK;
}

const char* stattn_comm_library_path(void) {
    Rccl* r = rccl();
    return r->lib ? r->path.c_str() : "";
}

int stattn_comm_destroy(stattn_handle* h) {
    if (!h) return STATTN_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    comm_release(h);
    return STATTN_OK;
}

int stattn_comm_info(const stattn_handle* h, int* rank, int* nranks) {
    if (!h) return STATTN_EINVAL;
    if (rank) *rank = h->comm_rank;
    if (nranks) *nranks = h->comm ? h->comm_nranks : 0;
    return STATTN_OK;
}

int stattn_comm_set_overlap(stattn_handle* h, int enable) {
    if (!h) return STATTN_EINVAL;
    h->comm_overlap = enable == 2 ? 2 : (enable ? 1 : 0);
    return STATTN_OK;
}

int stattn_allreduce_grads(stattn_handle* h) {
    if (!h) return STATTN_EINVAL;
    if (!h->have_bwd) return fail(h, STATTN_ESTATE, "allreduce_grads: no fresh gradient (call stattn_backward)");
    if (h->grads_reduced) return fail(h, STATTN_ESTATE, "allreduce_grads: this gradient has already been summed over the ranks");
    if (!h->comm || (h->comm_nranks < 2 && h->comm_covered == 0)) { h->grads_reduced = true; return STATTN_OK; }   // single rank: the sum is the buffer
    HIPCHK(h, hipSetDevice(h->device));
    // comm_t0 .. comm_t1 on the compute stream bracket what the step actually pays for the exchange: the whole
    // collective when nothing was overlapped, otherwise only the wait for the tail of the side stream
    HIPCHK(h, hipEventRecord(h->comm_t0, h->stream));
    if (h->comm_covered == 0) {
        // nothing was overlapped: ONE collective over the whole buffer on the compute stream
        NCCLCHK(h, rccl()->AllReduce(h->d_grads, h->d_grads, h->nflat, ncclFloat32, ncclSum, static_cast<ncclComm_t>(h->comm), h->stream));
    } else {
        if (h->comm_covered != h->nflat)
            return fail(h, STATTN_ESTATE, "allreduce_grads: internal error, %zu of %zu gradient elements were handed to the overlapped reduce",
                        h->comm_covered, h->nflat);
        HIPCHK(h, hipEventRecord(h->comm_done, h->comm_stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->comm_done, 0));
    }
    HIPCHK(h, hipEventRecord(h->comm_t1, h->stream));
    h->comm_timed = true;
    h->grads_reduced = true;
    return STATTN_OK;
}

int stattn_comm_stats(stattn_handle* h, int* nranks, int* overlap, int* regions, float* exposed_ms) {
    if (!h) return STATTN_EINVAL;
    if (nranks) *nranks = h->comm ? h->comm_nranks : 0;
    if (overlap) *overlap = h->comm ? h->comm_overlap : 0;
    if (regions) *regions = h->comm_regions;
    if (exposed_ms) {
        *exposed_ms = 0.f;
        if (h->comm_timed) {
            HIPCHK(h, hipSetDevice(h->device));
            HIPCHK(h, hipEventSynchronize(h->comm_t1));
            HIPCHK(h, hipEventElapsedTime(exposed_ms, h->comm_t0, h->comm_t1));
        }
    }
    return STATTN_OK;
}

int stattn_broadcast_params(stattn_handle* h, int root) {
    if (!h) return STATTN_EINVAL;
    if (!h->comm || h->comm_nranks < 2) return STATTN_OK;
    if (root < 0 || root >= h->comm_nranks) return fail(h, STATTN_EINVAL, "broadcast_params: bad root");
    HIPCHK(h, hipSetDevice(h->device));
    NCCLCHK(h, rccl()->Broadcast(h->d_params, h->d_params, h->nflat, ncclFloat32, root, static_cast<ncclComm_t>(h->comm), h->stream));
    h->ck_proj = false; h->have_fwd = false;
    return STATTN_OK;
}

int stattn_allreduce_scalars(stattn_handle* h, float* vals, int n) {
    if (!h || !vals || n < 1 || n > 64) return fail(h, STATTN_EINVAL, "allreduce_scalars: bad argument (1 <= n <= 64)");
    if (!h->comm || h->comm_nranks < 2) return STATTN_OK;
    HIPCHK(h, hipSetDevice(h->device));
    float* d;
    CHK(getbuf_t(h, "comm_scalars", (size_t)64, &d));
    HIPCHK(h, hipMemcpyAsync(d, vals, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    NCCLCHK(h, rccl()->AllReduce(d, d, (size_t)n, ncclFloat32, ncclSum, static_cast<ncclComm_t>(h->comm), h->stream));
    HIPCHK(h, hipMemcpyAsync(vals, d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return STATTN_OK;
}

}  // extern "C"
