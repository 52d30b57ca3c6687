// lpa_comm.hip -- slab-to-slab transport of the 1-D x decomposition: grouped nearest-neighbour face messages enqueued
// on a HIP stream, no host synchronisation.
//
// The reference moves its faces with MPI point-to-point calls, one Isend / Irecv per (patch, boundary, attribute) on
// three duplicated communicators (core/mpi/mpi_manager.py:96-298, core/mpi/sync_fields2d.c:365-640, tags :577-578).
// Here a rank has two neighbours and every exchange is ONE ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd on
// the stream the step runs on (RCCL: device-to-device over the xGMI links, one fused kernel per group), issued from C
// inside lpa_step -- the host cost of a message round is a few microseconds instead of a Python batch_isend_irecv.
//
// RCCL is dlopen'ed: the library carries no link-time dependency on it (it loads on a box without RCCL, and inside a
// PyTorch process it binds the RCCL that process has loaded already, so there is one RCCL and one HIP runtime).
#include <dlfcn.h>

#include "lpa_common.hpp"

namespace {

// ---- the few RCCL entry points used (signatures of rccl.h, NCCL 2.x ABI) --------------------------------------
struct NcclUniqueId { char internal[128]; };
typedef void *NcclComm;
enum { NCCL_FLOAT64 = 8 };   // ncclDataType_t: ncclFloat64 = ncclDouble = 8

struct RcclApi {
    void *handle = nullptr;
    int (*GetVersion)(int *) = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

RcclApi g_rccl;

int load_rccl(const char *path) {
    if (g_rccl.handle) return LPA_OK;
    const char *name = (path && *path) ? path : "librccl.so";
    void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h && !(path && *path)) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        lpa_set_error("lpa_comm: cannot load %s: %s", name, dlerror());
        return LPA_ERR_ARG;
    }
    RcclApi a;
    a.handle = h;
#define LPA_SYM(field, sym)                                                  \
    *(void **)(&a.field) = dlsym(h, sym);                                    \
    if (!a.field) {                                                          \
        lpa_set_error("lpa_comm: %s does not export %s", name, sym);         \
        dlclose(h);                                                          \
        return LPA_ERR_ARG;                                                  \
    }
    LPA_SYM(GetVersion, "ncclGetVersion")
    LPA_SYM(GetUniqueId, "ncclGetUniqueId")
    LPA_SYM(CommInitRank, "ncclCommInitRank")
    LPA_SYM(CommDestroy, "ncclCommDestroy")
    LPA_SYM(GroupStart, "ncclGroupStart")
    LPA_SYM(GroupEnd, "ncclGroupEnd")
    LPA_SYM(Send, "ncclSend")
    LPA_SYM(Recv, "ncclRecv")
    LPA_SYM(GetErrorString, "ncclGetErrorString")
#undef LPA_SYM
    g_rccl = a;
    return LPA_OK;
}

#define LPA_NCCL(call, what)                                                               \
    do {                                                                                   \
        int r_ = (call);                                                                   \
        if (r_ != 0) {                                                                     \
            lpa_set_error("lpa_comm: %s failed: %s", what, g_rccl.GetErrorString(r_));     \
            return LPA_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

// ---- loopback: every segment of a round in one launch ------------------------------------------------------------
constexpr int MAX_SEGS = 48;
struct CopySegs {
    const double *src[MAX_SEGS];
    double *dst[MAX_SEGS];
    long n[MAX_SEGS];
};

__global__ void __launch_bounds__(256) k_copy_segments(CopySegs s) {
    const int k = blockIdx.y;
    const double *__restrict__ a = s.src[k];
    double *__restrict__ b = s.dst[k];
    const long n = s.n[k];
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) b[t] = a[t];
}

}  // namespace

struct lpa_comm {
    int kind, rank, size, left, right, periodic, version;
    NcclComm nccl;
    hipStream_t side = nullptr;          // second stream of the overlapped steps (lpa_step), created on first use
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

// the communicator's second stream (high priority: what runs there goes first) and its events, created on first use
// (ev_early, may be NULL: a third event for a fork that precedes the ev_ready one)
int lpai_comm_side(lpa_comm *c, void **side, void **ev_ready, void **ev_done, void **ev_early) {
    LPA_REQUIRE(c && side && ev_ready && ev_done, "lpai_comm_side: bad args");
    if (!c->side) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);       // (hi = the numerically lowest = highest priority)
        if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, hi) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev[2], hipEventDisableTiming) != hipSuccess) {
            lpa_set_error("lpa_comm: cannot create the side stream");
            return LPA_ERR_HIP;
        }
    }
    *side = c->side; *ev_ready = c->ev[0]; *ev_done = c->ev[1];
    if (ev_early) *ev_early = c->ev[2];
    return LPA_OK;
}

extern "C" int lpa_comm_unique_id(void *id128, const char *librccl_path) {
    LPA_REQUIRE(id128, "lpa_comm_unique_id: null id");
    if (int e = load_rccl(librccl_path)) return e;
    NcclUniqueId id;
    LPA_NCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id128, &id, sizeof id);
    return LPA_OK;
}

static void ring_neighbours(lpa_comm *c) {
    c->left = (c->periodic || c->rank > 0) ? (c->rank - 1 + c->size) % c->size : -1;
    c->right = (c->periodic || c->rank < c->size - 1) ? (c->rank + 1) % c->size : -1;
}

extern "C" int lpa_comm_create_rccl(lpa_comm **out, const void *id128, int32_t rank, int32_t size, int32_t periodic,
                                    const char *librccl_path) {
    LPA_REQUIRE(out && id128 && size >= 1 && rank >= 0 && rank < size, "lpa_comm_create_rccl: bad args");
    if (int e = load_rccl(librccl_path)) return e;
    NcclUniqueId id;
    memcpy(&id, id128, sizeof id);
    lpa_comm *c = new lpa_comm();
    c->kind = LPA_COMM_RCCL; c->rank = rank; c->size = size; c->periodic = periodic != 0;
    ring_neighbours(c);
    c->version = 0;
    g_rccl.GetVersion(&c->version);
    int r = g_rccl.CommInitRank(&c->nccl, size, id, rank);
    if (r != 0) {
        lpa_set_error("lpa_comm: ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return LPA_ERR_HIP;
    }
    *out = c;
    return LPA_OK;
}

extern "C" int lpa_comm_create_loopback(lpa_comm **out, int32_t size, int32_t periodic) {
    LPA_REQUIRE(out && (size == 1 || size == 2) && periodic,
                "lpa_comm_create_loopback: a periodic ring of 1 slab, or of 2 (this slab and its translated copy)");
    lpa_comm *c = new lpa_comm();
    c->kind = LPA_COMM_LOOPBACK; c->rank = 0; c->size = size; c->periodic = 1; c->version = 0; c->nccl = nullptr;
    ring_neighbours(c);
    *out = c;
    return LPA_OK;
}

extern "C" int lpa_comm_destroy(lpa_comm *c) {
    if (!c) return LPA_OK;
    if (c->kind == LPA_COMM_RCCL && c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
        for (int k = 0; k < 3; k++) (void)hipEventDestroy(c->ev[k]);
    }
    delete c;
    return LPA_OK;
}

extern "C" int lpa_comm_info(const lpa_comm *c, int32_t info[6]) {
    LPA_REQUIRE(c && info, "lpa_comm_info: bad args");
    info[0] = c->kind; info[1] = c->rank; info[2] = c->size; info[3] = c->left; info[4] = c->right; info[5] = c->version;
    return LPA_OK;
}

extern "C" int lpa_comm_exchange(lpa_comm *c, const lpa_face_msg *msgs, int32_t nmsgs, void *stream) {
    LPA_REQUIRE(c && (msgs || nmsgs == 0) && nmsgs >= 0, "lpa_comm_exchange: bad args");
    const bool to_left = c->left >= 0, to_right = c->right >= 0;
    for (int k = 0; k < nmsgs; k++) {
        const lpa_face_msg &m = msgs[k];
        LPA_REQUIRE(m.n_send_lo >= 0 && m.n_send_hi >= 0 && m.n_recv_lo >= 0 && m.n_recv_hi >= 0,
                    "lpa_comm_exchange: negative count");
        LPA_REQUIRE(!(to_left && m.n_send_lo > 0) || m.send_lo, "lpa_comm_exchange: send_lo missing");
        LPA_REQUIRE(!(to_right && m.n_send_hi > 0) || m.send_hi, "lpa_comm_exchange: send_hi missing");
        LPA_REQUIRE(!(to_left && m.n_recv_lo > 0) || m.recv_lo, "lpa_comm_exchange: recv_lo missing");
        LPA_REQUIRE(!(to_right && m.n_recv_hi > 0) || m.recv_hi, "lpa_comm_exchange: recv_hi missing");
    }
    hipStream_t st = (hipStream_t)stream;
    if (c->kind == LPA_COMM_LOOPBACK) {
        // what the left neighbour (a copy of this slab) sends rightwards is this slab's own send_hi, and vice versa
        CopySegs s;
        int ns = 0;
        long longest = 0;
        auto flush = [&]() -> int {
            if (!ns) return LPA_OK;
            long nb = (longest + 255) / 256;
            if (nb > 256) nb = 256;
            hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)nb, ns), dim3(256), 0, st, s);
            LPA_CHECK_LAUNCH("lpa_comm_exchange (loopback)");
            ns = 0; longest = 0;
            return LPA_OK;
        };
        for (int k = 0; k < nmsgs; k++) {
            const lpa_face_msg &m = msgs[k];
            LPA_REQUIRE(m.n_send_hi == m.n_recv_lo && m.n_send_lo == m.n_recv_hi,
                        "lpa_comm_exchange: a loopback ring receives what it sends (counts differ)");
            if (m.n_send_hi > 0) {
                s.src[ns] = m.send_hi; s.dst[ns] = m.recv_lo; s.n[ns] = m.n_send_hi;
                if (m.n_send_hi > longest) longest = m.n_send_hi;
                if (++ns == MAX_SEGS) if (int e = flush()) return e;
            }
            if (m.n_send_lo > 0) {
                s.src[ns] = m.send_lo; s.dst[ns] = m.recv_hi; s.n[ns] = m.n_send_lo;
                if (m.n_send_lo > longest) longest = m.n_send_lo;
                if (++ns == MAX_SEGS) if (int e = flush()) return e;
            }
        }
        return flush();
    }
    // PAIRING RULE (ncclSend / ncclRecv carry no tag; messages between one pair of ranks match in posting order): per
    // message the sends go (hi, lo), the receives (lo, hi).  With three or more ranks the two neighbours differ; with
    // two ranks (left == right) the peer's first receive (its low face) takes my first send (my high face).  A ring of
    // one rank sends to itself: the same rule pairs send_hi with recv_lo.
    LPA_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
    int bad = 0;
    const char *what = "";
    for (int k = 0; k < nmsgs && !bad; k++) {
        const lpa_face_msg &m = msgs[k];
        if (!bad && to_right && m.n_send_hi > 0)
            bad = g_rccl.Send(m.send_hi, (size_t)m.n_send_hi, NCCL_FLOAT64, c->right, c->nccl, st), what = "ncclSend";
        if (!bad && to_left && m.n_send_lo > 0)
            bad = g_rccl.Send(m.send_lo, (size_t)m.n_send_lo, NCCL_FLOAT64, c->left, c->nccl, st), what = "ncclSend";
        if (!bad && to_left && m.n_recv_lo > 0)
            bad = g_rccl.Recv(m.recv_lo, (size_t)m.n_recv_lo, NCCL_FLOAT64, c->left, c->nccl, st), what = "ncclRecv";
        if (!bad && to_right && m.n_recv_hi > 0)
            bad = g_rccl.Recv(m.recv_hi, (size_t)m.n_recv_hi, NCCL_FLOAT64, c->right, c->nccl, st), what = "ncclRecv";
    }
    const int end = g_rccl.GroupEnd();      // (always closed, also after a failed post)
    if (bad) {
        lpa_set_error("lpa_comm: %s failed: %s", what, g_rccl.GetErrorString(bad));
        return LPA_ERR_HIP;
    }
    LPA_NCCL(end, "ncclGroupEnd");
    return LPA_OK;
}
