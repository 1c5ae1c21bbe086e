// Multi-GPU pieces of the hot path behind the C ABI (nothing upstream corresponds to this file -- the
// reference is single-process / single-device, scripts/tilediffusion.py:257-383 drives ONE `p`).
//
// A shard context owns one RCCL communicator + one HIP stream per LOCAL rank:
//   * mdtile_shard_init(ndev, dev_ids)           single process, one rank per listed device (ncclCommInitAll) -- the form a webui
//                                                 process can use: the plugin keeps one `p`, the engine spreads the tiles
//   * mdtile_shard_init_rank(n, rank, id, dev)   one rank of a process-per-GPU job (ncclCommInitRank; bench.py under torchrun)
// RCCL is dlopen'ed on first use (libmdtile.so itself links no collective library and loads on hosts without RCCL).
// Transport "copy": when the listed devices repeat (a 1-GPU box exercising the N-rank flow) or MDTILE_SHARD_TRANSPORT=copy, the
// ranks of a single-process context move their slabs with hipMemcpyAsync + events instead -- same packing, same summation.
//
// mdtile_halo_exchange: diffusion tiles are split in contiguous bands of tile rows, one per rank (mdtile/sharding.py).  After a
// rank accumulated its own tiles (mdtile_blend with MDTILE_BLEND_PARTIAL) only the canvas rows that tiles of two bands touch need
// another rank's data: each rank packs those `rows x W x N*C` fp32 slabs, swaps them with the peers that share them (grouped
// ncclSend / ncclRecv: xGMI is point-to-point, two neighbour transfers of ~1 MB beat a ring all-reduce of the 33.5 MB canvas),
// and k_halo_add sums own + received pieces in ASCENDING RANK ORDER on both sides -> bit-identical sums everywhere.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <thread>
#include <vector>

using namespace mdt;

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // optional (NCCL >= 2.14): the interruptible bring-up probe, mdtile_shard_probe_rank
    ncclResult_t (*CommInitRankConfig)(ncclComm_t*, int, ncclUniqueId, int, ncclConfig_t*) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};

Rccl* rccl() {
    static Rccl R;
    static bool tried = false;
    if (tried) return R.h ? &R : nullptr;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
        R.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (R.h) break;
    }
    if (!R.h) return nullptr;
#define MDT_SYM(field, sym)                                              \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.h, sym));      \
    if (!R.field) { R.h = nullptr; return nullptr; }
    MDT_SYM(CommInitAll, "ncclCommInitAll")
    MDT_SYM(CommInitRank, "ncclCommInitRank")
    MDT_SYM(GetUniqueId, "ncclGetUniqueId")
    MDT_SYM(CommDestroy, "ncclCommDestroy")
    MDT_SYM(GroupStart, "ncclGroupStart")
    MDT_SYM(GroupEnd, "ncclGroupEnd")
    MDT_SYM(Send, "ncclSend")
    MDT_SYM(Recv, "ncclRecv")
    MDT_SYM(AllReduce, "ncclAllReduce")
    MDT_SYM(Broadcast, "ncclBroadcast")
    MDT_SYM(AllGather, "ncclAllGather")
    MDT_SYM(GetErrorString, "ncclGetErrorString")
#undef MDT_SYM
    R.CommInitRankConfig = reinterpret_cast<decltype(R.CommInitRankConfig)>(dlsym(R.h, "ncclCommInitRankConfig"));
    R.CommGetAsyncError = reinterpret_cast<decltype(R.CommGetAsyncError)>(dlsym(R.h, "ncclCommGetAsyncError"));
    R.CommAbort = reinterpret_cast<decltype(R.CommAbort)>(dlsym(R.h, "ncclCommAbort"));
    return &R;
}

#define MDT_NCCL(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t _r = (expr);                                                               \
        if (_r != ncclSuccess) {                                                                \
            mdt::set_error("%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(_r), __FILE__, __LINE__); \
            return MDTILE_E_HIP;                                                                \
        }                                                                                       \
    } while (0)

// Calls between ncclGroupStart and ncclGroupEnd: remember the first failure, keep going to the GroupEnd (an open group would
// poison every later RCCL call of the process), report afterwards.
struct GroupScope {
    Rccl* R;
    ncclResult_t first = ncclSuccess;
    const char* what = nullptr;
    explicit GroupScope(Rccl* r) : R(r) { note(R->GroupStart(), "ncclGroupStart"); }
    void note(ncclResult_t r, const char* w) {
        if (r != ncclSuccess && first == ncclSuccess) { first = r; what = w; }
    }
    int end(const char* where) {
        note(R->GroupEnd(), "ncclGroupEnd");
        if (first == ncclSuccess) return MDTILE_OK;
        mdt::set_error("%s: %s failed: %s", where, what, R->GetErrorString(first));
        return MDTILE_E_HIP;
    }
};

// slabs of rows [lo, hi) of every plane of a [planes, H, W] canvas <-> contiguous [planes, hi - lo, W]
__global__ __launch_bounds__(256) void k_slab_pack(const float* __restrict__ canvas, float* __restrict__ slab, int H, int W, int lo, int rows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)rows * W;
    if (i >= per) return;
    const int plane = blockIdx.y;
    slab[(size_t)plane * per + i] = canvas[((size_t)plane * H + lo) * W + i];
}

constexpr int MAX_HALO_PEERS = 8;
struct HaloSet {
    int peer[MAX_HALO_PEERS], lo[MAX_HALO_PEERS], hi[MAX_HALO_PEERS];
    const float* recv[MAX_HALO_PEERS];     // received slab of that peer: [planes, hi - lo, W]
    int n;
};

// canvas rows [row_lo, row_hi): every element shared with peers becomes the sum of all contributions in ascending rank order
__global__ __launch_bounds__(256) void k_halo_add(float* __restrict__ canvas, int H, int W, int row_lo, int rows, int me, const HaloSet S) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * W) return;
    const int y = row_lo + (int)(i / W), x = (int)(i % W);
    const int plane = blockIdx.y;
    float* p = canvas + ((size_t)plane * H + y) * W + x;
    float acc = 0.0f;
    bool first = true, mine_added = false, shared = false;
    // peers are listed in ascending rank order; the own piece is inserted at its place
    for (int k = 0; k <= S.n; ++k) {
        const bool own_turn = !mine_added && (k == S.n || S.peer[k] > me);
        if (own_turn) {
            acc = first ? *p : acc + *p;
            first = false;
            mine_added = true;
        }
        if (k == S.n) break;
        if (y >= S.lo[k] && y < S.hi[k]) {
            const float v = S.recv[k][((size_t)plane * (S.hi[k] - S.lo[k]) + (y - S.lo[k])) * W + x];
            acc = first ? v : acc + v;
            first = false;
            shared = true;
        }
    }
    if (shared) *p = acc;
}

}  // namespace

struct mdtile_shard {
    int nranks = 0, nlocal = 0, first = 0;     // ranks [first, first + nlocal) live in this process
    bool copy_transport = false;
    std::vector<int> dev;                       // device of each LOCAL rank
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> ev;                 // copy transport: "the send buffers of local rank i are ready"
    std::vector<hipEvent_t> ev_done;            // copy transport: "local rank i has finished reading its peers' send buffers"
};

// copy transport, end of a call: the peers' send buffers may be overwritten by whatever their streams run next (the next
// call's k_slab_pack, the caller's own kernels), so every stream waits until every reader has consumed them.
static int copy_transport_fence(mdtile_shard* sh, const mdtile_stream_t* streams) {
    for (int i = 0; i < sh->nlocal; ++i) {
        MDT_HIP(hipSetDevice(sh->dev[i]));
        MDT_HIP(hipEventRecord(sh->ev_done[i], streams ? as_stream(streams[i]) : sh->stream[i]));
    }
    for (int j = 0; j < sh->nlocal; ++j) {
        MDT_HIP(hipSetDevice(sh->dev[j]));
        hipStream_t st = streams ? as_stream(streams[j]) : sh->stream[j];
        for (int i = 0; i < sh->nlocal; ++i)
            if (i != j) MDT_HIP(hipStreamWaitEvent(st, sh->ev_done[i], 0));
    }
    return MDTILE_OK;
}

static void shard_free(mdtile_shard* sh) {
    if (!sh) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int i = 0; i < (int)sh->dev.size(); ++i) {
        (void)hipSetDevice(sh->dev[i]);
        if (i < (int)sh->comm.size() && sh->comm[i] && rccl()) rccl()->CommDestroy(sh->comm[i]);
        if (i < (int)sh->stream.size() && sh->stream[i]) (void)hipStreamDestroy(sh->stream[i]);
        if (i < (int)sh->ev.size() && sh->ev[i]) (void)hipEventDestroy(sh->ev[i]);
        if (i < (int)sh->ev_done.size() && sh->ev_done[i]) (void)hipEventDestroy(sh->ev_done[i]);
    }
    (void)hipSetDevice(cur);
    delete sh;
}

static bool make_streams(mdtile_shard* sh) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    sh->stream.assign(sh->nlocal, nullptr);
    sh->ev.assign(sh->nlocal, nullptr);
    sh->ev_done.assign(sh->nlocal, nullptr);
    for (int i = 0; i < sh->nlocal; ++i) {
        if (hipSetDevice(sh->dev[i]) != hipSuccess || hipStreamCreateWithFlags(&sh->stream[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sh->ev[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sh->ev_done[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipSetDevice(cur);
            return false;
        }
    }
    (void)hipSetDevice(cur);
    return true;
}

extern "C" mdtile_shard* mdtile_shard_init(int ndev, const int* dev_ids) {
    if (ndev <= 0 || ndev > 64 || !dev_ids) {
        set_error("mdtile_shard_init: bad arguments ndev=%d", ndev);
        return nullptr;
    }
    auto* sh = new mdtile_shard;
    sh->nranks = sh->nlocal = ndev;
    sh->first = 0;
    sh->dev.assign(dev_ids, dev_ids + ndev);
    bool repeats = false;
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j) repeats |= dev_ids[i] == dev_ids[j];
    const char* env = getenv("MDTILE_SHARD_TRANSPORT");
    // one listed device needs no transport at all; MDTILE_SHARD_TRANSPORT=rccl still builds a 1-rank communicator (bring-up checks)
    const bool force_rccl = env && strcmp(env, "rccl") == 0;
    sh->copy_transport = repeats || (ndev == 1 && !force_rccl) || (env && strcmp(env, "copy") == 0);
    if (!make_streams(sh)) {
        set_error("mdtile_shard_init: stream / event creation failed");
        shard_free(sh);
        return nullptr;
    }
    if (!sh->copy_transport) {
        Rccl* R = rccl();
        if (!R) {
            set_error("mdtile_shard_init: librccl.so not found (dlopen): %s", dlerror());
            shard_free(sh);
            return nullptr;
        }
        sh->comm.assign(ndev, nullptr);
        ncclResult_t r = R->CommInitAll(sh->comm.data(), ndev, dev_ids);
        if (r != ncclSuccess) {
            set_error("ncclCommInitAll failed: %s", R->GetErrorString(r));
            shard_free(sh);
            return nullptr;
        }
    }
    return sh;
}

extern "C" int mdtile_shard_unique_id(void* id128) {
    MDT_CHECK_ARG(id128, "mdtile_shard_unique_id: null argument");
    Rccl* R = rccl();
    MDT_CHECK_ARG(R, "mdtile_shard_unique_id: librccl.so not found");
    ncclUniqueId id;
    MDT_NCCL(R->GetUniqueId(&id));
    memcpy(id128, &id, NCCL_UNIQUE_ID_BYTES);
    return MDTILE_OK;
}

extern "C" mdtile_shard* mdtile_shard_init_rank(int nranks, int rank, const void* id128, int device) {
    Rccl* R = rccl();
    if (nranks <= 0 || rank < 0 || rank >= nranks || !id128 || !R) {
        set_error("mdtile_shard_init_rank: bad arguments (nranks=%d rank=%d) or librccl.so missing", nranks, rank);
        return nullptr;
    }
    auto* sh = new mdtile_shard;
    sh->nranks = nranks;
    sh->nlocal = 1;
    sh->first = rank;
    sh->dev.assign(1, device);
    if (!make_streams(sh)) {
        set_error("mdtile_shard_init_rank: stream / event creation failed");
        shard_free(sh);
        return nullptr;
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device);
    ncclUniqueId id;
    memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
    sh->comm.assign(1, nullptr);
    ncclResult_t r = R->CommInitRank(&sh->comm[0], nranks, id, rank);
    (void)hipSetDevice(cur);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank failed: %s", R->GetErrorString(r));
        shard_free(sh);
        return nullptr;
    }
    return sh;
}

// Interruptible bring-up probe: the rendezvous of ncclCommInitRank blocks inside RCCL for as long as a peer is missing or a link does not come
// up, and nothing can reach a communicator that does not exist yet.  The probe makes the SAME rendezvous on a NON-BLOCKING communicator
// (ncclCommInitRankConfig, blocking = 0), polls ncclCommGetAsyncError, and on the deadline calls ncclCommAbort -- the calling thread is
// never left inside RCCL.  A probe that succeeds is thrown away again (the data plane keeps its blocking communicator: no ncclInProgress
// handling in every collective); it needs an id of its own.  Returns MDTILE_OK, or MDTILE_E_HIP with the reason (timeout included).
extern "C" int mdtile_shard_probe_rank(int nranks, int rank, const void* id128, int device, double timeout_s) {
    Rccl* R = rccl();
    MDT_CHECK_ARG(nranks > 0 && rank >= 0 && rank < nranks && id128 && timeout_s > 0.0, "mdtile_shard_probe_rank: bad arguments (nranks=%d rank=%d)", nranks, rank);
    MDT_CHECK_ARG(R, "mdtile_shard_probe_rank: librccl.so not found");
    if (!R->CommInitRankConfig || !R->CommGetAsyncError || !R->CommAbort) {
        set_error("mdtile_shard_probe_rank: this librccl has no ncclCommInitRankConfig / ncclCommGetAsyncError / ncclCommAbort");
        return MDTILE_E_HIP;
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    MDT_HIP(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.blocking = 0;
    ncclComm_t comm = nullptr;
    ncclResult_t r = R->CommInitRankConfig(&comm, nranks, id, rank, &cfg);
    const auto t0 = std::chrono::steady_clock::now();
    bool timed_out = false;
    auto late = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
    // OK is reported only after ncclCommGetAsyncError has said ncclSuccess on a NON-NULL communicator; "in progress" without a
    // communicator handle cannot be polled and counts as a failure
    auto settle = [&](ncclResult_t first) -> ncclResult_t {
        if (first != ncclInProgress && first != ncclSuccess) return first;
        if (!comm) return ncclInternalError;
        while (true) {
            ncclResult_t st = ncclSuccess;
            if (R->CommGetAsyncError(comm, &st) != ncclSuccess) return ncclInternalError;
            if (st != ncclInProgress) return st;
            if (late()) { timed_out = true; return ncclInProgress; }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    };
    r = settle(r);
    // Handshake before the probe communicator is thrown away: one 4-byte all-reduce completes on this rank only when EVERY rank has
    // finished its own bring-up and joined -- without it a rank that comes up early would abort the communicator while slower peers
    // are still polling their init on it.
    hipStream_t hs = nullptr;
    int* d_one = nullptr;
    if (r == ncclSuccess && !timed_out) {
        if (hipStreamCreateWithFlags(&hs, hipStreamNonBlocking) != hipSuccess || hipMalloc(&d_one, sizeof(int)) != hipSuccess ||
            hipMemsetAsync(d_one, 0, sizeof(int), hs) != hipSuccess) {
            r = ncclInternalError;
        } else {
            r = settle(R->AllReduce(d_one, d_one, 1, ncclInt32, ncclSum, comm, hs));
            while (r == ncclSuccess && !timed_out) {
                const hipError_t q = hipStreamQuery(hs);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) { r = ncclInternalError; break; }
                if (late()) { timed_out = true; break; }
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
    }
    int rc = MDTILE_OK;
    if (timed_out || r != ncclSuccess) {
        set_error(timed_out ? "mdtile_shard_probe_rank: the communicator of %d ranks did not come up within %.1f s (rank %d) -- aborted"
                            : "mdtile_shard_probe_rank: communicator bring-up failed (nranks %d, %.1f s budget, rank %d)", nranks, timeout_s, rank);
        rc = MDTILE_E_HIP;
    }
    if (comm) (void)R->CommAbort(comm);      // success or not: the probe communicator is not kept (abort also cancels a handshake still in flight)
    if (d_one) (void)hipFree(d_one);
    if (hs) (void)hipStreamDestroy(hs);
    (void)hipSetDevice(cur);
    return rc;
}

extern "C" void mdtile_shard_destroy(mdtile_shard* sh) { shard_free(sh); }

extern "C" int mdtile_shard_info(const mdtile_shard* sh, int* info4) {
    MDT_CHECK_ARG(sh && info4, "mdtile_shard_info: null argument");
    info4[0] = sh->nranks;
    info4[1] = sh->nlocal;
    info4[2] = sh->first;
    info4[3] = sh->copy_transport ? 0 : 1;
    return MDTILE_OK;
}

extern "C" mdtile_stream_t mdtile_shard_stream(const mdtile_shard* sh, int local_rank) {
    if (!sh || local_rank < 0 || local_rank >= sh->nlocal) return nullptr;
    return reinterpret_cast<mdtile_stream_t>(sh->stream[local_rank]);
}

// Canvas rows two bands share; ascending peer order.  band_rows[2 r], band_rows[2 r + 1] = rows [lo, hi) rank r's tiles touch.
static int halos_of(const int* band_rows, int nranks, int me, HaloSet& S) {
    S.n = 0;
    const int mlo = band_rows[2 * me], mhi = band_rows[2 * me + 1];
    if (mhi <= mlo) return MDTILE_OK;
    for (int r = 0; r < nranks; ++r) {
        if (r == me || band_rows[2 * r + 1] <= band_rows[2 * r]) continue;
        const int lo = mlo > band_rows[2 * r] ? mlo : band_rows[2 * r];
        const int hi = mhi < band_rows[2 * r + 1] ? mhi : band_rows[2 * r + 1];
        if (lo >= hi) continue;
        if (S.n == MAX_HALO_PEERS) {
            set_error("mdtile_halo_exchange: rank %d shares rows with more than %d bands", me, MAX_HALO_PEERS);
            return MDTILE_E_LIMIT;
        }
        S.peer[S.n] = r; S.lo[S.n] = lo; S.hi[S.n] = hi; S.recv[S.n] = nullptr;
        ++S.n;
    }
    return MDTILE_OK;
}

extern "C" size_t mdtile_halo_scratch_bytes(int nranks, int rank, const int* band_rows, int N, int C, int W) {
    HaloSet S;
    if (!band_rows || rank < 0 || rank >= nranks || halos_of(band_rows, nranks, rank, S) != MDTILE_OK) return 0;
    size_t floats = 0;
    for (int k = 0; k < S.n; ++k) floats += (size_t)(S.hi[k] - S.lo[k]) * W * N * C;
    return 2 * floats * sizeof(float) + 256;    // send + receive copies
}

extern "C" int mdtile_halo_exchange(mdtile_shard* sh, float* const* d_partial, void* const* d_scratch, int N, int C, int H, int W,
                                    const int* band_rows, const mdtile_stream_t* streams) {
    MDT_CHECK_ARG(sh && d_partial && d_scratch && band_rows, "mdtile_halo_exchange: null argument");
    MDT_CHECK_ARG(N > 0 && C > 0 && N * C <= 65535 && H > 0 && W > 0, "mdtile_halo_exchange: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
    const int planes = N * C;
    int cur = 0;
    MDT_HIP(hipGetDevice(&cur));
    std::vector<HaloSet> sets(sh->nlocal);
    std::vector<std::vector<float*>> sendp(sh->nlocal), recvp(sh->nlocal);
    // 1. pack
    for (int i = 0; i < sh->nlocal; ++i) {
        const int me = sh->first + i;
        HaloSet& S = sets[i];
        int rc = halos_of(band_rows, sh->nranks, me, S);
        if (rc != MDTILE_OK) return rc;
        hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
        MDT_HIP(hipSetDevice(sh->dev[i]));
        float* base = reinterpret_cast<float*>(d_scratch[i]);
        size_t off = 0;
        for (int k = 0; k < S.n; ++k) {
            const size_t n = (size_t)(S.hi[k] - S.lo[k]) * W * planes;
            sendp[i].push_back(base + off);
            recvp[i].push_back(base + off + n);
            S.recv[k] = base + off + n;
            off += 2 * n;
            hipLaunchKernelGGL(k_slab_pack, dim3(cdiv((long long)(S.hi[k] - S.lo[k]) * W, 256), planes), dim3(256), 0, st, d_partial[i],
                               sendp[i][k], H, W, S.lo[k], S.hi[k] - S.lo[k]);
        }
        if (sh->copy_transport) MDT_HIP(hipEventRecord(sh->ev[i], st));
    }
    MDT_LAUNCH_CHECK();
    // 2. swap
    if (sh->copy_transport) {
        for (int i = 0; i < sh->nlocal; ++i) {
            hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
            MDT_HIP(hipSetDevice(sh->dev[i]));
            for (int k = 0; k < sets[i].n; ++k) {
                const int j = sets[i].peer[k] - sh->first;          // the peer is local: single-process context
                MDT_CHECK_ARG(j >= 0 && j < sh->nlocal, "mdtile_halo_exchange: the copy transport needs every rank in this process");
                int kk = -1;
                for (int q = 0; q < sets[j].n; ++q)
                    if (sets[j].peer[q] == sh->first + i) kk = q;
                MDT_CHECK_ARG(kk >= 0, "mdtile_halo_exchange: asymmetric band table");
                const size_t bytes = (size_t)(sets[i].hi[k] - sets[i].lo[k]) * W * planes * sizeof(float);
                MDT_HIP(hipStreamWaitEvent(st, sh->ev[j], 0));
                MDT_HIP(hipMemcpyAsync(recvp[i][k], sendp[j][kk], bytes, hipMemcpyDeviceToDevice, st));
            }
        }
        int rc = copy_transport_fence(sh, streams);
        if (rc != MDTILE_OK) return rc;
    } else {
        GroupScope G(rccl());
        for (int i = 0; i < sh->nlocal; ++i) {
            hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
            for (int k = 0; k < sets[i].n; ++k) {
                const size_t n = (size_t)(sets[i].hi[k] - sets[i].lo[k]) * W * planes;
                G.note(G.R->Send(sendp[i][k], n, ncclFloat32, sets[i].peer[k], sh->comm[i], st), "ncclSend");
                G.note(G.R->Recv(recvp[i][k], n, ncclFloat32, sets[i].peer[k], sh->comm[i], st), "ncclRecv");
            }
        }
        int rc = G.end("mdtile_halo_exchange");
        if (rc != MDTILE_OK) return rc;
    }
    // 3. sum in ascending rank order
    for (int i = 0; i < sh->nlocal; ++i) {
        const HaloSet& S = sets[i];
        if (S.n == 0) continue;
        hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
        MDT_HIP(hipSetDevice(sh->dev[i]));
        int lo = S.lo[0], hi = S.hi[0];
        for (int k = 1; k < S.n; ++k) {
            lo = lo < S.lo[k] ? lo : S.lo[k];
            hi = hi > S.hi[k] ? hi : S.hi[k];
        }
        hipLaunchKernelGGL(k_halo_add, dim3(cdiv((long long)(hi - lo) * W, 256), planes), dim3(256), 0, st, d_partial[i], H, W, lo, hi - lo,
                           sh->first + i, S);
    }
    MDT_LAUNCH_CHECK();
    MDT_HIP(hipSetDevice(cur));
    return MDTILE_OK;
}

// all-reduce(sum) of `count` doubles per local rank (slow-mode GroupNorm barrier: [sum px*mean, sum px*var, sum px]; the
// sequence-parallel estimator's fp64 (sum, sum of squares) pairs).  In place.
extern "C" int mdtile_allreduce_stats(mdtile_shard* sh, double* const* d_buf, int count, const mdtile_stream_t* streams) {
    MDT_CHECK_ARG(sh && d_buf && count > 0, "mdtile_allreduce_stats: bad arguments");
    if (sh->nranks == 1 && sh->copy_transport) return MDTILE_OK;
    if (sh->copy_transport) {
        // single process, repeated devices: sum on local rank 0's stream in rank order, then copy back
        int cur = 0;
        MDT_HIP(hipGetDevice(&cur));
        std::vector<double> host((size_t)count * sh->nlocal);
        for (int i = 0; i < sh->nlocal; ++i) {
            hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
            MDT_HIP(hipSetDevice(sh->dev[i]));
            MDT_HIP(hipMemcpyAsync(host.data() + (size_t)i * count, d_buf[i], count * sizeof(double), hipMemcpyDeviceToHost, st));
            MDT_HIP(hipStreamSynchronize(st));
        }
        for (int c = 0; c < count; ++c) {
            double a = 0.0;
            for (int i = 0; i < sh->nlocal; ++i) a += host[(size_t)i * count + c];
            host[c] = a;
        }
        for (int i = 0; i < sh->nlocal; ++i) {
            hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
            MDT_HIP(hipSetDevice(sh->dev[i]));
            MDT_HIP(hipMemcpyAsync(d_buf[i], host.data(), count * sizeof(double), hipMemcpyHostToDevice, st));
            MDT_HIP(hipStreamSynchronize(st));
        }
        MDT_HIP(hipSetDevice(cur));
        return MDTILE_OK;
    }
    GroupScope G(rccl());
    for (int i = 0; i < sh->nlocal; ++i)
        G.note(G.R->AllReduce(d_buf[i], d_buf[i], count, ncclFloat64, ncclSum, sh->comm[i], streams ? as_stream(streams[i]) : sh->stream[i]), "ncclAllReduce");
    return G.end("mdtile_allreduce_stats");
}

// broadcast `bytes` from rank `root` to every rank (a region's model output to the bands that composite it, cfg5)
extern "C" int mdtile_shard_bcast(mdtile_shard* sh, void* const* d_buf, size_t bytes, int root, const mdtile_stream_t* streams) {
    MDT_CHECK_ARG(sh && d_buf && root >= 0 && root < sh->nranks, "mdtile_shard_bcast: bad arguments");
    if ((sh->nranks == 1 && sh->copy_transport) || bytes == 0) return MDTILE_OK;
    if (sh->copy_transport) {
        const int jr = root - sh->first;
        MDT_CHECK_ARG(jr >= 0 && jr < sh->nlocal, "mdtile_shard_bcast: the copy transport needs every rank in this process");
        int cur = 0;
        MDT_HIP(hipGetDevice(&cur));
        hipStream_t sr = streams ? as_stream(streams[jr]) : sh->stream[jr];
        MDT_HIP(hipSetDevice(sh->dev[jr]));
        MDT_HIP(hipEventRecord(sh->ev[jr], sr));
        for (int i = 0; i < sh->nlocal; ++i) {
            if (i == jr) continue;
            hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
            MDT_HIP(hipSetDevice(sh->dev[i]));
            MDT_HIP(hipStreamWaitEvent(st, sh->ev[jr], 0));
            MDT_HIP(hipMemcpyAsync(d_buf[i], d_buf[jr], bytes, hipMemcpyDeviceToDevice, st));
        }
        int rc = copy_transport_fence(sh, streams);
        MDT_HIP(hipSetDevice(cur));
        return rc;
    }
    GroupScope G(rccl());
    for (int i = 0; i < sh->nlocal; ++i)
        G.note(G.R->Broadcast(d_buf[i], d_buf[i], bytes, ncclUint8, root, sh->comm[i], streams ? as_stream(streams[i]) : sh->stream[i]), "ncclBroadcast");
    return G.end("mdtile_shard_bcast");
}

// Grouped point-to-point: ops[i][0 .. n_ops[i]) belong to LOCAL rank i; every op may send and / or receive (bytes == 0 skips that
// half).  The k-th send of rank a to rank b pairs with the k-th receive of rank b from rank a.  One ncclGroup for the whole call
// (both halo directions of a row band, every tile rectangle of an image gather ... travel concurrently over their own xGMI links).
extern "C" int mdtile_shard_p2p(mdtile_shard* sh, const mdtile_p2p* const* ops, const int* n_ops, const mdtile_stream_t* streams) {
    MDT_CHECK_ARG(sh && ops && n_ops, "mdtile_shard_p2p: null argument");
    for (int i = 0; i < sh->nlocal; ++i)
        for (int k = 0; k < n_ops[i]; ++k) {
            const mdtile_p2p& o = ops[i][k];
            MDT_CHECK_ARG(o.peer >= 0 && o.peer < sh->nranks && (o.send_bytes == 0 || o.send) && (o.recv_bytes == 0 || o.recv),
                          "mdtile_shard_p2p: bad op %d of local rank %d (peer %d)", k, i, o.peer);
        }
    if (sh->copy_transport) {
        int cur = 0;
        MDT_HIP(hipGetDevice(&cur));
        for (int i = 0; i < sh->nlocal; ++i) {
            MDT_HIP(hipSetDevice(sh->dev[i]));
            MDT_HIP(hipEventRecord(sh->ev[i], streams ? as_stream(streams[i]) : sh->stream[i]));
        }
        for (int i = 0; i < sh->nlocal; ++i) {
            hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
            MDT_HIP(hipSetDevice(sh->dev[i]));
            for (int k = 0; k < n_ops[i]; ++k) {
                const mdtile_p2p& o = ops[i][k];
                if (o.recv_bytes == 0) continue;
                const int j = o.peer - sh->first;
                MDT_CHECK_ARG(j >= 0 && j < sh->nlocal, "mdtile_shard_p2p: the copy transport needs every rank in this process");
                int ord = 0;                                      // this is my ord-th receive from that peer ...
                for (int q = 0; q < k; ++q) ord += ops[i][q].peer == o.peer && ops[i][q].recv_bytes > 0;
                const mdtile_p2p* src = nullptr;                  // ... it pairs with the peer's ord-th send to me
                for (int q = 0, seen = 0; q < n_ops[j] && !src; ++q)
                    if (ops[j][q].peer == sh->first + i && ops[j][q].send_bytes > 0 && seen++ == ord) src = &ops[j][q];
                MDT_CHECK_ARG(src && src->send_bytes == o.recv_bytes, "mdtile_shard_p2p: receive %d of rank %d has no matching send", k, sh->first + i);
                MDT_HIP(hipStreamWaitEvent(st, sh->ev[j], 0));
                MDT_HIP(hipMemcpyAsync(o.recv, src->send, o.recv_bytes, hipMemcpyDeviceToDevice, st));
            }
        }
        int rc = copy_transport_fence(sh, streams);
        MDT_HIP(hipSetDevice(cur));
        return rc;
    }
    GroupScope G(rccl());
    for (int i = 0; i < sh->nlocal; ++i) {
        hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
        for (int k = 0; k < n_ops[i]; ++k) {
            const mdtile_p2p& o = ops[i][k];
            if (o.send_bytes) G.note(G.R->Send(o.send, o.send_bytes, ncclUint8, o.peer, sh->comm[i], st), "ncclSend");
            if (o.recv_bytes) G.note(G.R->Recv(o.recv, o.recv_bytes, ncclUint8, o.peer, sh->comm[i], st), "ncclRecv");
        }
    }
    return G.end("mdtile_shard_p2p");
}

// all-gather: every rank contributes `bytes`; d_recv[i] (nranks * bytes) receives them in rank order.
extern "C" int mdtile_shard_allgather(mdtile_shard* sh, const void* const* d_send, void* const* d_recv, size_t bytes, const mdtile_stream_t* streams) {
    MDT_CHECK_ARG(sh && d_send && d_recv, "mdtile_shard_allgather: null argument");
    if (bytes == 0) return MDTILE_OK;
    if (sh->copy_transport) {
        MDT_CHECK_ARG(sh->nlocal == sh->nranks, "mdtile_shard_allgather: the copy transport needs every rank in this process");
        int cur = 0;
        MDT_HIP(hipGetDevice(&cur));
        for (int i = 0; i < sh->nlocal; ++i) {
            MDT_HIP(hipSetDevice(sh->dev[i]));
            MDT_HIP(hipEventRecord(sh->ev[i], streams ? as_stream(streams[i]) : sh->stream[i]));
        }
        for (int i = 0; i < sh->nlocal; ++i) {
            hipStream_t st = streams ? as_stream(streams[i]) : sh->stream[i];
            MDT_HIP(hipSetDevice(sh->dev[i]));
            for (int j = 0; j < sh->nlocal; ++j) {
                if (j != i) MDT_HIP(hipStreamWaitEvent(st, sh->ev[j], 0));
                MDT_HIP(hipMemcpyAsync(static_cast<char*>(d_recv[i]) + (size_t)j * bytes, d_send[j], bytes, hipMemcpyDeviceToDevice, st));
            }
        }
        int rc = copy_transport_fence(sh, streams);
        MDT_HIP(hipSetDevice(cur));
        return rc;
    }
    GroupScope G(rccl());
    for (int i = 0; i < sh->nlocal; ++i)
        G.note(G.R->AllGather(d_send[i], d_recv[i], bytes, ncclUint8, sh->comm[i], streams ? as_stream(streams[i]) : sh->stream[i]), "ncclAllGather");
    return G.end("mdtile_shard_allgather");
}

namespace {
// selfcheck payloads: word k of rank r
__host__ __device__ inline unsigned chk_word(int what, int r, int k) { return 0x9E3779B9u * (unsigned)(what * 131 + r * 17 + 1) + (unsigned)k * 2654435761u; }
__global__ void k_chk_fill(unsigned* p, int what, int r, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) p[k] = chk_word(what, r, k);
}
__global__ void k_chk_fill_f64(double* p, int r, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) p[k] = (double)(r + 1) * 0.5 + (double)k;
}
}  // namespace

// Bring-up check of a context (any rank count, both transports): one all-reduce, one broadcast, one grouped ring send / receive
// (a self send / receive when the context has one rank) and one all-gather on small payloads, results verified on the host.
// d_scratch[i]: >= mdtile_shard_selfcheck_bytes(sh) device bytes per local rank.  Synchronises the streams it uses.
extern "C" size_t mdtile_shard_selfcheck_bytes(const mdtile_shard* sh) { return sh ? (size_t)(4 + sh->nranks) * 1024 : 0; }

extern "C" int mdtile_shard_selfcheck(mdtile_shard* sh, void* const* d_scratch, const mdtile_stream_t* streams) {
    MDT_CHECK_ARG(sh && d_scratch, "mdtile_shard_selfcheck: null argument");
    constexpr int NW = 256;                                // 1 KiB payloads
    const int nl = sh->nlocal, n = sh->nranks;
    int cur = 0;
    MDT_HIP(hipGetDevice(&cur));
    auto st_of = [&](int i) { return streams ? as_stream(streams[i]) : sh->stream[i]; };
    auto at = [&](int i, int kib) { return static_cast<char*>(d_scratch[i]) + (size_t)kib * 1024; };
    std::vector<unsigned> host(NW * (size_t)(n > 1 ? n : 1));
    std::vector<double> hostd(8);
    auto sync_all = [&]() -> int {
        for (int i = 0; i < nl; ++i) {
            MDT_HIP(hipSetDevice(sh->dev[i]));
            MDT_HIP(hipStreamSynchronize(st_of(i)));
        }
        return MDTILE_OK;
    };
    int rc;
    // (1) all-reduce of 8 doubles: word k becomes sum_r ((r + 1) / 2 + k)
    for (int i = 0; i < nl; ++i) {
        MDT_HIP(hipSetDevice(sh->dev[i]));
        hipLaunchKernelGGL(k_chk_fill_f64, dim3(1), dim3(64), 0, st_of(i), reinterpret_cast<double*>(at(i, 0)), sh->first + i, 8);
    }
    {
        std::vector<double*> bufs(nl);
        for (int i = 0; i < nl; ++i) bufs[i] = reinterpret_cast<double*>(at(i, 0));
        if ((rc = mdtile_allreduce_stats(sh, bufs.data(), 8, streams)) != MDTILE_OK) return rc;
    }
    if ((rc = sync_all()) != MDTILE_OK) return rc;
    for (int i = 0; i < nl; ++i) {
        MDT_HIP(hipSetDevice(sh->dev[i]));
        MDT_HIP(hipMemcpy(hostd.data(), at(i, 0), 8 * sizeof(double), hipMemcpyDeviceToHost));
        for (int k = 0; k < 8; ++k) {
            const double want = 0.25 * n * (n + 1) + (double)k * n;
            MDT_CHECK_ARG(hostd[k] == want, "mdtile_shard_selfcheck: all-reduce word %d of rank %d = %.17g, expected %.17g", k, sh->first + i, hostd[k], want);
        }
    }
    // (2) broadcast from the last rank
    for (int i = 0; i < nl; ++i) {
        MDT_HIP(hipSetDevice(sh->dev[i]));
        hipLaunchKernelGGL(k_chk_fill, dim3(1), dim3(NW), 0, st_of(i), reinterpret_cast<unsigned*>(at(i, 1)), 2, sh->first + i, NW);
    }
    {
        std::vector<void*> bufs(nl);
        for (int i = 0; i < nl; ++i) bufs[i] = at(i, 1);
        if ((rc = mdtile_shard_bcast(sh, bufs.data(), NW * 4, n - 1, streams)) != MDTILE_OK) return rc;
    }
    if ((rc = sync_all()) != MDTILE_OK) return rc;
    for (int i = 0; i < nl; ++i) {
        MDT_HIP(hipSetDevice(sh->dev[i]));
        MDT_HIP(hipMemcpy(host.data(), at(i, 1), NW * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < NW; ++k)
            MDT_CHECK_ARG(host[k] == chk_word(2, n - 1, k), "mdtile_shard_selfcheck: broadcast word %d wrong on rank %d", k, sh->first + i);
    }
    // (3) ring: send to rank + 1, receive from rank - 1, one group (one rank: a self send / receive)
    {
        std::vector<std::vector<mdtile_p2p>> mine(nl);
        std::vector<const mdtile_p2p*> ops(nl);
        std::vector<int> cnt(nl);
        for (int i = 0; i < nl; ++i) {
            MDT_HIP(hipSetDevice(sh->dev[i]));
            const int me = sh->first + i, next = (me + 1) % n, prev = (me + n - 1) % n;
            hipLaunchKernelGGL(k_chk_fill, dim3(1), dim3(NW), 0, st_of(i), reinterpret_cast<unsigned*>(at(i, 2)), 3, me, NW);
            MDT_HIP(hipMemsetAsync(at(i, 3), 0, NW * 4, st_of(i)));
            if (next == prev) {         // one or two ranks: the same peer on both sides, one op carries both halves
                mine[i] = {mdtile_p2p{next, at(i, 2), NW * 4, at(i, 3), NW * 4}};
            } else {
                mine[i] = {mdtile_p2p{next, at(i, 2), NW * 4, nullptr, 0}, mdtile_p2p{prev, nullptr, 0, at(i, 3), NW * 4}};
            }
            ops[i] = mine[i].data();
            cnt[i] = (int)mine[i].size();
        }
        if ((rc = mdtile_shard_p2p(sh, ops.data(), cnt.data(), streams)) != MDTILE_OK) return rc;
    }
    if ((rc = sync_all()) != MDTILE_OK) return rc;
    for (int i = 0; i < nl; ++i) {
        MDT_HIP(hipSetDevice(sh->dev[i]));
        MDT_HIP(hipMemcpy(host.data(), at(i, 3), NW * 4, hipMemcpyDeviceToHost));
        const int from = (sh->first + i + n - 1) % n;
        for (int k = 0; k < NW; ++k)
            MDT_CHECK_ARG(host[k] == chk_word(3, from, k), "mdtile_shard_selfcheck: ring word %d wrong on rank %d", k, sh->first + i);
    }
    // (4) all-gather of 1 KiB per rank
    {
        std::vector<const void*> snd(nl);
        std::vector<void*> rcv(nl);
        for (int i = 0; i < nl; ++i) {
            MDT_HIP(hipSetDevice(sh->dev[i]));
            hipLaunchKernelGGL(k_chk_fill, dim3(1), dim3(NW), 0, st_of(i), reinterpret_cast<unsigned*>(at(i, 2)), 4, sh->first + i, NW);
            snd[i] = at(i, 2);
            rcv[i] = at(i, 4);
        }
        if ((rc = mdtile_shard_allgather(sh, snd.data(), rcv.data(), NW * 4, streams)) != MDTILE_OK) return rc;
    }
    if ((rc = sync_all()) != MDTILE_OK) return rc;
    for (int i = 0; i < nl; ++i) {
        MDT_HIP(hipSetDevice(sh->dev[i]));
        MDT_HIP(hipMemcpy(host.data(), at(i, 4), (size_t)n * NW * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < n; ++r)
            for (int k = 0; k < NW; ++k)
                MDT_CHECK_ARG(host[(size_t)r * NW + k] == chk_word(4, r, k), "mdtile_shard_selfcheck: all-gather word %d of rank %d wrong on rank %d", k, r, sh->first + i);
    }
    MDT_HIP(hipSetDevice(cur));
    return MDTILE_OK;
}
