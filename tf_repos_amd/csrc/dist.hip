// Native driver of the row-sharded multi-GPU step.  One process per GPU; tables sharded by row
// (owner = id % world, local row = id / world), dense parameters replicated, synchronous steps.  Replaces the reference's
// asynchronous parameter server (set_dist_env, DeepFM.py:237-282; run_dist.sh) -- there the TF C++ runtime moves variables
// over gRPC; here the whole step, including its collectives, is enqueued from C++ onto HIP streams:
//
//   route stream : ids(t+1) -> de-duplicate -> bucket by owner -> all-gather of the split sizes -> [host reads W*W ints] ->
//                  all-to-all of the distinct local rows -> index of every entry in the reply -> owner groups the requested rows
//   main stream  : owner packs [row | linear weight] records -> all-to-all -> forward + backward (weight gradients on the
//                  engine's side stream) -> per-distinct-id gradients packed -> all-to-all -> owner segment-sum + table optimizer
//   dense stream : flat gradient arena -> all-reduce -> dense optimizer (beside the gradient exchange)
//
// Routing depends only on the ids, so the NEXT batch is routed while the current one trains; the rows themselves are always
// fetched after the previous step's update (synchronous SGD: N ranks == 1 rank on the same global batch).
//
// The collectives go through a small transport table: RCCL (resolved with dlsym from the librccl.so.1 the process already
// has -- the library itself has no link-time dependency on it) with one communicator per stream, or caller-supplied callbacks
// (tests: two ranks sharing one GPU, staged through host memory over gloo).
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: every call goes through dlsym'd pointers

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

namespace {

using namespace dctr;

constexpr int N_CHANNELS = 3;       // 0 = rows/gradients (main stream), 1 = routing, 2 = dense all-reduce

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllToAllv) AllToAllv = nullptr;      // RCCL extension (optional): one call instead of 2*world send/recv calls
};

RcclApi g_rccl;

int load_rccl(const char* path) {
    if (g_rccl.lib != nullptr) return DCTR_OK;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);        // the copy the process already uses (torch's), if any
    if (lib == nullptr && path != nullptr && path[0] != 0) lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (lib == nullptr) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (lib == nullptr) { set_error("cannot load RCCL: %s", dlerror()); return DCTR_ERR_UNSUPPORTED; }
#define DCTR_SYM(field, name)                                                                      \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, #name));                     \
    if (g_rccl.field == nullptr) { set_error("RCCL symbol %s not found", #name); return DCTR_ERR_UNSUPPORTED; }
    DCTR_SYM(GetUniqueId, ncclGetUniqueId);
    DCTR_SYM(CommInitRank, ncclCommInitRank);
    DCTR_SYM(CommDestroy, ncclCommDestroy);
    DCTR_SYM(GroupStart, ncclGroupStart);
    DCTR_SYM(GroupEnd, ncclGroupEnd);
    DCTR_SYM(Send, ncclSend);
    DCTR_SYM(Recv, ncclRecv);
    DCTR_SYM(AllReduce, ncclAllReduce);
    DCTR_SYM(AllGather, ncclAllGather);
    DCTR_SYM(GetErrorString, ncclGetErrorString);
#undef DCTR_SYM
    g_rccl.AllToAllv = reinterpret_cast<decltype(g_rccl.AllToAllv)>(dlsym(lib, "ncclAllToAllv"));
    if (getenv("DCTR_NO_ALLTOALLV") != nullptr) g_rccl.AllToAllv = nullptr;
    g_rccl.lib = lib;
    return DCTR_OK;
}

#define DCTR_NCCL_CHECK(expr)                                                                              \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) {                                                                           \
            set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);      \
            return DCTR_ERR_HIP;                                                                           \
        }                                                                                                  \
    } while (0)

struct RcclCtx {
    int world = 0, rank = 0;
    ncclComm_t comm[N_CHANNELS] = {nullptr, nullptr, nullptr};
};

int rccl_all_gather_i32(void* ctx, int ch, const int32_t* send, int n, int32_t* recv, void* st) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    DCTR_NCCL_CHECK(g_rccl.AllGather(send, recv, (size_t)n, ncclInt32, c->comm[ch], as_stream(st)));
    return DCTR_OK;
}

int rccl_all_to_all(void* ctx, int ch, const void* send, const int64_t* scnt, void* recv, const int64_t* rcnt, int64_t rec,
                    void* st) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    const char* s = static_cast<const char*>(send);
    char* r = static_cast<char*>(recv);
    if (g_rccl.AllToAllv != nullptr) {
        size_t sc[64], sd[64], rc[64], rd[64];
        size_t so = 0, ro = 0;
        for (int p = 0; p < c->world; ++p) {
            sc[p] = (size_t)(scnt[p] * rec); sd[p] = so; so += sc[p];
            rc[p] = (size_t)(rcnt[p] * rec); rd[p] = ro; ro += rc[p];
        }
        DCTR_NCCL_CHECK(g_rccl.AllToAllv(send, sc, sd, recv, rc, rd, ncclUint8, c->comm[ch], as_stream(st)));
        return DCTR_OK;
    }
    DCTR_NCCL_CHECK(g_rccl.GroupStart());
    int64_t so = 0, ro = 0;
    ncclResult_t bad = ncclSuccess;
    for (int p = 0; p < c->world; ++p) {
        // every peer at once: xGMI is point-to-point, so all 7 links of this GPU carry traffic concurrently
        if (scnt[p] > 0) { ncclResult_t e = g_rccl.Send(s + so * rec, (size_t)(scnt[p] * rec), ncclUint8, p, c->comm[ch], as_stream(st)); if (e != ncclSuccess) bad = e; }
        if (rcnt[p] > 0) { ncclResult_t e = g_rccl.Recv(r + ro * rec, (size_t)(rcnt[p] * rec), ncclUint8, p, c->comm[ch], as_stream(st)); if (e != ncclSuccess) bad = e; }
        so += scnt[p]; ro += rcnt[p];
    }
    DCTR_NCCL_CHECK(g_rccl.GroupEnd());
    DCTR_NCCL_CHECK(bad);
    return DCTR_OK;
}

int rccl_all_reduce_f32(void* ctx, int ch, float* buf, int64_t n, void* st) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    DCTR_NCCL_CHECK(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm[ch], as_stream(st)));
    return DCTR_OK;
}

// what one batch's ids resolve to: split sizes of the exchanges, rows this rank serves, index of every entry in the reply
struct RouteState {
    Group* g = nullptr;                 // requester-side grouping over the GLOBAL id space
    int32_t *send_rows = nullptr, *upos = nullptr, *idx = nullptr, *counts = nullptr, *recv_rows = nullptr, *all_counts = nullptr;
    int32_t* h_all_counts = nullptr;    // pinned [world*world]; row s = rank s's send counts
    std::vector<int64_t> scnt, rcnt;
    int64_t n_send = 0, n_recv = 0;
    const int32_t* ids = nullptr;
    int B = 0;
    int phase = 0;                      // 0 = empty, 1 = begun (split sizes in flight), 2 = complete
    hipEvent_t counts_ready = nullptr;  // the split sizes are in h_all_counts
    hipEvent_t ready = nullptr;         // the whole route is complete (recorded on the stream that computed it)
    hipEvent_t start = nullptr;         // recorded on the main stream when the routing of this batch is requested
};

// The routing of the next batch is ENQUEUED by a worker thread of its own (it owns the route stream and communicator 1): that is
// ~20 launches, two collectives and the one host wait of the step taken off the thread that enqueues the step itself, which was
// the bottleneck (0.47 ms of host time per 0.48 ms step).
struct RouteWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false, has_job = false, done = true;
    int which = 0;
    const int32_t* ids = nullptr;
    int B = 0;
    int rc = DCTR_OK;
    std::string err;
    int device = 0;
};

}  // namespace

struct dctr_dist {
    dctr_engine* E = nullptr;
    int world = 1, rank = 0;
    dctr_transport t{};
    RcclCtx* rccl = nullptr;
    RouteState rs[2];
    int parity = 0;
    int pending = -1;                   // index of the prefetched route, -1 = none
    hipStream_t s_route = nullptr, s_dense = nullptr;
    hipEvent_t ev_fb = nullptr, ev_dense = nullptr;
    float *rows_out = nullptr, *rows_back = nullptr, *send_grads = nullptr, *recv_grads = nullptr, *d_loss = nullptr;
    int64_t cap = 0, cap_owner = 0;
    bool overlap = true;
    bool finish_early = true;
    RouteWorker* worker = nullptr;
    // DCTR_DIST_TIMING=1: host microseconds the enqueueing thread spends in each phase of dctr_dist_train_step (printed at destroy)
    bool timing = false;
    double t_phase[6] = {0, 0, 0, 0, 0, 0};
    int64_t t_steps = 0;
};

namespace {

struct PhaseClock {
    dctr_dist* D;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseClock(dctr_dist* d) : D(d) { if (D->timing) t0 = std::chrono::steady_clock::now(); }
    void lap(int i) {
        if (!D->timing) return;
        const auto t1 = std::chrono::steady_clock::now();
        D->t_phase[i] += std::chrono::duration<double, std::micro>(t1 - t0).count();
        t0 = t1;
    }
};

int route_begin(dctr_dist* D, RouteState& r, const int32_t* ids, int B, hipStream_t s) {
    dctr_engine* E = D->E;
    const int W = D->world;
    // (this state's buffers are free: a side-stream route starts behind r.start, recorded on the main stream after the step that
    //  last used the state was enqueued there; an inline route runs on the main stream itself -- no per-step `done` record)
    DCTR_TRY(group_ids(r.g, ids, B, E->F, s));
    DCTR_TRY(dctr_route_unique(reinterpret_cast<dctr_group_t>(r.g), W, r.send_rows, r.upos, r.counts, s));
    DCTR_TRY(D->t.all_gather_i32(D->t.ctx, 1, r.counts, W, r.all_counts, s));
    DCTR_HIP_CHECK(hipMemcpyAsync(r.h_all_counts, r.all_counts, sizeof(int32_t) * W * W, hipMemcpyDeviceToHost, s));
    DCTR_HIP_CHECK(hipEventRecord(r.counts_ready, s));
    r.ids = ids; r.B = B; r.phase = 1;
    return DCTR_OK;
}

int route_finish(dctr_dist* D, RouteState& r, int which, hipStream_t s) {
    dctr_engine* E = D->E;
    const int W = D->world;
    DCTR_HIP_CHECK(hipEventSynchronize(r.counts_ready));                      // the step's only host wait: W*W split sizes
    r.n_send = r.n_recv = 0;
    for (int p = 0; p < W; ++p) {
        r.scnt[p] = r.h_all_counts[D->rank * W + p];
        r.rcnt[p] = r.h_all_counts[p * W + D->rank];
        r.n_send += r.scnt[p]; r.n_recv += r.rcnt[p];
    }
    DCTR_REQUIRE(r.n_send <= D->cap && r.n_recv <= D->cap_owner, "route: split sizes out of range (%lld sent, %lld received)",
                 (long long)r.n_send, (long long)r.n_recv);
    DCTR_TRY(D->t.all_to_all(D->t.ctx, 1, r.send_rows, r.scnt.data(), r.recv_rows, r.rcnt.data(), sizeof(int32_t), s));
    DCTR_TRY(dctr_entry_index(reinterpret_cast<dctr_group_t>(r.g), r.ids, r.B * E->F, r.upos, r.idx, s));
    DCTR_TRY(dctr_table_group_rows(E, which, r.recv_rows, (int)r.n_recv, s));
    DCTR_HIP_CHECK(hipEventRecord(r.ready, s));
    r.phase = 2;
    return DCTR_OK;
}

void route_worker_main(dctr_dist* D) {
    RouteWorker* w = D->worker;
    hipSetDevice(w->device);
    std::unique_lock<std::mutex> lk(w->mu);
    while (true) {
        w->cv.wait(lk, [&] { return w->has_job || w->stop; });
        if (w->stop) return;
        w->has_job = false;
        const int which = w->which;
        const int32_t* ids = w->ids;
        const int B = w->B;
        lk.unlock();
        RouteState& r = D->rs[which];
        int rc = DCTR_OK;
        if (hipStreamWaitEvent(D->s_route, r.start, 0) != hipSuccess) { set_error("route worker: hipStreamWaitEvent failed"); rc = DCTR_ERR_HIP; }
        if (rc == DCTR_OK) rc = route_begin(D, r, ids, B, D->s_route);
        if (rc == DCTR_OK) rc = route_finish(D, r, which, D->s_route);
        lk.lock();
        w->rc = rc;
        if (rc != DCTR_OK) w->err = get_error();
        w->done = true;
        w->cv.notify_all();
    }
}

int wait_worker(dctr_dist* D) {
    RouteWorker* w = D->worker;
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->done; });
    if (w->rc != DCTR_OK) { set_error("%s", w->err.c_str()); const int rc = w->rc; w->rc = DCTR_OK; return rc; }
    return DCTR_OK;
}

// the route of (ids, B) on the main stream: the prefetched one if it matches, else computed inline
int take_route(dctr_dist* D, const int32_t* ids, int B, hipStream_t M, int* which) {
    if (D->pending >= 0) {
        const int w = D->pending;
        RouteState& r = D->rs[w];
        D->pending = -1;
        if (D->worker != nullptr) DCTR_TRY(wait_worker(D));       // everything of this route has been enqueued (normally long ago)
        else if (r.phase == 1) DCTR_TRY(route_finish(D, r, w, D->s_route));
        DCTR_HIP_CHECK(hipStreamWaitEvent(M, r.ready, 0));
        if (r.ids == ids && r.B == B) { *which = w; return DCTR_OK; }
        // a prefetch for some other batch: it has been waited for (its buffers are quiescent); route this one now
    }
    const int w = D->parity;
    D->parity ^= 1;
    DCTR_TRY(route_begin(D, D->rs[w], ids, B, M));
    DCTR_TRY(route_finish(D, D->rs[w], w, M));
    *which = w;
    return DCTR_OK;
}

int prefetch_begin(dctr_dist* D, const int32_t* next_ids, int next_B, hipStream_t M) {
    const int w = D->parity;
    D->parity ^= 1;
    RouteState& r = D->rs[w];
    DCTR_HIP_CHECK(hipEventRecord(r.start, M));          // next_ids were produced before this point of the main stream
    if (D->worker != nullptr) {
        RouteWorker* wk = D->worker;
        {
            std::lock_guard<std::mutex> lk(wk->mu);
            wk->which = w; wk->ids = next_ids; wk->B = next_B;
            wk->has_job = true; wk->done = false;
        }
        wk->cv.notify_all();
    } else {
        DCTR_HIP_CHECK(hipStreamWaitEvent(D->s_route, r.start, 0));
        DCTR_TRY(route_begin(D, r, next_ids, next_B, D->s_route));
    }
    D->pending = w;
    return DCTR_OK;
}

int fetch_and_forward(dctr_dist* D, RouteState& r, const float* vals, const float* labels, int B, bool train, hipStream_t M) {
    dctr_engine* E = D->E;
    const int64_t rec = (int64_t)(E->K + 4) * sizeof(float);
    DCTR_TRY(dctr_table_gather_packed(E, r.recv_rows, (int)r.n_recv, D->rows_out, M));
    DCTR_TRY(D->t.all_to_all(D->t.ctx, 0, D->rows_out, r.rcnt.data(), D->rows_back, r.scnt.data(), rec, M));
    // weight gradients stay un-joined on the engine's side stream: the dense update continues there (train_step)
    return sharded_forward_backward(E, D->rows_back, (int)r.n_send, r.idx, vals, labels, B, B * D->world, train, false, M);
}

int dense_update(dctr_dist* D, hipStream_t s) {
    float* flat = nullptr;
    int64_t n = 0;
    DCTR_TRY(dctr_dense_grads(D->E, &flat, &n, s));
    DCTR_TRY(D->t.all_reduce_f32(D->t.ctx, 2, flat, n, s));
    return dctr_dense_apply(D->E, s);
}

int dist_alloc(dctr_dist* D) {
    dctr_engine* E = D->E;
    const int W = D->world;
    D->cap = (int64_t)E->MB * E->F;
    D->cap_owner = D->cap * W;
    const size_t P = (size_t)E->K + 4;
    for (RouteState& r : D->rs) {
        DCTR_TRY(group_create(E->cfg.feature_size, D->cap, E->K, &r.g));
        DCTR_HIP_CHECK(hipMalloc(&r.send_rows, D->cap * 4));
        DCTR_HIP_CHECK(hipMalloc(&r.upos, D->cap * 4));
        DCTR_HIP_CHECK(hipMalloc(&r.idx, D->cap * 4));
        DCTR_HIP_CHECK(hipMalloc(&r.counts, sizeof(int32_t) * 2 * W));
        DCTR_HIP_CHECK(hipMalloc(&r.recv_rows, D->cap_owner * 4));
        DCTR_HIP_CHECK(hipMalloc(&r.all_counts, sizeof(int32_t) * W * W));
        DCTR_HIP_CHECK(hipHostMalloc(&r.h_all_counts, sizeof(int32_t) * W * W, hipHostMallocDefault));
        r.scnt.assign(W, 0); r.rcnt.assign(W, 0);
        DCTR_HIP_CHECK(hipEventCreateWithFlags(&r.counts_ready, hipEventDisableTiming));
        DCTR_HIP_CHECK(hipEventCreateWithFlags(&r.ready, hipEventDisableTiming));
        DCTR_HIP_CHECK(hipEventCreateWithFlags(&r.start, hipEventDisableTiming));
    }
    DCTR_HIP_CHECK(hipMalloc(&D->rows_out, D->cap_owner * P * 4));
    DCTR_HIP_CHECK(hipMalloc(&D->rows_back, D->cap * P * 4));
    DCTR_HIP_CHECK(hipMalloc(&D->send_grads, D->cap * P * 4));
    DCTR_HIP_CHECK(hipMalloc(&D->recv_grads, D->cap_owner * P * 4));
    DCTR_HIP_CHECK(hipMalloc(&D->d_loss, 4 * sizeof(float)));
    // no streams of its own: the GPU has 4 hardware queues by default and more streams than queues serialise against each
    // other (measured: a 4th/5th stream landed on the main stream's queue; GPU_MAX_HW_QUEUES=8 was 2.3x SLOWER).  The routing
    // runs on the engine's grouping stream (idle in the sharded step), the dense update continues on its wgrad stream.
    D->s_route = E->s_group;
    D->s_dense = E->s_wgrad;
    DCTR_HIP_CHECK(hipEventCreateWithFlags(&D->ev_fb, hipEventDisableTiming));
    DCTR_HIP_CHECK(hipEventCreateWithFlags(&D->ev_dense, hipEventDisableTiming));
    const char* ov = getenv("DCTR_SHARD_OVERLAP");
    D->overlap = !(ov != nullptr && ov[0] == '0');
    const char* fe = getenv("DCTR_SHARD_FINISH_EARLY");
    D->finish_early = !(fe != nullptr && fe[0] == '0');
    D->timing = getenv("DCTR_DIST_TIMING") != nullptr;
    const char* th = getenv("DCTR_SHARD_THREAD");
    if (D->overlap && !(th != nullptr && th[0] == '0')) {
        D->worker = new RouteWorker();
        DCTR_HIP_CHECK(hipGetDevice(&D->worker->device));
    }
    // both owner-side grouping states exist before the first step (creating one lazily would allocate in the middle of a step)
    DCTR_TRY(dctr_table_group_rows(E, 1, nullptr, 0, nullptr));
    DCTR_HIP_CHECK(hipDeviceSynchronize());
    if (D->worker != nullptr) D->worker->th = std::thread(route_worker_main, D);
    return DCTR_OK;
}

}  // namespace

extern "C" {

int dctr_rccl_unique_id(const char* rccl_path, char* id128) {
    DCTR_REQUIRE(id128 != nullptr, "null argument");
    DCTR_TRY(load_rccl(rccl_path));
    ncclUniqueId id;
    DCTR_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == DCTR_RCCL_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return DCTR_OK;
}

int dctr_dist_create(dctr_handle E, int rank, int world, const dctr_transport* t, dctr_dist_t* out) {
    DCTR_REQUIRE(E && t && out && t->all_gather_i32 && t->all_to_all && t->all_reduce_f32, "null argument");
    DCTR_REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, "bad rank/world %d/%d", rank, world);
    DCTR_REQUIRE(E->cfg.shard_world == world && E->cfg.shard_rank == rank, "engine was created for shard %d/%d, not %d/%d",
                 E->cfg.shard_rank, E->cfg.shard_world, rank, world);
    dctr_dist* D = new dctr_dist();
    D->E = E; D->world = world; D->rank = rank; D->t = *t;
    E->owner_lag_opt_in = true;       // (this driver sets E->want_loss on every step: the owner side may run the time-blocked sweep, lag.h)
    // batch_norm: the column sums of every BN layer go through the transport's all-reduce on the main stream's channel
    E->bn_sync.all_reduce = D->t.all_reduce_f32; E->bn_sync.ctx = D->t.ctx; E->bn_sync.world = world;
    int rc = dist_alloc(D);
    if (rc != DCTR_OK) { dctr_dist_destroy(D); return rc; }
    *out = D;
    return DCTR_OK;
}

int dctr_dist_create_rccl(dctr_handle E, int rank, int world, const char* ids, const char* rccl_path, dctr_dist_t* out) {
    DCTR_REQUIRE(E && ids && out, "null argument");
    DCTR_TRY(load_rccl(rccl_path));
    RcclCtx* c = new RcclCtx();
    c->world = world; c->rank = rank;
    for (int ch = 0; ch < N_CHANNELS; ++ch) {
        ncclUniqueId id;
        memcpy(&id, ids + (size_t)ch * DCTR_RCCL_ID_BYTES, sizeof(id));
        ncclResult_t r = g_rccl.CommInitRank(&c->comm[ch], world, id, rank);
        if (r != ncclSuccess) {
            set_error("ncclCommInitRank (channel %d) failed: %s", ch, g_rccl.GetErrorString(r));
            delete c;
            return DCTR_ERR_HIP;
        }
    }
    dctr_transport t{};
    t.ctx = c;
    t.all_gather_i32 = rccl_all_gather_i32;
    t.all_to_all = rccl_all_to_all;
    t.all_reduce_f32 = rccl_all_reduce_f32;
    int rc = dctr_dist_create(E, rank, world, &t, out);
    if (rc != DCTR_OK) { delete c; return rc; }
    (*out)->rccl = c;
    return DCTR_OK;
}

int dctr_dist_destroy(dctr_dist_t D) {
    if (D == nullptr) return DCTR_OK;
    if (D->worker != nullptr) {
        {
            std::unique_lock<std::mutex> lk(D->worker->mu);
            D->worker->cv.wait(lk, [&] { return D->worker->done; });
            D->worker->stop = true;
        }
        D->worker->cv.notify_all();
        if (D->worker->th.joinable()) D->worker->th.join();
        delete D->worker;
        D->worker = nullptr;
    }
    hipDeviceSynchronize();
    if (D->timing && D->t_steps > 0)
        fprintf(stderr, "[dctr_dist rank %d] host us/step over %lld steps: route %.1f | fetch+fwd+bwd %.1f | dense update %.1f | pack+grad exchange %.1f | table apply %.1f\n",
                D->rank, (long long)D->t_steps, D->t_phase[0] / D->t_steps, D->t_phase[1] / D->t_steps, D->t_phase[2] / D->t_steps,
                D->t_phase[3] / D->t_steps, D->t_phase[4] / D->t_steps);
    for (RouteState& r : D->rs) {
        if (r.g) group_destroy(r.g);
        hipFree(r.send_rows); hipFree(r.upos); hipFree(r.idx); hipFree(r.counts); hipFree(r.recv_rows); hipFree(r.all_counts);
        if (r.h_all_counts) hipHostFree(r.h_all_counts);
        if (r.counts_ready) hipEventDestroy(r.counts_ready);
        if (r.ready) hipEventDestroy(r.ready);
        if (r.start) hipEventDestroy(r.start);
    }
    hipFree(D->rows_out); hipFree(D->rows_back); hipFree(D->send_grads); hipFree(D->recv_grads); hipFree(D->d_loss);
    if (D->ev_fb) hipEventDestroy(D->ev_fb);
    if (D->ev_dense) hipEventDestroy(D->ev_dense);
    if (D->rccl) {
        for (int ch = 0; ch < N_CHANNELS; ++ch)
            if (D->rccl->comm[ch]) g_rccl.CommDestroy(D->rccl->comm[ch]);
        delete D->rccl;
    }
    delete D;
    return DCTR_OK;
}

int dctr_dist_train_step(dctr_dist_t D, const int32_t* d_ids, const float* d_vals, const float* d_labels, int B,
                         const int32_t* d_next_ids, int next_B, float* h_loss, void* stream) {
    DCTR_REQUIRE(D && d_ids && d_vals && d_labels, "null argument");
    dctr_engine* E = D->E;
    DCTR_REQUIRE(B > 0 && B <= E->MB && next_B >= 0 && next_B <= E->MB, "batch %d (next %d) outside (0, max_batch=%d]", B, next_B, E->MB);
    hipStream_t M = as_stream(stream);
    int w = 0;
    E->want_loss = h_loss != nullptr;         // (a loss-reporting step sweeps the whole shard: its l2 term needs every row, lag.h)
    PhaseClock clk(D);
    DCTR_TRY(take_route(D, d_ids, B, M, &w));
    RouteState& r = D->rs[w];
    const bool prefetch = D->overlap && d_next_ids != nullptr && next_B > 0;
    // the first half of the next batch's routing is enqueued BEFORE this step's work so that it runs under the MLP GEMMs
    if (prefetch) DCTR_TRY(prefetch_begin(D, d_next_ids, next_B, M));
    clk.lap(0);
    DCTR_TRY(fetch_and_forward(D, r, d_vals, d_labels, B, true, M));
    clk.lap(1);
    // dense side, beside the gradient exchange (the logit gradient already carries 1/global_batch: sum over ranks = mean)
    // (the side stream already holds the weight gradients; it still needs the output-layer / cross-network partials from M)
    hipStream_t sd = D->s_dense;
    DCTR_HIP_CHECK(hipEventRecord(D->ev_fb, M));
    DCTR_HIP_CHECK(hipStreamWaitEvent(sd, D->ev_fb, 0));
    if (!D->overlap) {          // serial variant: everything back on M before going on
        DCTR_HIP_CHECK(hipEventRecord(D->ev_dense, sd));
        DCTR_HIP_CHECK(hipStreamWaitEvent(M, D->ev_dense, 0));
        sd = M;
    }
    DCTR_TRY(dense_update(D, sd));
    if (sd != M) DCTR_HIP_CHECK(hipEventRecord(D->ev_dense, sd));
    clk.lap(2);
    if (prefetch && D->finish_early && D->worker == nullptr) DCTR_TRY(route_finish(D, D->rs[D->pending], D->pending, D->s_route));
    // sparse side: per-distinct-id gradients in send order -> owners -> segment-sum + table optimizer
    const int64_t rec = (int64_t)(E->K + 4) * sizeof(float);
    DCTR_TRY(dctr_sharded_pack_row_grads(E, reinterpret_cast<dctr_group_t>(r.g), B, r.upos, D->send_grads, M));
    DCTR_TRY(D->t.all_to_all(D->t.ctx, 0, D->send_grads, r.scnt.data(), D->recv_grads, r.rcnt.data(), rec, M));
    clk.lap(3);
    DCTR_TRY(dctr_table_apply_packed(E, w, (int)r.n_recv, D->recv_grads, M));
    clk.lap(4);
    if (D->timing) ++D->t_steps;
    if (sd != M) DCTR_HIP_CHECK(hipStreamWaitEvent(M, D->ev_dense, 0));
    if (prefetch && !D->finish_early && D->worker == nullptr) DCTR_TRY(route_finish(D, D->rs[D->pending], D->pending, D->s_route));
    if (h_loss != nullptr) {
        // loss = mean xent over the GLOBAL batch + l2_reg * (l2_loss(tables, all shards) + l2_loss(regularised dense params))
        float sc[4];
        DCTR_TRY(dctr_read_scalars(E, sc, M));
        DCTR_HIP_CHECK(hipMemcpyAsync(D->d_loss, sc, 3 * sizeof(float), hipMemcpyHostToDevice, M));
        DCTR_TRY(D->t.all_reduce_f32(D->t.ctx, 0, D->d_loss, 3, M));
        float tot[3];
        DCTR_HIP_CHECK(hipMemcpyAsync(tot, D->d_loss, sizeof(tot), hipMemcpyDeviceToHost, M));
        DCTR_HIP_CHECK(hipStreamSynchronize(M));
        *h_loss = tot[0] / (float)((int64_t)B * D->world) + E->cfg.l2_reg * 0.5f * (tot[1] + tot[2] + sc[3]);
    }
    return DCTR_OK;
}

int dctr_dist_predict(dctr_dist_t D, const int32_t* d_ids, const float* d_vals, int B, float* d_prob, void* stream) {
    DCTR_REQUIRE(D && d_ids && d_vals, "null argument");
    dctr_engine* E = D->E;
    DCTR_REQUIRE(B > 0 && B <= E->MB, "batch %d outside (0, max_batch=%d]", B, E->MB);
    hipStream_t M = as_stream(stream);
    int w = 0;
    DCTR_TRY(take_route(D, d_ids, B, M, &w));
    RouteState& r = D->rs[w];
    DCTR_TRY(fetch_and_forward(D, r, d_vals, nullptr, B, false, M));
    if (d_prob != nullptr) {
        float* p = nullptr;
        DCTR_TRY(dctr_last_outputs(E, &p, nullptr));
        DCTR_HIP_CHECK(hipMemcpyAsync(d_prob, p, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, M));
    }
    return DCTR_OK;
}

}  // extern "C"
