// Time-blocked dense-exact table sweep: the background sweep over one block of the table per step, and the flush that brings
// every row to the present.  See lag.h for the scheme; the row arithmetic is opt_update (opt_rules.h), call for call what
// opt_table_untouched_kernel / opt_lin_dense_kernel (dense_ops.hip) apply to an untouched row in the classic sweep.
#include "lag.h"

namespace dctr {

namespace {

__device__ __forceinline__ float wave_sum_l(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// FLUSH = false: rows of block (t mod period) that the batch does not touch (slot word 0) advance to t.
// FLUSH = true : every row advances to t + target_offset; sum theta^2 (at the target) goes to the sharded sums when asked for.
// KQ lanes per row (a float4 piece each, lane kq == 0 also carries the row's linear weight), UNR rows in flight per lane.
// OCC: waves per SIMD the register allocation is held to.  The sweep of the steps runs BESIDE the MLP's products: a 2 x 7 split-precision GEMM wave
// holds 224 registers and two of them fill a SIMD's file but for 64 -- a sweep wave of <= 64 registers (OCC = 8, one row piece per lane) is placed
// beside them, one of 114 (OCC = 1, four row pieces per lane) takes the place of a GEMM wave for as long as it lives, most of which it spends
// waiting for memory (PMC, tools/r06_lag_pmc.sh: SQ_WAIT_ANY 44 % of its wave cycles).
template <int KQ, bool FLUSH, int UNR, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void lag_advance_kernel(int64_t rows, float4* __restrict__ emb, float4* __restrict__ s0, float4* __restrict__ s1,
                                                         float* __restrict__ lin, float* __restrict__ l0, float* __restrict__ l1,
                                                         const int32_t* __restrict__ slot, uint8_t* __restrict__ ts,
                                                         const StepState* __restrict__ S, float l2, int period, int target_offset,
                                                         float* __restrict__ sumsq_emb, float* __restrict__ sumsq_lin, int ld4, int lin_ld) {
    const int64_t T = S->t;
    const Hyper h = S->hyper;
    int64_t r0 = 0, r1 = rows, target = T + target_offset;
    if (!FLUSH) {
        const int64_t rpb = (rows + period - 1) / period;
        r0 = (T % period) * rpb;
        r1 = r0 + rpb < rows ? r0 + rpb : rows;
        target = T;
    }
    const int64_t n_items = (r1 - r0) * KQ;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float sq = 0.f, sql = 0.f;
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t0 < n_items; t0 += stride * UNR) {
        float4 th[UNR], m[UNR], v[UNR];
        float lt[UNR], lm[UNR], lv[UNR];
        int n[UNR];
        int64_t row[UNR];
        bool live[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int64_t t = t0 + j * stride;
            live[j] = t < n_items;
            row[j] = r0 + (live[j] ? t / KQ : 0);
            if (live[j] && !FLUSH) live[j] = slot[row[j]] == 0;       // the batch's rows: stepped by the touched-rows pass, with their gradient
            n[j] = live[j] ? lag_behind(target, ts[row[j]]) : 0;
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int kq = (int)((t0 + j * stride) % KQ);
            // (a flush that reports sum theta^2 reads every row; otherwise rows already at the target are left alone)
            // (what is not loaded is ZERO, not undefined: a lane that is not behind runs the replay's steps with the identity's coefficients,
            //  lag.h lag_replay_rows -- m and v must be finite for theta to come out untouched)
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            th[j] = m[j] = v[j] = z4;
            lt[j] = lm[j] = lv[j] = 0.f;
            if (live[j] && (n[j] > 0 || (FLUSH && sumsq_emb != nullptr))) {
                const size_t i4 = (size_t)row[j] * ld4 + kq;
                th[j] = emb[i4];
                if (n[j] > 0) { m[j] = s0[i4]; v[j] = s1[i4]; }
                if (kq == 0 && lin != nullptr) {
                    const size_t il = (size_t)row[j] * lin_ld;
                    lt[j] = lin[il];
                    if (n[j] > 0) { lm[j] = l0[il]; lv[j] = l1[il]; }
                }
            } else {
                live[j] = false;
            }
        }
        int nn[UNR], nl[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            nn[j] = live[j] ? n[j] : 0;
            nl[j] = (live[j] && lin != nullptr && (int)((t0 + j * stride) % KQ) == 0) ? n[j] : 0;
        }
        lag_catch_up_rows_lin<UNR>(S, h, l2, target, nn, th, m, v, nl, lt, lm, lv);
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            if (!live[j]) continue;
            const int kq = (int)((t0 + j * stride) % KQ);
            const size_t i4 = (size_t)row[j] * ld4 + kq;
            if (n[j] > 0) { emb[i4] = th[j]; s0[i4] = m[j]; s1[i4] = v[j]; }
            sq += th[j].x * th[j].x + th[j].y * th[j].y + th[j].z * th[j].z + th[j].w * th[j].w;
            if (kq == 0) {
                if (lin != nullptr) {
                    const size_t il = (size_t)row[j] * lin_ld;
                    if (n[j] > 0) { lin[il] = lt[j]; l0[il] = lm[j]; l1[il] = lv[j]; }
                    sql += lt[j] * lt[j];
                }
                if (n[j] > 0) ts[row[j]] = (uint8_t)target;
            }
        }
    }
    if (FLUSH && sumsq_emb != nullptr) {
        __shared__ float red[2][4];
        sq = wave_sum_l(sq);
        sql = wave_sum_l(sql);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sq; red[1][threadIdx.x >> 6] = sql; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(sumsq_emb + (blockIdx.x & (SUMSQ_SHARDS - 1)), red[0][0] + red[0][1] + red[0][2] + red[0][3]);
            if (sumsq_lin != nullptr) atomicAdd(sumsq_lin + (blockIdx.x & (SUMSQ_SHARDS - 1)), red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        }
    }
}

// The background sweep, software-pipelined (round 6).  lag_advance_kernel gives every thread ONE trip: all waves of the grid request their
// rows at once, wait out the memory system together (26 MB), replay together, store together (26 MB) -- three phases in lockstep,
// 18-21 us alone on an idle chip for 52 MB of traffic (9 us at the copy rate) and 5.6 us of VALU work (tools/lag_probe.hip).  Here a thread
// walks NCH chunks of UNR row pieces: the rows of chunk c + 1 are requested BEFORE chunk c is replayed and stored, so the replay's
// arithmetic runs under the next chunk's memory latency and the stores drain under the next replay.  Rows are requested unconditionally
// (the batch's own rows, ~10 % of a block, are fetched and dropped: the slot word and the stamp arrive with them instead of a dependent
// round trip ahead of them).  What is computed per row is lag_catch_up_rows_lin, call for call what lag_advance_kernel computes.
template <int KQ, int UNR>
__global__ __launch_bounds__(256) void lag_sweep_pipe_kernel(int64_t rows, float4* __restrict__ emb, float4* __restrict__ s0, float4* __restrict__ s1,
                                                            float* __restrict__ lin, float* __restrict__ l0, float* __restrict__ l1,
                                                            const int32_t* __restrict__ slot, uint8_t* __restrict__ ts,
                                                            const StepState* __restrict__ S, float l2, int period, int ld4, int lin_ld) {
    const int64_t T = S->t;
    const Hyper h = S->hyper;
    const int64_t rpb = (rows + period - 1) / period;
    const int64_t r0 = (T % period) * rpb;
    const int64_t r1 = r0 + rpb < rows ? r0 + rpb : rows;
    const int64_t n_items = (r1 - r0) * KQ;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    struct Chunk {
        float4 th[UNR], m[UNR], v[UNR];
        float lt[UNR], lm[UNR], lv[UNR];
        int32_t sl[UNR];
        uint8_t st[UNR];
        int64_t row[UNR];
        bool in[UNR];
    };
    auto request = [&](Chunk& c, int64_t t0) {
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int64_t t = t0 + j * stride;
            c.in[j] = t < n_items;
            c.row[j] = r0 + (c.in[j] ? t / KQ : 0);
            const int kq = (int)(t % KQ);
            const size_t i4 = (size_t)c.row[j] * ld4 + kq;
            c.sl[j] = slot[c.row[j]];
            c.st[j] = ts[c.row[j]];
            c.th[j] = emb[i4]; c.m[j] = s0[i4]; c.v[j] = s1[i4];
            c.lt[j] = c.lm[j] = c.lv[j] = 0.f;
            if (kq == 0 && lin != nullptr) {
                const size_t il = (size_t)c.row[j] * lin_ld;
                c.lt[j] = lin[il]; c.lm[j] = l0[il]; c.lv[j] = l1[il];
            }
        }
    };
    auto advance = [&](Chunk& c, int64_t t0) {
        int nn[UNR], nl[UNR];
        bool live[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int kq = (int)((t0 + j * stride) % KQ);
            const int n = lag_behind(T, c.st[j]);
            live[j] = c.in[j] && c.sl[j] == 0 && n > 0;           // the batch's rows: stepped by the touched-rows pass, with their gradient
            nn[j] = live[j] ? n : 0;
            nl[j] = (live[j] && lin != nullptr && kq == 0) ? n : 0;
        }
        lag_catch_up_rows_lin<UNR>(S, h, l2, T, nn, c.th, c.m, c.v, nl, c.lt, c.lm, c.lv);
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            if (!live[j]) continue;
            const int kq = (int)((t0 + j * stride) % KQ);
            const size_t i4 = (size_t)c.row[j] * ld4 + kq;
            emb[i4] = c.th[j]; s0[i4] = c.m[j]; s1[i4] = c.v[j];
            if (kq == 0) {
                if (lin != nullptr) {
                    const size_t il = (size_t)c.row[j] * lin_ld;
                    lin[il] = c.lt[j]; l0[il] = c.lm[j]; l1[il] = c.lv[j];
                }
                ts[c.row[j]] = (uint8_t)T;
            }
        }
    };
    Chunk a, b;
    int64_t t0 = tid;
    if (t0 >= n_items) return;
    request(a, t0);
    for (;;) {            // two chunks per trip: the buffers change roles without a copy
        const int64_t t1 = t0 + stride * UNR;
        const bool more1 = t1 < n_items;
        if (more1) request(b, t1);
        advance(a, t0);
        if (!more1) break;
        const int64_t t2 = t1 + stride * UNR;
        const bool more2 = t2 < n_items;
        if (more2) request(a, t2);
        advance(b, t1);
        if (!more2) break;
        t0 = t2;
    }
}

__global__ __launch_bounds__(256) void lag_stamp_kernel(uint8_t* __restrict__ ts, int64_t rows, const StepState* __restrict__ S) {
    const uint8_t v = (uint8_t)S->t;
    const uint32_t v4 = v * 0x01010101u;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= rows) *reinterpret_cast<uint32_t*>(ts + i) = v4;
    else for (int64_t k = i; k < rows; ++k) ts[k] = v;
}

template <bool FLUSH>
int launch_advance(int K, int64_t rows, float* emb, float* s0, float* s1, float* lin, float* l0, float* l1, const int32_t* slot, uint8_t* ts,
                   const StepState* state, float l2, int period, int target_offset, float* sumsq_emb, float* sumsq_lin, hipStream_t st, int ld, int lin_ld) {
    const int KQ = K / 4;
    const int ld4 = ld > 0 ? ld / 4 : KQ;
    const int64_t span = FLUSH ? rows : (rows + period - 1) / period;
    // the sweep runs UNDER the MLP GEMMs like the classic background pass: a small grid (2 blocks per CU); the flush has the chip
    static const int bpc = getenv("DCTR_LAG_BLOCKS_PER_CU") ? atoi(getenv("DCTR_LAG_BLOCKS_PER_CU")) : 2;     // A/B knob
    const int grid = (int)std::min<int64_t>(ceil_div(span * KQ, 256 * 4), FLUSH ? 256 * 8 : 256 * bpc);
    float4 *e4 = reinterpret_cast<float4*>(emb), *a4 = reinterpret_cast<float4*>(s0), *b4 = reinterpret_cast<float4*>(s1);
    // A/B knob DCTR_LAG_PIPE=1: software-pipelined trips (lag_sweep_pipe_kernel) on DCTR_LAG_PIPE_BLOCKS_PER_CU blocks per CU
    static const bool pipe = [] { const char* v = getenv("DCTR_LAG_PIPE"); return v != nullptr && v[0] == '1'; }();
    if (!FLUSH && pipe && KQ <= 16) {
        static const int pbpc = getenv("DCTR_LAG_PIPE_BLOCKS_PER_CU") ? atoi(getenv("DCTR_LAG_PIPE_BLOCKS_PER_CU")) : 1;     // A/B knob
        const int pgrid = (int)std::min<int64_t>(ceil_div(span * KQ, 256 * 2), 256 * pbpc);
        switch (KQ) {
#define DCTR_P(Q) case Q: lag_sweep_pipe_kernel<Q, 2><<<pgrid, 256, 0, st>>>(rows, e4, a4, b4, lin, l0, l1, slot, ts, state, l2, period, ld4, lin_ld); break
            DCTR_P(1); DCTR_P(2); DCTR_P(4); DCTR_P(8); DCTR_P(16);
#undef DCTR_P
        }
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    // A/B knob DCTR_LAG_SMALL=1: one row piece per lane, <= 64 registers, as many blocks as the block of the table needs
    static const bool small = [] { const char* v = getenv("DCTR_LAG_SMALL"); return v != nullptr && v[0] == '1'; }();
    if (!FLUSH && small && KQ <= 16) {
        const int sgrid = ceil_div(span * KQ, 256);
        switch (KQ) {
#define DCTR_S(Q) case Q: lag_advance_kernel<Q, false, 1, 8><<<sgrid, 256, 0, st>>>(rows, e4, a4, b4, lin, l0, l1, slot, ts, state, l2, period, target_offset, sumsq_emb, sumsq_lin, ld4, lin_ld); break
            DCTR_S(1); DCTR_S(2); DCTR_S(4); DCTR_S(8); DCTR_S(16);
#undef DCTR_S
        }
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    switch (KQ) {
#define DCTR_A(Q) case Q: lag_advance_kernel<Q, FLUSH, 4><<<grid, 256, 0, st>>>(rows, e4, a4, b4, lin, l0, l1, slot, ts, state, l2, period, target_offset, sumsq_emb, sumsq_lin, ld4, lin_ld); break
        DCTR_A(1); DCTR_A(2); DCTR_A(4); DCTR_A(8); DCTR_A(16); DCTR_A(32); DCTR_A(64);
#undef DCTR_A
        default: set_error("lag: K=%d unsupported", K); return DCTR_ERR_UNSUPPORTED;
    }
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace

int lag_sweep(int K, int64_t rows, float* emb, float* s0, float* s1, float* lin, float* l0, float* l1, const int32_t* slot, uint8_t* ts,
              const StepState* state, float l2, int period, hipStream_t st, int ld, int lin_ld) {
    DCTR_REQUIRE(period >= 2 && period <= LAG_MAX_PERIOD, "lag_sweep: period %d outside [2, %d]", period, LAG_MAX_PERIOD);
    return launch_advance<false>(K, rows, emb, s0, s1, lin, l0, l1, slot, ts, state, l2, period, 0, nullptr, nullptr, st, ld, lin_ld);
}

int lag_flush(int K, int64_t rows, float* emb, float* s0, float* s1, float* lin, float* l0, float* l1, uint8_t* ts, const StepState* state,
              float l2, int target_offset, float* sumsq_emb, float* sumsq_lin, hipStream_t st, int ld, int lin_ld) {
    return launch_advance<true>(K, rows, emb, s0, s1, lin, l0, l1, nullptr, ts, state, l2, 1, target_offset, sumsq_emb, sumsq_lin, st, ld, lin_ld);
}

int lag_stamp(uint8_t* ts, int64_t rows, const StepState* state, hipStream_t st) {
    lag_stamp_kernel<<<ceil_div(ceil_div(rows, 4), 256), 256, 0, st>>>(ts, rows, state);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

}  // namespace dctr
