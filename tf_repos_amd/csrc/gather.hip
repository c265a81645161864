// K2: embedding gather + value scale + fused first-order / FM second-order / bi-interaction.
// Replaces tf.nn.embedding_lookup + tf.multiply + tf.reduce_sum/tf.square of
// DeepFM.py:125-135, NFM.py:118-128, PNN.py:129-136, AFM.py:123-130, DCN.py:134-138.
//
// HBM-bound (<= 1 flop/byte).  Mapping: TPE = KQ*FS lanes per example, KQ = K/4 lanes cover one
// table row as float4 (a 16-byte coalesced piece each), FS lane groups split the F fields, so every
// lane keeps ~F/FS independent 16-byte row loads in flight and the per-example reductions over the
// fields are register-local plus log2(FS) cross-lane steps.  No LDS: nothing is reused across lanes.
#include "common.h"
#include "lag.h"

namespace dctr {

// LAG: rows may lag behind the present (lag.h): a gathered row stamped earlier than step t-1 is advanced to t-1 in registers
// (its Adam slots are read for that; nothing is written back -- the touched-rows step of this batch redoes it and stores)
template <int KQ, int FS, int MODE, int U, bool LAG>
__global__ __launch_bounds__(256) void gather_fwd_kernel(
    const float4* __restrict__ emb, const float* __restrict__ lin, int64_t rows, int emb_ld4, int lin_ld,
    const int32_t* __restrict__ ids, const float* __restrict__ vals, int B, int F,
    float* __restrict__ e_out, int e_ld, float* __restrict__ yw_out, float* __restrict__ sum_out,
    float* __restrict__ red_out, int32_t* __restrict__ status, LagView L, int nt) {
    constexpr int TPE = KQ * FS;           // lanes per example (power of two, <= 64)
    constexpr int EPB = 256 / TPE;         // examples per block
    if (nt & 2) __builtin_amdgcn_s_setprio(3);      // A/B knob DCTR_GATHER_PRIO=1 (the step's forward waits for this kernel)
    const int tid = threadIdx.x;
    const int sub = tid % TPE;
    const int kq = sub % KQ;
    const int fs = sub / KQ;
    const int b = blockIdx.x * EPB + tid / TPE;
    const bool live = b < B;

    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);   // sum_f e
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);   // sum_f e^2
    float yw = 0.f;

    if (live) {
        const int32_t* idr = ids + (size_t)b * F;
        const float* vr = vals + (size_t)b * F;
        float4* er = reinterpret_cast<float4*>(e_out + (size_t)b * e_ld);
        for (int f0 = fs; f0 < F; f0 += FS * U) {
            int32_t id[U];
            float v[U];
            float4 r[U];
            float w[U];
            int nlag[U];
            float4 m[LAG ? U : 1], vv[LAG ? U : 1];
            float lm[LAG ? U : 1], lv[LAG ? U : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * FS;
                id[u] = (f < F) ? idr[f] : 0;
                v[u] = (f < F) ? vr[f] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * FS;
                const bool ok = (f < F) && (id[u] >= 0) && ((int64_t)id[u] < rows);
                if ((f < F) && !ok) {       // TF CPU gather: InvalidArgumentError [TF-1.4]
                    atomicExch(&status[1], id[u]);
                    atomicExch(&status[0], 1);
                }
                r[u] = ok ? emb[(size_t)id[u] * emb_ld4 + kq] : make_float4(0.f, 0.f, 0.f, 0.f);
                w[u] = (ok && lin != nullptr && kq == 0) ? lin[(size_t)id[u] * lin_ld] : 0.f;
                if constexpr (LAG) {
                    // the Adam slots are read WITH the row, not after its stamp has come back: the gather is a chain of dependent
                    // random reads, and one stage less is worth more than the bytes of the rows that turn out to be current
                    nlag[u] = ok ? lag_behind(L.state->t - 1, L.ts[id[u]]) : 0;
                    m[u] = ok ? L.s0[(size_t)id[u] * L.ld4 + kq] : make_float4(0.f, 0.f, 0.f, 0.f);
                    vv[u] = ok ? L.s1[(size_t)id[u] * L.ld4 + kq] : make_float4(0.f, 0.f, 0.f, 0.f);
                    lm[u] = lv[u] = 0.f;
                    if (ok && lin != nullptr && kq == 0) { lm[u] = L.l0[(size_t)id[u] * L.lin_ld]; lv[u] = L.l1[(size_t)id[u] * L.lin_ld]; }
                }
            }
            if constexpr (LAG) {
                int nl[U];
#pragma unroll
                for (int u = 0; u < U; ++u) nl[u] = (lin != nullptr && kq == 0) ? nlag[u] : 0;
                const Hyper hh = L.state->hyper;
                const int64_t Tm1 = L.state->t - 1;
                lag_catch_up_rows_lin<U, true>(L.state, hh, L.l2, Tm1, nlag, r, m, vv, nl, w, lm, lv);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * FS;
                if (f < F) {
                    float4 e;
                    e.x = r[u].x * v[u]; e.y = r[u].y * v[u]; e.z = r[u].z * v[u]; e.w = r[u].w * v[u];
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    if (nt & 1) __builtin_nontemporal_store(f4v{e.x, e.y, e.z, e.w}, reinterpret_cast<f4v*>(&er[(size_t)f * KQ + kq]));
                    else er[(size_t)f * KQ + kq] = e;
                    s.x += e.x; s.y += e.y; s.z += e.z; s.w += e.w;
                    q.x += e.x * e.x; q.y += e.y * e.y; q.z += e.z * e.z; q.w += e.w * e.w;
                    yw += w[u] * v[u];
                }
            }
        }
    }
    // reduce over the FS field groups (lanes sub, sub+KQ, ...)
#pragma unroll
    for (int off = KQ; off < TPE; off <<= 1) {
        s.x += __shfl_xor(s.x, off); s.y += __shfl_xor(s.y, off);
        s.z += __shfl_xor(s.z, off); s.w += __shfl_xor(s.w, off);
        if (MODE != DCTR_GATHER_RAW) {
            q.x += __shfl_xor(q.x, off); q.y += __shfl_xor(q.y, off);
            q.z += __shfl_xor(q.z, off); q.w += __shfl_xor(q.w, off);
        }
        yw += __shfl_xor(yw, off);
    }
    float4 h;   // 0.5*(S^2 - Q) per component
    h.x = 0.5f * (s.x * s.x - q.x); h.y = 0.5f * (s.y * s.y - q.y);
    h.z = 0.5f * (s.z * s.z - q.z); h.w = 0.5f * (s.w * s.w - q.w);
    float yv = h.x + h.y + h.z + h.w;
    if (MODE == DCTR_GATHER_FM) {
#pragma unroll
        for (int off = 1; off < KQ; off <<= 1) yv += __shfl_xor(yv, off);
    }
    if (live && fs == 0) {
        if (sum_out != nullptr) reinterpret_cast<float4*>(sum_out + (size_t)b * KQ * 4)[kq] = s;
        if (MODE == DCTR_GATHER_BI) reinterpret_cast<float4*>(red_out + (size_t)b * KQ * 4)[kq] = h;
        if (kq == 0) {
            if (yw_out != nullptr) yw_out[b] = yw;
            if (MODE == DCTR_GATHER_FM) red_out[b] = yv;
        }
    }
}

template <int KQ, int FS, int U = 4>
static int launch_gather(const float* emb, const float* lin, int64_t rows, int emb_ld, int lin_ld, const int32_t* ids,
                         const float* vals, int B, int F, int mode, float* e, int e_ld, float* yw,
                         float* sum, float* red, int32_t* status, hipStream_t st, const LagView* lag) {
    constexpr int EPB = 256 / (KQ * FS);
    dim3 grid(ceil_div(B, EPB)), block(256);
    const float4* emb4 = reinterpret_cast<const float4*>(emb);
    LagView L = lag ? *lag : LagView{};
    if (L.ld4 == 0) L.ld4 = KQ;
    // A/B knob DCTR_GATHER_NT=1: nontemporal stores of e.  Measured SLOWER on the HBM-resident tables it was meant for (K = 32, 32 M rows:
    // 13.25 -> 13.89 us; K = 16, 64 M rows: 11.51 -> 11.92; profiles/r05_gather_k32_variants.txt): off.
    static const int nt = (getenv("DCTR_GATHER_NT") ? (atoi(getenv("DCTR_GATHER_NT")) & 1) : 0) | ((getenv("DCTR_GATHER_PRIO") && getenv("DCTR_GATHER_PRIO")[0] == '1') ? 2 : 0);
#define DCTR_GK(MODE_)                                                                                                                  \
    if (lag) gather_fwd_kernel<KQ, FS, MODE_, U, true><<<grid, block, 0, st>>>(emb4, lin, rows, emb_ld / 4, lin_ld, ids, vals, B, F, e, e_ld, yw, sum, red, status, L, nt); \
    else gather_fwd_kernel<KQ, FS, MODE_, U, false><<<grid, block, 0, st>>>(emb4, lin, rows, emb_ld / 4, lin_ld, ids, vals, B, F, e, e_ld, yw, sum, red, status, L, nt)
    switch (mode) {
        case DCTR_GATHER_RAW: DCTR_GK(DCTR_GATHER_RAW); break;
        case DCTR_GATHER_FM: DCTR_GK(DCTR_GATHER_FM); break;
        case DCTR_GATHER_BI: DCTR_GK(DCTR_GATHER_BI); break;
        default:
            set_error("gather: bad mode %d", mode);
            return DCTR_ERR_INVALID_ARG;
    }
#undef DCTR_GK
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// emb_ld / lin_ld: row strides (floats) of the table being read -- K and 1 for the engine's own tables, K+4 for the packed
// [row | linear weight | pad] buffer of rows received from their owners in the row-sharded path
int embed_gather_strided(const float* emb, int emb_ld, const float* lin, int lin_ld, int64_t rows, const int32_t* ids,
                     const float* vals, int B, int F, int K, int mode, float* e, int e_ld, float* yw,
                     float* sum, float* red, int32_t* status, hipStream_t st, const LagView* lag) {
    DCTR_REQUIRE(lag == nullptr || (lag->ld4 == 0 ? (emb_ld == K && lin_ld == 1) : (emb_ld == 4 * lag->ld4 && lin_ld == lag->lin_ld)),
                 "gather: lagging rows live in the engine's own tables (the Adam slots share the table's row strides)");
    DCTR_REQUIRE(K % 4 == 0 && K >= 4 && K <= 256, "embedding_size must be a multiple of 4 in [4,256], got %d", K);
    DCTR_REQUIRE(emb_ld % 4 == 0 && emb_ld >= K && lin_ld >= 1, "gather: bad table strides emb_ld=%d lin_ld=%d", emb_ld, lin_ld);
    DCTR_REQUIRE(e_ld % 4 == 0 && e_ld >= F * K, "gather: e_ld=%d must be a multiple of 4 and >= F*K=%d", e_ld, F * K);
    DCTR_REQUIRE(status != nullptr, "gather: status word required");
    DCTR_REQUIRE(mode == DCTR_GATHER_RAW || red != nullptr, "gather: reduction output required for mode %d", mode);
    if (B <= 0) return DCTR_OK;
    // lanes per example = KQ*FS; U = row loads in flight per lane.  FS*U is chosen >= 39 (Criteo's F) with one trip through the
    // field loop where it fits: the gather is a latency-bound random read, so what matters is how many 16-byte row pieces are in
    // flight (measured on c2, cache-resident table: <4,4,4> 6.2 us, <4,8,5> 5.2 us, <4,16,3> 5.7 us)
#define DCTR_G(Q, FS_, U_) return launch_gather<Q, FS_, U_>(emb, lin, rows, emb_ld, lin_ld, ids, vals, B, F, mode, e, e_ld, yw, sum, red, status, st, lag)
    switch (K / 4) {
        case 1:  DCTR_G(1, 16, 3);
        case 2:  DCTR_G(2, 8, 5);
        case 4:  DCTR_G(4, 8, 5);
        case 8: {
            // HBM-resident table (V=1e8, uniform ids, B=8192; tools/gather_variants.py): <8,8,5> 20.0 us (4.28 TB/s), <8,4,5> 20.5,
            // <8,4,10> 20.4
            static const int v = getenv("DCTR_GATHER_K32") ? atoi(getenv("DCTR_GATHER_K32")) : 1;       // A/B knob
            if (v == 0) DCTR_G(8, 4, 5);
            if (v == 2) DCTR_G(8, 4, 10);
            DCTR_G(8, 8, 5);
        }
        case 16: DCTR_G(16, 4, 5);
        case 32: DCTR_G(32, 2, 4);
        case 64: DCTR_G(64, 1, 4);
        default:
            set_error("embedding_size %d unsupported (K/4 must be a power of two <= 64)", K);
            return DCTR_ERR_UNSUPPORTED;
    }
#undef DCTR_G
}

int embed_gather_fwd(const float* emb, const float* lin, int64_t rows, const int32_t* ids, const float* vals, int B, int F, int K,
                     int mode, float* e, int e_ld, float* yw, float* sum, float* red, int32_t* status, hipStream_t st, const LagView* lag) {
    return embed_gather_strided(emb, K, lin, 1, rows, ids, vals, B, F, K, mode, e, e_ld, yw, sum, red, status, st, lag);
}

}  // namespace dctr

extern "C" int dctr_embed_gather_fwd(const float* d_emb, const float* d_lin, int64_t rows,
                                     const int32_t* d_ids, const float* d_vals, int B, int F, int K,
                                     int mode, float* d_e, int e_ld, float* d_yw, float* d_sum,
                                     float* d_red, int32_t* d_status, void* stream) {
    return dctr::embed_gather_fwd(d_emb, d_lin, rows, d_ids, d_vals, B, F, K, mode, d_e, e_ld, d_yw,
                                  d_sum, d_red, d_status, dctr::as_stream(stream), nullptr);
}

extern "C" int dctr_embed_gather_strided(const float* d_emb, int emb_ld, const float* d_lin, int lin_ld, int64_t rows,
                                         const int32_t* d_ids, const float* d_vals, int B, int F, int K, int mode, float* d_e, int e_ld,
                                         float* d_yw, float* d_sum, float* d_red, int32_t* d_status, void* stream) {
    return dctr::embed_gather_strided(d_emb, emb_ld, d_lin, lin_ld, rows, d_ids, d_vals, B, F, K, mode, d_e, e_ld, d_yw, d_sum, d_red,
                                      d_status, dctr::as_stream(stream), nullptr);
}
