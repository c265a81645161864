// K8: sparse table gradient = IndexedSlices -> unsorted_segment_sum of the gather gradients
// (DeepFM.py:126,130 differentiated by optimizer.minimize DeepFM.py:213).
//
// Integer/byte work, HBM/L2-latency bound -- no sort, no GEMM reshaping:
//   group_ids:  a direct-mapped slot word per table row groups the batch's B*F ids:
//     count    slot[id] += multiplicity (wave-aggregated: lanes of a wave holding the same id issue
//              ONE atomic; entries are walked field-major so Criteo's per-field hot ids -- the 13
//              numeric ids hit by every example -- collapse to one atomic per wave)
//     finalize per distinct id u: segment [seg_start[u], +cnt[u]) carved with one atomicAdd,
//              slot[id] = u+1, compact gradient row u zeroed
//     fill     perm[] = entry indices grouped by distinct id, seg_of[] = u per grouped position
//   scatter:    walkers of K/4 lanes stride over runs of R grouped positions, accumulate row
//              gradients in registers and flush once per (run, distinct id) with float atomics into
//              the compact [U, K] buffer.  The FM / bi-interaction backward is fused here so dE never
//              round-trips through HBM for those terms.
#include "common.h"
#include "ops.h"
#include "opt_rules.h"
#include "lag.h"

namespace dctr {

__device__ __forceinline__ float wave_sum_g(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// forget the previous batch: slot words of its distinct ids back to 0; the last block to finish also zeroes the counters
// (ticket in counters[3]) so that no separate one-thread launch is needed
__global__ void group_reset_kernel(int32_t* __restrict__ slot, const int32_t* __restrict__ uniq,
                                   int32_t* __restrict__ counters, int n_max) {
    __shared__ int last;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int U = counters[0];
    if (u < U && u < n_max) slot[uniq[u]] = 0;
    __syncthreads();
    if (threadIdx.x == 0) last = (atomicAdd(&counters[3], 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (last && threadIdx.x == 0) {
        counters[0] = 0;
        counters[1] = 0;
        counters[2] = 0;
        counters[3] = 0;
        counters[4] = 0;
    }
}

// Wave-level grouping of equal ids without memory traffic: for every lane, the lowest lane of the wave holding the
// same id (its "leader"), the number of lanes sharing the id and this lane's rank among them.  The loop runs once
// per DISTINCT id in the wave and only does ballots/shuffles; the atomics are issued afterwards by all leaders at
// once, so their latencies overlap instead of serialising.
// (at most WAVE_GROUP_ROUNDS ids are looked for: a wave of 64 different ids -- most waves of a high-cardinality field -- would
//  otherwise spend 64 rounds finding nothing to merge; lanes still pending after the rounds go on alone, which only costs the
//  odd duplicate an atomic of its own.  The hot ids of a field sit in many lanes, so they come up as a leader early.)
constexpr int WAVE_GROUP_ROUNDS = 12;
struct WaveGroup { int leader; int count; int rank; };
__device__ __forceinline__ WaveGroup wave_group(int id, bool valid, int lane) {
    WaveGroup g{lane, valid ? 1 : 0, 0};
    bool pending = valid;
    for (int round = 0; round < WAVE_GROUP_ROUNDS; ++round) {
        const unsigned long long m = __ballot(pending);
        if (m == 0ull) break;
        const int leader = __ffsll((long long)m) - 1;
        const int lid = __shfl(id, leader);
        const bool same = pending && (id == lid);
        const unsigned long long sm = __ballot(same);
        if (same) {
            g.leader = leader;
            g.count = __popcll(sm);
            g.rank = __popcll(sm & ((1ull << lane) - 1ull));
        }
        pending = pending && !same;
    }
    return g;
}

// entry i <-> (f = i / B, b = i % B): field-major walk
// (blocks of GROUP_BLOCK = 1024 threads: every block ends with ONE atomic on a shared counter, and a single word sustains only
//  ~90 atomics/us -- 624 blocks of 256 spent ~8 us of this kernel and of group_segments_kernel queueing on it)
constexpr int GROUP_BLOCK = 1024;
__global__ __launch_bounds__(GROUP_BLOCK) void group_count_kernel(const int32_t* __restrict__ ids, int B, int F,
                                                         int64_t rows, int32_t* __restrict__ slot,
                                                         int32_t* __restrict__ uniq, int32_t* __restrict__ counters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = B * F;
    const int lane = threadIdx.x & 63;
    int id = -1;
    bool valid = false;
    if (i < n) {
        const int f = i / B, b = i - f * B;
        id = ids[(size_t)b * F + f];
        valid = (id >= 0) && ((int64_t)id < rows);
    }
    const WaveGroup g = wave_group(id, valid, lane);
    bool first = false;     // this wave is the first to touch the id
    if (valid && g.leader == lane) first = atomicAdd(&slot[id], g.count) == 0;
    // compact-slot allocation: ONE atomic on the shared counter per 256-thread block (a single word sustains only ~90
    // atomics/us, so one per wave -- 2500 per batch -- cost more than everything else in this kernel)
    __shared__ int wcount[GROUP_BLOCK / 64];
    __shared__ int bbase;
    const int wave = threadIdx.x >> 6;
    const unsigned long long fm = __ballot(first);
    if (lane == 0) wcount[wave] = __popcll(fm);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < (int)blockDim.x / 64; ++w) tot += wcount[w];
        bbase = tot ? atomicAdd(&counters[0], tot) : 0;
    }
    __syncthreads();
    if (first) {
        int base = bbase;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        uniq[base + __popcll(fm & ((1ull << lane) - 1ull))] = id;
    }
}

// ---- CSR batches (F == 1, ids in example order): a wave of 64 neighbouring entries is one example's multi-hot list -- all
// different -- so the wave-level aggregation above finds nothing to merge and a hot id (the same category in ~10 % of the
// examples) takes one global atomic per occurrence, serialised at ~90/us on its word.  These variants aggregate a CHUNK of 2048
// entries per block in an LDS hash table first: one global atomic per (block, distinct id).
constexpr int HCHUNK = 2048;           // entries per block (8 per thread)
constexpr int HSIZE = 4096;            // LDS hash slots (load factor <= 0.5)
__device__ __forceinline__ int hash_find(int* hkey, int id) {
    int h = (int)(((uint32_t)id * 2654435761u) >> 20);                       // top 12 bits
    while (true) {
        const int prev = atomicCAS(&hkey[h], -1, id);
        if (prev == -1 || prev == id) return h;
        h = (h + 1) & (HSIZE - 1);
    }
}

__global__ __launch_bounds__(256) void group_count_hash_kernel(const int32_t* __restrict__ ids, int n, int64_t rows,
                                                              int32_t* __restrict__ slot, int32_t* __restrict__ uniq,
                                                              int32_t* __restrict__ counters) {
    __shared__ int hkey[HSIZE];
    __shared__ int hcnt[HSIZE];
    __shared__ int wsum[4];
    __shared__ int bbase;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int h = tid; h < HSIZE; h += 256) { hkey[h] = -1; hcnt[h] = 0; }
    __syncthreads();
    const int base = blockIdx.x * HCHUNK;
#pragma unroll
    for (int k = 0; k < HCHUNK / 256; ++k) {
        const int i = base + k * 256 + tid;
        if (i < n) {
            const int id = ids[i];
            if (id >= 0 && (int64_t)id < rows) atomicAdd(&hcnt[hash_find(hkey, id)], 1);
        }
    }
    __syncthreads();
    int nf = 0;                                         // ids this block is the first to touch (hcnt -> -1 marks them)
    for (int h = tid; h < HSIZE; h += 256)
        if (hkey[h] >= 0 && atomicAdd(&slot[hkey[h]], hcnt[h]) == 0) { hcnt[h] = -1; ++nf; }
    int incl = nf;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        bbase = tot ? atomicAdd(&counters[0], tot) : 0;
    }
    __syncthreads();
    int o = bbase + incl - nf;
    for (int w = 0; w < wave; ++w) o += wsum[w];
    for (int h = tid; h < HSIZE; h += 256)
        if (hkey[h] >= 0 && hcnt[h] == -1) uniq[o++] = hkey[h];
}

__global__ __launch_bounds__(256) void group_fill_hash_kernel(const int32_t* __restrict__ ids, int n, int64_t rows,
                                                             const int32_t* __restrict__ slot, int32_t* __restrict__ cursor,
                                                             int32_t* __restrict__ perm, int32_t* __restrict__ seg_of) {
    __shared__ int hkey[HSIZE];                         // id, then (after the reservation) its distinct-id index u
    __shared__ int hcnt[HSIZE];
    __shared__ int hbase[HSIZE];
    const int tid = threadIdx.x;
    for (int h = tid; h < HSIZE; h += 256) { hkey[h] = -1; hcnt[h] = 0; }
    __syncthreads();
    const int base = blockIdx.x * HCHUNK;
    int myh[HCHUNK / 256], myrank[HCHUNK / 256];
#pragma unroll
    for (int k = 0; k < HCHUNK / 256; ++k) {
        const int i = base + k * 256 + tid;
        myh[k] = -1; myrank[k] = 0;
        if (i < n) {
            const int id = ids[i];
            if (id >= 0 && (int64_t)id < rows) { myh[k] = hash_find(hkey, id); myrank[k] = atomicAdd(&hcnt[myh[k]], 1); }
        }
    }
    __syncthreads();
    for (int h = tid; h < HSIZE; h += 256)
        if (hkey[h] >= 0) {
            const int u = slot[hkey[h]] - 1;
            hbase[h] = atomicAdd(&cursor[u], hcnt[h]);  // this block's run inside the id's segment
            hkey[h] = u;
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HCHUNK / 256; ++k)
        if (myh[k] >= 0) {
            const int pos = hbase[myh[k]] + myrank[k];
            perm[pos] = base + k * 256 + tid;
            seg_of[pos] = hkey[myh[k]];
        }
}

// segments at least this long are not walked in runs (their per-run atomic flushes all land on ONE 64-byte compact row: 4096
// entries in runs of 16 = 256 flushes x 17 floats serialised at ~90 atomics/us -- 48 us, the whole scatter); a block reduces
// chunks of them in registers + LDS and flushes once per chunk
constexpr int SHORT_SEGMENT = 8;        // embed_scatter_apply: segments up to this long take one walker (KQ lanes), entries loaded together
// segments from this many entries on are cut into chunks of as many, reduced by a block each and joined by atomics.  A wave
// covers 64/KQ entries per load, so the wave-per-segment path below this bound costs up to bound * KQ / 256 dependent trips of
// four loads: 4 at K = 16, but 64 at the K = 256 AFM runs at (a 128-example batch's hot ids: 124 us of latency) -- hence by width
__host__ __device__ constexpr int long_segment(int KQ) { return KQ >= 64 ? 64 : KQ >= 32 ? 128 : 256; }

// one thread per distinct id: carve its segment of the grouped-entry array (one atomic per block on the running total)
__global__ __launch_bounds__(GROUP_BLOCK) void group_segments_kernel(int32_t* __restrict__ slot, const int32_t* __restrict__ uniq,
                                                            int32_t* __restrict__ cnt, int32_t* __restrict__ seg_start,
                                                            int32_t* __restrict__ cursor, int32_t* __restrict__ counters,
                                                            float* __restrict__ glin, int32_t* __restrict__ long_list, int long_cap,
                                                            int32_t* __restrict__ done, int32_t* __restrict__ medium_list, int medium_cap, int LONG_SEGMENT) {
    __shared__ int wsum[GROUP_BLOCK / 64];
    __shared__ int bbase;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (int)blockDim.x / 64;
    const int U = counters[0];
    int c = 0, id = 0;
    if (u < U) { id = uniq[u]; c = slot[id]; }
    // inclusive scan of c within the wave
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < n_waves; ++w) tot += wsum[w];
        bbase = tot ? atomicAdd(&counters[1], tot) : 0;
    }
    __syncthreads();
    if (u < U) {
        int s = bbase + incl - c;
        for (int w = 0; w < wave; ++w) s += wsum[w];
        cnt[u] = c;
        seg_start[u] = s;
        cursor[u] = s;
        slot[id] = u + 1;
        glin[u] = 0.f;
        done[u] = 0;                        // entries of the segment folded so far (scatter_apply_kernel's completion ticket)
        if (c >= LONG_SEGMENT) {            // a hot id (Criteo's numeric fields: every example): reduced by a block of its own
            const int k = atomicAdd(&counters[2], 1);
            if (k < long_cap) long_list[k] = u;
        }
    }
    // medium segments (SHORT_SEGMENT < c < LONG_SEGMENT, ~10^3 per Criteo batch): the list embed_scatter_apply deals to waves;
    // one atomic per block on the shared counter
    const bool med = u < U && c > SHORT_SEGMENT && c < LONG_SEGMENT;
    const unsigned long long mm = __ballot(med);
    __syncthreads();
    if (lane == 0) wsum[wave] = __popcll(mm);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < n_waves; ++w) tot += wsum[w];
        bbase = tot ? atomicAdd(&counters[4], tot) : 0;
    }
    __syncthreads();
    if (med) {
        int k = bbase + __popcll(mm & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) k += wsum[w];
        if (k < medium_cap) medium_list[k] = u;
    }
}

// zero the compact gradient rows of the U distinct ids (KQ float4 lanes per row)
template <int KQ>
__global__ __launch_bounds__(256) void group_finalize_kernel(const int32_t* __restrict__ counters, float4* __restrict__ gemb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t / KQ >= counters[0]) return;
    gemb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void group_fill_kernel(const int32_t* __restrict__ ids, int B, int F, int64_t rows,
                                                        const int32_t* __restrict__ slot, int32_t* __restrict__ cursor,
                                                        int32_t* __restrict__ perm, int32_t* __restrict__ seg_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = B * F;
    const int lane = threadIdx.x & 63;
    int id = -1;
    bool valid = false;
    if (i < n) {
        const int f = i / B, b = i - f * B;
        id = ids[(size_t)b * F + f];
        valid = (id >= 0) && ((int64_t)id < rows);
    }
    const WaveGroup g = wave_group(id, valid, lane);
    int u = 0, base = 0;
    if (valid) u = slot[id] - 1;
    if (valid && g.leader == lane) base = atomicAdd(&cursor[u], g.count);
    base = __shfl(base, g.leader);
    if (valid) {
        const int pos = base + g.rank;
        perm[pos] = i;
        seg_of[pos] = u;
    }
}

// Small batches (the reference's own B = 128 / 256: ~10^4 entries): the whole grouping -- counters cleared, distinct ids counted, segments
// carved, entries filled -- as ONE block of 1024 threads that walks the entries in trips, the three phases separated by __syncthreads()
// instead of kernel boundaries.  Four enqueue calls (memset + count + segments + fill) become one: at B = 256 the step runs at the pace of
// the host's HIP calls (engine.hip record_train: lean step).  Every output is what the three kernels leave, up to the ORDER of the distinct
// ids in `uniq` -- which already depended on which block's atomic came first.
constexpr int GROUP_ONE_BLOCK_MAX = 16384;      // entries
__global__ __launch_bounds__(GROUP_BLOCK) void group_one_block_kernel(const int32_t* __restrict__ ids, int B, int F, int64_t rows, int32_t* __restrict__ slot,
                                                                      int32_t* __restrict__ uniq, int32_t* __restrict__ counters, int32_t* __restrict__ cnt,
                                                                      int32_t* __restrict__ seg_start, int32_t* __restrict__ cursor, float* __restrict__ glin,
                                                                      int32_t* __restrict__ long_list, int long_cap, int32_t* __restrict__ done,
                                                                      int32_t* __restrict__ medium_list, int medium_cap, int LONG_SEGMENT,
                                                                      int32_t* __restrict__ perm, int32_t* __restrict__ seg_of) {
    __shared__ int s_U, s_total, s_long, s_med;
    __shared__ int wsum[GROUP_BLOCK / 64];
    const int n = B * F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_U = 0; s_total = 0; s_long = 0; s_med = 0; }
    __syncthreads();
    // ---- phase 1: count (group_count_kernel's body per trip; the compact index comes from an LDS counter)
    for (int base = 0; base < n; base += GROUP_BLOCK) {
        const int i = base + (int)threadIdx.x;
        int id = -1;
        bool valid = false;
        if (i < n) {
            const int f = i / B, b = i - f * B;
            id = ids[(size_t)b * F + f];
            valid = (id >= 0) && ((int64_t)id < rows);
        }
        const WaveGroup g = wave_group(id, valid, lane);
        bool first = false;
        if (valid && g.leader == lane) first = atomicAdd(&slot[id], g.count) == 0;
        const unsigned long long fm = __ballot(first);
        int wbase = 0;
        if (lane == 0 && fm != 0ull) wbase = atomicAdd(&s_U, __popcll(fm));
        wbase = __shfl(wbase, 0);
        if (first) uniq[wbase + __popcll(fm & ((1ull << lane) - 1ull))] = id;
    }
    __syncthreads();
    const int U = s_U;
    // ---- phase 2: segments (group_segments_kernel's body per trip of 1024 distinct ids)
    for (int base = 0; base < U; base += GROUP_BLOCK) {
        const int u = base + (int)threadIdx.x;
        int c = 0, id = 0;
        if (u < U) { id = uniq[u]; c = slot[id]; }
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        const int trip_base = s_total;
        int before = 0, tot = 0;
        for (int w = 0; w < GROUP_BLOCK / 64; ++w) { if (w < wave) before += wsum[w]; tot += wsum[w]; }
        if (u < U) {
            const int s0 = trip_base + before + incl - c;
            cnt[u] = c;
            seg_start[u] = s0;
            cursor[u] = s0;
            slot[id] = u + 1;
            glin[u] = 0.f;
            done[u] = 0;
            if (c >= LONG_SEGMENT) {
                const int k = atomicAdd(&s_long, 1);
                if (k < long_cap) long_list[k] = u;
            } else if (c > SHORT_SEGMENT) {
                const int k = atomicAdd(&s_med, 1);
                if (k < medium_cap) medium_list[k] = u;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_total = trip_base + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { counters[0] = U; counters[1] = s_total; counters[2] = s_long; counters[3] = 0; counters[4] = s_med; }
    __syncthreads();          // (slot / cursor written above are read below by other waves of THIS block: workgroup-scope ordering)
    // ---- phase 3: fill (group_fill_kernel's body per trip)
    for (int base = 0; base < n; base += GROUP_BLOCK) {
        const int i = base + (int)threadIdx.x;
        int id = -1;
        bool valid = false;
        if (i < n) {
            const int f = i / B, b = i - f * B;
            id = ids[(size_t)b * F + f];
            valid = (id >= 0) && ((int64_t)id < rows);
        }
        const WaveGroup g = wave_group(id, valid, lane);
        int u = 0, pos0 = 0;
        if (valid) u = slot[id] - 1;
        if (valid && g.leader == lane) pos0 = atomicAdd(&cursor[u], g.count);
        pos0 = __shfl(pos0, g.leader);
        if (valid) {
            const int pos = pos0 + g.rank;
            perm[pos] = i;
            seg_of[pos] = u;
        }
    }
}

// grouped positions per walker: 8 measured best on c2 once the long segments have their own blocks (6: 20.2 us, 8: 20.0, 12: 22.6,
// 16: 27.0); A/B knob DCTR_SCATTER_RUN

// gradient of one grouped entry i = f*B + b w.r.t. its table row piece kq, times the entry's value
template <int KQ, int MODE>
__device__ __forceinline__ float4 entry_grad(int i, int kq, const float4* __restrict__ dE, int de_ld4, const float4* __restrict__ e,
                                             int e_ld4, const float4* __restrict__ S, const float* __restrict__ coef,
                                             const float* __restrict__ vals, int B, int F, int& b_out, float& v_out,
                                             const int32_t* __restrict__ entry_row) {
    // fixed-F batches: entry i = f*B + b, value vals[b,f].  CSR batches (entry_row != nullptr, F == 1): entry i belongs to
    // example entry_row[i] (whose gradient row it reads), value vals[i] (or 1)
    int f = i / B, b = i - f * B;
    float v;
    if (entry_row != nullptr) { v = vals != nullptr ? vals[i] : 1.0f; b = entry_row[i]; f = 0; }
    else v = vals[(size_t)b * F + f];
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dE != nullptr) d = dE[(size_t)b * de_ld4 + (size_t)f * KQ + kq];
    if (MODE == DCTR_GATHER_FM) {
        // y_v = 0.5 sum_k[(sum_f e)^2 - sum_f e^2]  =>  d y_v / d e[b,f,k] = S[b,k] - e[b,f,k]   (DeepFM.py:133-135)
        const float4 ee = e[(size_t)b * e_ld4 + (size_t)f * KQ + kq];
        const float4 s = S[(size_t)b * KQ + kq];
        const float c = coef[b];
        d.x += c * (s.x - ee.x); d.y += c * (s.y - ee.y); d.z += c * (s.z - ee.z); d.w += c * (s.w - ee.w);
    } else if (MODE == DCTR_GATHER_BI) {
        // bi[b,k] = 0.5[(sum_f e)^2 - sum_f e^2]  =>  d e[b,f,k] = dbi[b,k] (S[b,k] - e[b,f,k])    (NFM.py:126-128)
        const float4 ee = e[(size_t)b * e_ld4 + (size_t)f * KQ + kq];
        const float4 s = S[(size_t)b * KQ + kq];
        const float4 c = reinterpret_cast<const float4*>(coef)[(size_t)b * KQ + kq];
        d.x += c.x * (s.x - ee.x); d.y += c.y * (s.y - ee.y); d.z += c.z * (s.z - ee.z); d.w += c.w * (s.w - ee.w);
    }
    b_out = b; v_out = v;
    return make_float4(d.x * v, d.y * v, d.z * v, d.w * v);
}

template <int KQ, int MODE>
__global__ __launch_bounds__(256) void scatter_bwd_kernel(
    const int32_t* __restrict__ perm, const int32_t* __restrict__ seg_of, const int32_t* __restrict__ counters,
    const int32_t* __restrict__ seg_start, const int32_t* __restrict__ cnt,
    const float4* __restrict__ dE, int de_ld4, const float4* __restrict__ e, int e_ld4,
    const float4* __restrict__ S, const float* __restrict__ coef, const float* __restrict__ dy,
    const float* __restrict__ vals, int B, int F, float* __restrict__ gemb, float* __restrict__ glin, int dy_ld, int run,
    int walker_blocks, const int32_t* __restrict__ long_list, int long_cap, const int32_t* __restrict__ entry_row) {
    constexpr int LONG_SEGMENT = long_segment(KQ), LONG_CHUNK = long_segment(KQ);
    if ((int)blockIdx.x >= walker_blocks) {
        // ---- long segments, cut into chunks of LONG_CHUNK entries dealt round-robin to these blocks: KQ lanes per entry,
        // 256/KQ entries in flight per pass, tree reduction in LDS, ONE atomic flush per (chunk, row piece)
        constexpr int EL = 256 / KQ;
        __shared__ float4 red[256];
        __shared__ float redl[256];
        const int kq = threadIdx.x % KQ, el = threadIdx.x / KQ;
        const int n_long = min(counters[2], long_cap);
        const int NB = (int)gridDim.x - walker_blocks, blk = (int)blockIdx.x - walker_blocks;
        int base = 0;
        for (int li = 0; li < n_long; ++li) {
            const int u = long_list[li];
            const int s0 = seg_start[u], len = cnt[u];
            const int nch = (len + LONG_CHUNK - 1) / LONG_CHUNK;
            for (int c = ((blk - base) % NB + NB) % NB; c < nch; c += NB) {
                const int c0 = s0 + c * LONG_CHUNK, c1 = min(s0 + len, c0 + LONG_CHUNK);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                float accl = 0.f;
                for (int j = c0 + el; j < c1; j += EL) {
                    int b; float v;
                    const float4 d = entry_grad<KQ, MODE>(perm[j], kq, dE, de_ld4, e, e_ld4, S, coef, vals, B, F, b, v, entry_row);
                    acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
                    if (kq == 0 && dy != nullptr) accl += dy[(size_t)b * dy_ld] * v;
                }
                red[threadIdx.x] = acc; redl[threadIdx.x] = accl;
                __syncthreads();
                for (int half = EL / 2; half >= 1; half >>= 1) {        // tree over the entry lanes (EL is a power of two)
                    if (el < half) {
                        const float4 o = red[threadIdx.x + half * KQ];
                        float4 m = red[threadIdx.x];
                        m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
                        red[threadIdx.x] = m;
                        redl[threadIdx.x] += redl[threadIdx.x + half * KQ];
                    }
                    __syncthreads();
                }
                if (el == 0) {
                    float* g = gemb + ((size_t)u * KQ + kq) * 4;
                    const float4 m = red[kq];
                    atomicAdd(g + 0, m.x); atomicAdd(g + 1, m.y); atomicAdd(g + 2, m.z); atomicAdd(g + 3, m.w);
                    if (kq == 0 && glin != nullptr) atomicAdd(glin + u, redl[0]);
                }
                __syncthreads();
            }
            base += nch;
        }
        return;
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = t / KQ, kq = t % KQ;
    const int total = counters[1];
    const int j0 = w * run;
    if (j0 >= total) return;
    const int jend = min(total, j0 + run);
    int cur = -1;
    bool inside = false;          // the current segment started inside this run
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float accl = 0.f;
    // a segment that lies entirely inside the run (the common case: most ids occur once per batch) is written with plain
    // 16-byte stores; only segments that straddle runs use float atomics on the pre-zeroed compact row
    auto flush = [&](bool whole) {
        float* g = gemb + ((size_t)cur * KQ + kq) * 4;
        if (whole) {
            *reinterpret_cast<float4*>(g) = acc;
            if (kq == 0 && glin != nullptr) glin[cur] = accl;
        } else {
            atomicAdd(g + 0, acc.x); atomicAdd(g + 1, acc.y); atomicAdd(g + 2, acc.z); atomicAdd(g + 3, acc.w);
            if (kq == 0 && glin != nullptr) atomicAdd(glin + cur, accl);
        }
    };
    for (int j = j0; j < jend; ++j) {
        const int u = seg_of[j];
        if (u != cur) {
            if (cur >= 0) flush(inside);                      // ended by a key change: it ended inside the run
            if (cnt[u] >= LONG_SEGMENT) { cur = -1; continue; }   // (a long segment: left to its block)
            inside = (j > j0) || (seg_start[u] == j0);
            cur = u;
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            accl = 0.f;
        }
        int b; float v;
        const float4 d = entry_grad<KQ, MODE>(perm[j], kq, dE, de_ld4, e, e_ld4, S, coef, vals, B, F, b, v, entry_row);
        acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
        if (kq == 0 && dy != nullptr) accl += dy[(size_t)b * dy_ld] * v;
    }
    if (cur >= 0) flush(inside && (seg_start[cur] + cnt[cur] <= jend));
}

template <int KQ>
static int launch_scatter(Group* g, const float* dE, int de_ld, const float* e, int e_ld, const float* S,
                          const float* coef, const float* dy, const float* vals, int B, int F, int mode,
                          float* gemb, float* glin, int dy_ld, hipStream_t st, const int32_t* entry_row) {
    const int64_t n = (int64_t)B * F;
    static const int run = getenv("DCTR_SCATTER_RUN") ? atoi(getenv("DCTR_SCATTER_RUN")) : 8;
    const int walkers = ceil_div(n, run);
    const int walker_blocks = ceil_div((int64_t)walkers * KQ, 256);
    const int long_blocks = (int)std::min<int64_t>(ceil_div(n, long_segment(KQ)), 256);
    dim3 grid(walker_blocks + long_blocks), block(256);
#define DCTR_SC(MODE_)                                                                                         \
    scatter_bwd_kernel<KQ, MODE_><<<grid, block, 0, st>>>(                                                     \
        g->perm, g->seg_of, g->counters, g->seg_start, g->cnt, reinterpret_cast<const float4*>(dE), de_ld / 4,                       \
        reinterpret_cast<const float4*>(e), e_ld / 4, reinterpret_cast<const float4*>(S), coef, dy, vals, B, F, \
        gemb, glin, dy_ld, run, walker_blocks, g->long_list, (int)g->long_cap, entry_row)
    switch (mode) {
        case DCTR_GATHER_RAW: DCTR_SC(DCTR_GATHER_RAW); break;
        case DCTR_GATHER_FM:  DCTR_SC(DCTR_GATHER_FM); break;
        case DCTR_GATHER_BI:  DCTR_SC(DCTR_GATHER_BI); break;
        default: set_error("scatter: bad mode %d", mode); return DCTR_ERR_INVALID_ARG;
    }
#undef DCTR_SC
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// ---- scatter + table optimizer in ONE launch (the training step's tail) -------------------------------------------------------
// Same walk as scatter_bwd_kernel, but a segment's gradient sum goes straight into the optimizer step of its table row instead
// of into the compact [U, K] buffer that a second launch (opt_table_kernel<.., false>) used to re-read:
//   * segments of up to SHORT_SEGMENT entries (most ids occur once per batch): one walker of K/4 lanes per distinct id, the
//     entries' gradients loaded together, summed in registers, applied at once;
//   * up to long_segment(KQ): one wave per segment (a compacted list built by group_segments_kernel), folded with shuffles;
//   * longer ones, reduced chunk-wise by blocks of their own: partial sums meet in the compact row by float atomics, every
//     chunk then adds its entry count to done[u]; the block that completes cnt[u] takes the total back out with
//     atomicExch(.., 0) -- which also leaves the row zeroed for the next batch, so the fused path needs no group_finalize
//     pass -- and applies it.  No fences: everything that crosses blocks is an atomic, and an add is waited for (its returned
//     value consumed) before its ticket is drawn.
// Row arithmetic = opt_table_kernel<KIND, KQ, false>, operation for operation (l2*theta + segment sum, opt_update, sum theta^2).
struct TableStep {
    float4* emb; float4* s0; float4* s1;        // [rows, K] parameters and optimizer slots
    float* lin; float* l0; float* l1;           // [rows] linear weights and slots (nullptr: the model has none)
    const Hyper* hdev; Hyper hval;
    float l2;
    float* sumsq_emb; float* sumsq_lin;         // SUMSQ_SHARDS-way sharded sum theta^2 of the visited rows (pre-update)
    const int32_t* uniq;                        // distinct id u -> table row
    int32_t* slot;                              // the grouping's slot word of every visited row goes back to 0 (no group_reset pass)
    int32_t* done;                              // [U] completion tickets
    uint8_t* ts; const StepState* state;        // lagging rows (lag.h; Adam): advanced to t-1 before this step's update, stamped t; nullptr = classic
    int ld4; int lin_ld;                        // row strides of emb / s0 / s1 (float4 units) and of lin / l0 / l1 (floats): KQ and 1, or the record stride (engine.h)
    int hi_prio;                                // A/B knob DCTR_TAIL_PRIO=1: the fused tail's waves raise their issue priority (the next gather waits for this kernel)
};

// the row's pieces are LOADED as soon as the distinct id is known (before the gradient loads: one latency instead of two) and
// stepped once the gradient sum is there
struct RowRegs { float4 th, a, b; float lt, la, lb; int64_t r; int nlag; };

template <int KIND, int KQ>
__device__ __forceinline__ RowRegs table_row_load(const TableStep& T, int u, int kq) {
    constexpr bool TWO = (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL);
    RowRegs R;
    R.r = T.uniq[u];
    const size_t i4 = (size_t)R.r * T.ld4 + kq, il = (size_t)R.r * T.lin_ld;
    R.th = T.emb[i4]; R.a = T.s0[i4]; R.b = TWO ? T.s1[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
    R.lt = R.la = R.lb = 0.f;
    if (kq == 0 && T.lin != nullptr) { R.lt = T.lin[il]; R.la = T.l0[il]; R.lb = TWO ? T.l1[il] : 0.f; }
    R.nlag = (KIND == DCTR_OPT_ADAM && T.ts != nullptr) ? lag_behind(T.state->t - 1, T.ts[R.r]) : 0;
    return R;
}

template <int KIND, int KQ>
__device__ __forceinline__ void table_row_step(const TableStep& T, const Hyper& h, RowRegs& R, int kq, float4 gs, float gl, float& sq, float& sql) {
    constexpr bool TWO = (KIND == DCTR_OPT_ADAM || KIND == DCTR_OPT_FTRL);
    const size_t i4 = (size_t)R.r * T.ld4 + kq, il = (size_t)R.r * T.lin_ld;
    float4 th = R.th, a = R.a, b = R.b;
    if (KIND == DCTR_OPT_ADAM && R.nlag > 0) {          // the steps no batch touched this row: replayed first (lag.h)
        const int64_t first = T.state->t - R.nlag;
        lag_catch_up4_lin(T.state, h, T.l2, first + R.nlag - 1, R.nlag, th, a, b, (kq == 0 && T.lin != nullptr) ? R.nlag : 0, R.lt, R.la, R.lb);
    }
    sq += th.x * th.x + th.y * th.y + th.z * th.z + th.w * th.w;
    float4 g = make_float4(T.l2 * th.x, T.l2 * th.y, T.l2 * th.z, T.l2 * th.w);
    g.x += gs.x; g.y += gs.y; g.z += gs.z; g.w += gs.w;
    opt_update(KIND, h, th.x, a.x, b.x, g.x);
    opt_update(KIND, h, th.y, a.y, b.y, g.y);
    opt_update(KIND, h, th.z, a.z, b.z, g.z);
    opt_update(KIND, h, th.w, a.w, b.w, g.w);
    T.emb[i4] = th; T.s0[i4] = a;
    if (TWO) T.s1[i4] = b;
    if (kq == 0) T.slot[R.r] = 0;
    if (KIND == DCTR_OPT_ADAM && kq == 0 && T.ts != nullptr) T.ts[R.r] = (uint8_t)T.state->t;
    if (kq == 0 && T.lin != nullptr) {
        float lt = R.lt, la = R.la, lb = R.lb;
        sql += lt * lt;
        float lg = T.l2 * lt;
        lg += gl;
        opt_update(KIND, h, lt, la, lb, lg);
        T.lin[il] = lt; T.l0[il] = la;
        if (TWO) T.l1[il] = lb;
    }
}

template <int KIND, int KQ>
__device__ __forceinline__ void table_step_row(const TableStep& T, const Hyper& h, int u, int kq, float4 gs, float gl, float& sq, float& sql) {
    RowRegs R = table_row_load<KIND, KQ>(T, u, kq);
    table_row_step<KIND, KQ>(T, h, R, kq, gs, gl, sq, sql);
}

// returned-atomic float add: the result is consumed (asm), so the wave waits until the add has been PERFORMED at the device-scope
// point of coherence -- what orders it before the completion ticket that follows, without a release fence (a __threadfence()
// writes back the XCD's whole L2: ~3.5 us each, MI355X_MICROARCH.md "inter-workgroup visibility")
__device__ __forceinline__ void atomic_add4_performed(float* p, float4 v, float* pl, float vl) {
    const float o0 = atomicAdd(p + 0, v.x), o1 = atomicAdd(p + 1, v.y), o2 = atomicAdd(p + 2, v.z), o3 = atomicAdd(p + 3, v.w);
    const float o4 = pl != nullptr ? atomicAdd(pl, vl) : 0.f;             // (all five in flight, one wait)
    asm volatile("" ::"v"(o0), "v"(o1), "v"(o2), "v"(o3), "v"(o4));
}


template <int KIND, int KQ, int MODE>
__global__ __launch_bounds__(256) void scatter_apply_kernel(
    const int32_t* __restrict__ perm, const int32_t* __restrict__ counters,
    const int32_t* __restrict__ seg_start, const int32_t* __restrict__ cnt,
    const float4* __restrict__ dE, int de_ld4, const float4* __restrict__ e, int e_ld4,
    const float4* __restrict__ S, const float* __restrict__ coef, const float* __restrict__ dy,
    const float* __restrict__ vals, int B, int F, float* __restrict__ gemb, float* __restrict__ glin, int dy_ld,
    int short_blocks, int medium_blocks, const int32_t* __restrict__ medium_list, int medium_cap,
    const int32_t* __restrict__ long_list, int long_cap, const int32_t* __restrict__ entry_row, TableStep T) {
    if (T.hi_prio) __builtin_amdgcn_s_setprio(3);
    const Hyper h = load_hyper(T.hdev, T.hval);
    float sq = 0.f, sql = 0.f;
    __shared__ float4 red[256];
    __shared__ float redl[256];
    __shared__ int last_flag;
    auto grad = [&](int i, int kq, float& gl) {
        int b; float v;
        const float4 d = entry_grad<KQ, MODE>(i, kq, dE, de_ld4, e, e_ld4, S, coef, vals, B, F, b, v, entry_row);
        gl = (kq == 0 && dy != nullptr) ? dy[(size_t)b * dy_ld] * v : 0.f;
        return d;
    };
    // (block order = dispatch order: the long segments' blocks -- the longest dependent chains -- start first, the many short
    //  walkers fill in behind them)
    constexpr int LONG_CHUNK = long_segment(KQ);
    const int long_blocks = (int)gridDim.x - short_blocks - medium_blocks;
    if ((int)blockIdx.x >= long_blocks + medium_blocks) {
        // ---- short segments (most ids occur once or twice per batch): walker u owns distinct id u, its entries' gradients are
        // loaded together (up to 4 independent loads in flight per lane) and the sum goes straight into the row's optimizer step
        const int t = ((int)blockIdx.x - long_blocks - medium_blocks) * blockDim.x + threadIdx.x;
        const int u = t / KQ, kq = t % KQ;
        const int len = u < counters[0] ? cnt[u] : 0;
        if (len >= 1 && len <= SHORT_SEGMENT) {
            const int s0 = seg_start[u];
            RowRegs R = table_row_load<KIND, KQ>(T, u, kq);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            float accl = 0.f;
            if (len == 1) {
                acc = grad(perm[s0], kq, accl);
            } else {
#pragma unroll
                for (int base = 0; base < SHORT_SEGMENT; base += 4) {
                    if (base >= len) break;
                    int pj[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pj[k] = perm[s0 + min(base + k, len - 1)];       // (clamped: the loads carry no branch)
                    float4 d[4]; float gl[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = grad(pj[k], kq, gl[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (base + k < len) { acc.x += d[k].x; acc.y += d[k].y; acc.z += d[k].z; acc.w += d[k].w; accl += gl[k]; }
                }
            }
            table_row_step<KIND, KQ>(T, h, R, kq, acc, accl, sq, sql);
        }
    } else if ((int)blockIdx.x >= long_blocks) {
        // ---- medium segments (the hot categories of each field): one WAVE per segment, 64/KQ entries in flight per pass, folded
        // with shuffles -- still no atomics
        constexpr int EL = 64 / KQ;
        const int lane = threadIdx.x & 63, kq = lane % KQ, el = lane / KQ;
        const int n_med = min(counters[4], medium_cap);
        const int wave = ((int)blockIdx.x - long_blocks) * 4 + (threadIdx.x >> 6), n_waves = medium_blocks * 4;
        for (int mi = wave; mi < n_med; mi += n_waves) {
            const int u = medium_list[mi];
            const int s0 = seg_start[u], len = cnt[u];
            RowRegs R;
            if (el == 0) R = table_row_load<KIND, KQ>(T, u, kq);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            float accl = 0.f;
            for (int j0 = el; j0 < len; j0 += 4 * EL) {              // 4 passes of the wave in flight (clamped: the loads carry no branch)
                int pj[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) pj[k] = perm[s0 + min(j0 + k * EL, len - 1)];
                float4 d[4]; float gl[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = grad(pj[k], kq, gl[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (j0 + k * EL < len) { acc.x += d[k].x; acc.y += d[k].y; acc.z += d[k].z; acc.w += d[k].w; accl += gl[k]; }
            }
#pragma unroll
            for (int o = KQ; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
                accl += __shfl_xor(accl, o);
            }
            if (el == 0) table_row_step<KIND, KQ>(T, h, R, kq, acc, accl, sq, sql);
        }
    } else {
        // ---- long segments (Criteo's numeric ids: every example of the batch): chunks of LONG_CHUNK entries dealt round-robin to
        // these blocks, partial sums meet in the compact row by (performed) float atomics, the block that completes cnt[u] takes
        // the total back out -- leaving zeros for the next batch -- and steps the row
        constexpr int EL = 256 / KQ;
        const int kq = threadIdx.x % KQ, el = threadIdx.x / KQ;
        const int n_long = min(counters[2], long_cap);
        const int NB = long_blocks, blk = (int)blockIdx.x;
        int base = 0;
        for (int li = 0; li < n_long; ++li) {
            const int u = long_list[li];
            const int s0 = seg_start[u], len = cnt[u];
            const int nch = (len + LONG_CHUNK - 1) / LONG_CHUNK;
            for (int c = ((blk - base) % NB + NB) % NB; c < nch; c += NB) {
                const int c0 = s0 + c * LONG_CHUNK, c1 = min(s0 + len, c0 + LONG_CHUNK);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                float accl = 0.f;
                for (int j0 = c0 + el; j0 < c1; j0 += 4 * EL) {
                    int pj[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pj[k] = perm[min(j0 + k * EL, c1 - 1)];
                    float4 d[4]; float gl[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = grad(pj[k], kq, gl[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (j0 + k * EL < c1) { acc.x += d[k].x; acc.y += d[k].y; acc.z += d[k].z; acc.w += d[k].w; accl += gl[k]; }
                }
                red[threadIdx.x] = acc; redl[threadIdx.x] = accl;
                __syncthreads();
                for (int half = EL / 2; half >= 1; half >>= 1) {
                    if (el < half) {
                        const float4 o = red[threadIdx.x + half * KQ];
                        float4 m = red[threadIdx.x];
                        m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
                        red[threadIdx.x] = m;
                        redl[threadIdx.x] += redl[threadIdx.x + half * KQ];
                    }
                    __syncthreads();
                }
                float* g = gemb + ((size_t)u * KQ + kq) * 4;
                if (el == 0) {
                    const float4 m = red[kq];
                    atomic_add4_performed(g, m, (kq == 0 && dy != nullptr) ? glin + u : nullptr, redl[0]);
                }
                __syncthreads();                                    // every piece of the chunk's sum has been performed
                if (threadIdx.x == 0) last_flag = (atomicAdd(&T.done[u], c1 - c0) + (c1 - c0) == len);
                __syncthreads();
                if (last_flag && el == 0) {
                    float4 tot;
                    tot.x = atomicExch(g + 0, 0.f); tot.y = atomicExch(g + 1, 0.f); tot.z = atomicExch(g + 2, 0.f); tot.w = atomicExch(g + 3, 0.f);
                    const float tl = (kq == 0 && dy != nullptr) ? atomicExch(glin + u, 0.f) : 0.f;
                    table_step_row<KIND, KQ>(T, h, u, kq, tot, tl, sq, sql);
                }
                __syncthreads();
            }
            base += nch;
        }
    }
    if (T.sumsq_emb != nullptr) {
        __shared__ float rs[2][4];
        sq = wave_sum_g(sq);
        sql = wave_sum_g(sql);
        if ((threadIdx.x & 63) == 0) { rs[0][threadIdx.x >> 6] = sq; rs[1][threadIdx.x >> 6] = sql; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float a = rs[0][0] + rs[0][1] + rs[0][2] + rs[0][3], bq = rs[1][0] + rs[1][1] + rs[1][2] + rs[1][3];
            if (a != 0.f) atomicAdd(T.sumsq_emb + (blockIdx.x & (SUMSQ_SHARDS - 1)), a);
            if (T.sumsq_lin != nullptr && bq != 0.f) atomicAdd(T.sumsq_lin + (blockIdx.x & (SUMSQ_SHARDS - 1)), bq);
        }
    }
}

template <int KIND, int KQ>
static int launch_scatter_apply(Group* g, const float* dE, int de_ld, const float* e, int e_ld, const float* S, const float* coef,
                                const float* dy, const float* vals, int B, int F, int mode, int dy_ld, hipStream_t st,
                                const int32_t* entry_row, const TableStep& T) {
    const int64_t n = (int64_t)B * F;
    const int short_blocks = ceil_div(n * KQ, 256);                                 // one walker per possible distinct id
    const int medium_blocks = (int)std::min<int64_t>(ceil_div(g->medium_cap, 4), 512);
    const int long_blocks = (int)std::min<int64_t>(ceil_div(n, long_segment(KQ)), 256);
    dim3 grid(short_blocks + medium_blocks + long_blocks), block(256);
#define DCTR_SA(MODE_)                                                                                         \
    scatter_apply_kernel<KIND, KQ, MODE_><<<grid, block, 0, st>>>(                                             \
        g->perm, g->counters, g->seg_start, g->cnt, reinterpret_cast<const float4*>(dE), de_ld / 4,             \
        reinterpret_cast<const float4*>(e), e_ld / 4, reinterpret_cast<const float4*>(S), coef, dy, vals, B, F, \
        g->gemb, g->glin, dy_ld, short_blocks, medium_blocks, g->medium_list, (int)g->medium_cap, g->long_list, \
        (int)g->long_cap, entry_row, T)
    switch (mode) {
        case DCTR_GATHER_RAW: DCTR_SA(DCTR_GATHER_RAW); break;
        case DCTR_GATHER_FM:  DCTR_SA(DCTR_GATHER_FM); break;
        case DCTR_GATHER_BI:  DCTR_SA(DCTR_GATHER_BI); break;
        default: set_error("scatter: bad mode %d", mode); return DCTR_ERR_INVALID_ARG;
    }
#undef DCTR_SA
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

// the step's tail in one launch: segment sums of the row gradients + the optimizer step of the batch's distinct rows (and their
// linear weights).  Requires a grouping made with group_ids(.., zero_gemb = false) on a group whose compact rows are all zero
// (g->gemb_clean), and leaves them so.
int embed_scatter_apply(Group* g, int kind, const Hyper* hdev, const Hyper& hval, float* emb, float* e0, float* e1, float* lin,
                        float* l0, float* l1, float l2, float* sumsq_emb, float* sumsq_lin, const float* dE, int de_ld,
                        const float* e, int e_ld, const float* S, const float* coef, const float* dy, const float* vals, int B, int F,
                        int K, int mode, hipStream_t st, int dy_ld, const int32_t* entry_row, uint8_t* lag_ts, const StepState* lag_state,
                        int tab_ld, int tab_lin_ld) {
    DCTR_REQUIRE(tab_ld % 4 == 0 && (tab_ld == 0 || tab_ld >= K), "scatter_apply: table row stride %d", tab_ld);
    DCTR_REQUIRE(K == g->K, "scatter: K=%d but group was created with K=%d", K, g->K);
    DCTR_REQUIRE(lag_ts == nullptr || (kind == DCTR_OPT_ADAM && lag_state != nullptr), "scatter_apply: lagging rows are an Adam-only scheme");
    DCTR_REQUIRE(g->gemb_clean, "scatter_apply: the group's compact gradient rows are not known to be zero");
    DCTR_REQUIRE(dE == nullptr || de_ld % 4 == 0, "scatter: de_ld must be a multiple of 4");
    DCTR_REQUIRE(mode == DCTR_GATHER_RAW || (e != nullptr && S != nullptr && coef != nullptr && e_ld % 4 == 0),
                 "scatter: FM/BI modes need e, S and coef");
    DCTR_REQUIRE((lin != nullptr) == (dy != nullptr), "scatter_apply: linear weights and their gradient source go together");
    TableStep T{reinterpret_cast<float4*>(emb), reinterpret_cast<float4*>(e0), reinterpret_cast<float4*>(e1), lin, l0, l1, hdev, hval,
                l2, sumsq_emb, sumsq_lin, g->uniq, g->slot, g->done, lag_ts, lag_state, tab_ld > 0 ? tab_ld / 4 : K / 4, tab_lin_ld, 0};
    static const bool tail_prio = [] { const char* v = getenv("DCTR_TAIL_PRIO"); return v != nullptr && v[0] == '1'; }();
    T.hi_prio = tail_prio ? 1 : 0;
    g->slots_clean = true;                  // (every distinct id of the grouping is visited exactly once, and each visit clears its slot word)
#define DCTR_Q(KD, Q) case Q: return launch_scatter_apply<KD, Q>(g, dE, de_ld, e, e_ld, S, coef, dy, vals, B, F, mode, dy_ld, st, entry_row, T)
#define DCTR_KD(KD) case KD: switch (K / 4) { DCTR_Q(KD, 1); DCTR_Q(KD, 2); DCTR_Q(KD, 4); DCTR_Q(KD, 8); DCTR_Q(KD, 16); DCTR_Q(KD, 32); DCTR_Q(KD, 64); \
                              default: set_error("scatter: K=%d unsupported", K); return DCTR_ERR_UNSUPPORTED; }
    switch (kind) {
        DCTR_KD(DCTR_OPT_ADAM) DCTR_KD(DCTR_OPT_ADAGRAD) DCTR_KD(DCTR_OPT_MOMENTUM) DCTR_KD(DCTR_OPT_FTRL)
        default: set_error("unknown optimizer kind %d", kind); return DCTR_ERR_INVALID_ARG;
    }
#undef DCTR_KD
#undef DCTR_Q
}

int group_create(int64_t rows, int64_t max_entries, int K, Group** out) {
    DCTR_REQUIRE(rows > 0 && max_entries > 0 && K % 4 == 0 && K >= 4, "group_create: bad sizes rows=%lld n=%lld K=%d",
                 (long long)rows, (long long)max_entries, K);
    Group* g = new Group();
    g->rows = rows; g->max_entries = max_entries; g->K = K;
    const size_t n = (size_t)max_entries;
    DCTR_HIP_CHECK(hipMalloc(&g->slot, (size_t)rows * 4));
    DCTR_HIP_CHECK(hipMemset(g->slot, 0, (size_t)rows * 4));
    DCTR_HIP_CHECK(hipMalloc(&g->uniq, n * 4));
    DCTR_HIP_CHECK(hipMalloc(&g->cnt, n * 4));
    DCTR_HIP_CHECK(hipMalloc(&g->seg_start, n * 4));
    DCTR_HIP_CHECK(hipMalloc(&g->cursor, n * 4));
    DCTR_HIP_CHECK(hipMalloc(&g->perm, n * 4));
    DCTR_HIP_CHECK(hipMalloc(&g->seg_of, n * 4));
    DCTR_HIP_CHECK(hipMalloc(&g->counters, 32));
    DCTR_HIP_CHECK(hipMemset(g->counters, 0, 32));
    DCTR_HIP_CHECK(hipMalloc(&g->gemb, n * K * 4));
    DCTR_HIP_CHECK(hipMemset(g->gemb, 0, n * K * 4));
    g->gemb_clean = true;
    g->slots_clean = true;
    DCTR_HIP_CHECK(hipMalloc(&g->glin, n * 4));
    DCTR_HIP_CHECK(hipMalloc(&g->done, n * 4));
    g->medium_cap = (int64_t)(n / (SHORT_SEGMENT + 1)) + 1;   // at most n / 9 segments are longer than SHORT_SEGMENT
    DCTR_HIP_CHECK(hipMalloc(&g->medium_list, (size_t)g->medium_cap * 4));
    g->long_cap = (int64_t)(n / long_segment(K / 4)) + 1;        // at most n / long_segment segments can be that long
    DCTR_HIP_CHECK(hipMalloc(&g->long_list, (size_t)g->long_cap * 4));
    // the memsets above run on the null stream and may still be pending: a first use on a non-blocking stream must not overtake them
    DCTR_HIP_CHECK(hipDeviceSynchronize());
    *out = g;
    return DCTR_OK;
}

int group_destroy(Group* g) {
    if (!g) return DCTR_OK;
    hipFree(g->slot); hipFree(g->uniq); hipFree(g->cnt); hipFree(g->seg_start); hipFree(g->cursor);
    hipFree(g->perm); hipFree(g->seg_of); hipFree(g->counters); hipFree(g->gemb); hipFree(g->glin); hipFree(g->long_list); hipFree(g->done); hipFree(g->medium_list);
    delete g;
    return DCTR_OK;
}

int group_ids(Group* g, const int32_t* ids, int B, int F, hipStream_t st, bool zero_gemb) {
    const int64_t n = (int64_t)B * F;
    DCTR_REQUIRE(n <= g->max_entries, "group_ids: B*F=%lld exceeds capacity %lld", (long long)n, (long long)g->max_entries);
    // always forget the previous batch (slot words back to 0, U = 0), even for an empty one -- unless the fused scatter + table step
    // has already put every slot word back (then only the counters are cleared)
    // small batches: ONE launch for the whole grouping (group_one_block_kernel clears the counters itself).  A/B knob DCTR_GROUP_ONE_BLOCK=0
    static const bool one_block_on = [] { const char* v = getenv("DCTR_GROUP_ONE_BLOCK"); return v == nullptr || v[0] != '0'; }();
    if (one_block_on && g->slots_clean && !zero_gemb && g->gemb_clean && n > 0 && n <= GROUP_ONE_BLOCK_MAX) {
        g->slots_clean = false;
        group_one_block_kernel<<<1, GROUP_BLOCK, 0, st>>>(ids, B, F, g->rows, g->slot, g->uniq, g->counters, g->cnt, g->seg_start, g->cursor, g->glin, g->long_list,
                                                          (int)g->long_cap, g->done, g->medium_list, (int)g->medium_cap, long_segment(g->K / 4), g->perm, g->seg_of);
        DCTR_LAUNCH_CHECK();
        return DCTR_OK;
    }
    if (g->slots_clean) DCTR_HIP_CHECK(hipMemsetAsync(g->counters, 0, 32, st));
    else group_reset_kernel<<<ceil_div(g->max_entries, 256), 256, 0, st>>>(g->slot, g->uniq, g->counters, (int)g->max_entries);
    if (n <= 0) { DCTR_LAUNCH_CHECK(); return DCTR_OK; }
    g->slots_clean = false;
    const int nb = ceil_div(n, 256);
    static const bool no_hash = getenv("DCTR_GROUP_NO_HASH") != nullptr;       // A/B knob
    const bool hashed = F == 1 && n >= 4 * HCHUNK && !no_hash;                  // CSR-ordered ids: aggregate per 2048-entry chunk in LDS
    if (hashed) group_count_hash_kernel<<<ceil_div(n, HCHUNK), 256, 0, st>>>(ids, (int)n, g->rows, g->slot, g->uniq, g->counters);
    else group_count_kernel<<<ceil_div(n, GROUP_BLOCK), GROUP_BLOCK, 0, st>>>(ids, B, F, g->rows, g->slot, g->uniq, g->counters);
    group_segments_kernel<<<ceil_div(n, GROUP_BLOCK), GROUP_BLOCK, 0, st>>>(g->slot, g->uniq, g->cnt, g->seg_start, g->cursor, g->counters, g->glin, g->long_list,
                                             (int)g->long_cap, g->done, g->medium_list, (int)g->medium_cap, long_segment(g->K / 4));
    const int KQ = g->K / 4;
    dim3 fgrid(ceil_div(n * KQ, 256));
    float4* gemb4 = reinterpret_cast<float4*>(g->gemb);
    // (the fused scatter + optimizer launch leaves the compact rows zeroed itself: its groupings skip this pass -- unless a plain
    //  scatter has written the rows since)
    if (!zero_gemb) {
        if (!g->gemb_clean) {       // a plain scatter has written compact rows since: zero them all once, the fused path keeps them so
            DCTR_HIP_CHECK(hipMemsetAsync(g->gemb, 0, (size_t)g->max_entries * g->K * 4, st));
            g->gemb_clean = true;
        }
    } else { switch (KQ) {
#define DCTR_FIN(Q) case Q: group_finalize_kernel<Q><<<fgrid, 256, 0, st>>>(g->counters, gemb4); break
        DCTR_FIN(1); DCTR_FIN(2); DCTR_FIN(4); DCTR_FIN(8); DCTR_FIN(16); DCTR_FIN(32); DCTR_FIN(64);
#undef DCTR_FIN
        default: set_error("group: K=%d unsupported", g->K); return DCTR_ERR_UNSUPPORTED;
    } }
    if (hashed) group_fill_hash_kernel<<<ceil_div(n, HCHUNK), 256, 0, st>>>(ids, (int)n, g->rows, g->slot, g->cursor, g->perm, g->seg_of);
    else group_fill_kernel<<<nb, 256, 0, st>>>(ids, B, F, g->rows, g->slot, g->cursor, g->perm, g->seg_of);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int embed_scatter_bwd(Group* g, const float* dE, int de_ld, const float* e, int e_ld, const float* S,
                      const float* coef, const float* dy, const float* vals, int B, int F, int K, int mode,
                      float* gemb, float* glin, hipStream_t st, int dy_ld, const int32_t* entry_row) {
    DCTR_REQUIRE(K == g->K, "scatter: K=%d but group was created with K=%d", K, g->K);
    DCTR_REQUIRE(dE == nullptr || de_ld % 4 == 0, "scatter: de_ld must be a multiple of 4");
    DCTR_REQUIRE(mode == DCTR_GATHER_RAW || (e != nullptr && S != nullptr && coef != nullptr && e_ld % 4 == 0),
                 "scatter: FM/BI modes need e, S and coef");
    if (gemb == nullptr) gemb = g->gemb;
    if (glin == nullptr && dy != nullptr) glin = g->glin;
    if (gemb == g->gemb) g->gemb_clean = false;             // (the fused path's invariant: all-zero compact rows between batches)
    switch (K / 4) {
#define DCTR_L(Q) case Q: return launch_scatter<Q>(g, dE, de_ld, e, e_ld, S, coef, dy, vals, B, F, mode, gemb, glin, dy_ld, st, entry_row)
        DCTR_L(1); DCTR_L(2); DCTR_L(4); DCTR_L(8); DCTR_L(16); DCTR_L(32); DCTR_L(64);
#undef DCTR_L
        default: set_error("scatter: K=%d unsupported", K); return DCTR_ERR_UNSUPPORTED;
    }
}

}  // namespace dctr

using namespace dctr;

extern "C" {

int dctr_group_create(int64_t rows, int64_t max_entries, int K, dctr_group_t* g) {
    DCTR_REQUIRE(g != nullptr, "null out pointer");
    Group* p = nullptr;
    DCTR_TRY(group_create(rows, max_entries, K, &p));
    *g = reinterpret_cast<dctr_group_t>(p);
    return DCTR_OK;
}
int dctr_group_destroy(dctr_group_t g) { return group_destroy(reinterpret_cast<Group*>(g)); }
int dctr_group_ids(dctr_group_t g, const int32_t* d_ids, int B, int F, void* stream) {
    DCTR_REQUIRE(g != nullptr, "null group");
    return group_ids(reinterpret_cast<Group*>(g), d_ids, B, F, as_stream(stream));
}
int dctr_group_num_unique(dctr_group_t g, int32_t* h_U, void* stream) {
    DCTR_REQUIRE(g != nullptr && h_U != nullptr, "null pointer");
    Group* p = reinterpret_cast<Group*>(g);
    DCTR_HIP_CHECK(hipMemcpyAsync(h_U, p->counters, 4, hipMemcpyDeviceToHost, as_stream(stream)));
    DCTR_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return DCTR_OK;
}
int dctr_group_buffers(dctr_group_t g, const int32_t** d_uniq, const int32_t** d_seg_start, const int32_t** d_cnt,
                       const int32_t** d_perm, const int32_t** d_slot, const int32_t** d_counters,
                       float** d_gemb, float** d_glin) {
    DCTR_REQUIRE(g != nullptr, "null group");
    Group* p = reinterpret_cast<Group*>(g);
    if (d_uniq) *d_uniq = p->uniq;
    if (d_seg_start) *d_seg_start = p->seg_start;
    if (d_cnt) *d_cnt = p->cnt;
    if (d_perm) *d_perm = p->perm;
    if (d_slot) *d_slot = p->slot;
    if (d_counters) *d_counters = p->counters;
    if (d_gemb) *d_gemb = p->gemb;
    if (d_glin) *d_glin = p->glin;
    return DCTR_OK;
}
int dctr_embed_scatter_bwd(dctr_group_t g, const float* d_dE, int de_ld, const float* d_e, int e_ld,
                           const float* d_sum, const float* d_coef, const float* d_dy, const float* d_vals, int B,
                           int F, int K, int mode, float* d_gemb, float* d_glin, void* stream) {
    DCTR_REQUIRE(g != nullptr, "null group");
    return embed_scatter_bwd(reinterpret_cast<Group*>(g), d_dE, de_ld, d_e, e_ld, d_sum, d_coef, d_dy, d_vals, B, F, K,
                             mode, d_gemb, d_glin, as_stream(stream));
}

// the step's tail as an op: segment sums of the row gradients straight into the optimizer step of the batch's distinct rows
// (scatter_apply_kernel).  `hyper` as for dctr_opt_table.  The group must have been filled by dctr_group_ids for these ids.
int dctr_embed_scatter_apply(dctr_group_t g, int kind, const float* hyper, float* d_emb, float* d_emb_s0, float* d_emb_s1, float* d_lin,
                             float* d_lin_s0, float* d_lin_s1, float l2, float* d_sumsq, const float* d_dE, int de_ld, const float* d_e,
                             int e_ld, const float* d_sum, const float* d_coef, const float* d_dy, const float* d_vals, int B, int F, int K,
                             int mode, void* stream) {
    DCTR_REQUIRE(g != nullptr && hyper != nullptr && d_emb != nullptr && d_emb_s0 != nullptr, "group, hyper, table and its first slot required");
    Group* G = reinterpret_cast<Group*>(g);
    Hyper h{};
    h.lr = hyper[0];
    h.beta1 = 0.9f; h.beta2 = 0.999f; h.eps = 1e-8f; h.momentum = 0.95f; h.lr_t = h.lr;
    if (kind == DCTR_OPT_ADAM) {
        h.beta1 = hyper[1]; h.beta2 = hyper[2]; h.eps = hyper[3];
        const double t = (double)hyper[4];
        h.lr_t = (float)((double)h.lr * sqrt(1.0 - pow((double)h.beta2, t)) / (1.0 - pow((double)h.beta1, t)));
    } else if (kind == DCTR_OPT_MOMENTUM) {
        h.momentum = hyper[1];
    }
    hipStream_t st = as_stream(stream);
    if (!G->gemb_clean) {           // a plain dctr_embed_scatter_bwd has used the compact rows since: the fused launch needs them zero
        DCTR_HIP_CHECK(hipMemsetAsync(G->gemb, 0, (size_t)G->max_entries * G->K * 4, st));
        G->gemb_clean = true;
    }
    return embed_scatter_apply(G, kind, nullptr, h, d_emb, d_emb_s0, d_emb_s1, d_lin, d_lin_s0, d_lin_s1, l2, d_sumsq,
                               d_sumsq ? d_sumsq + SUMSQ_SHARDS : nullptr, d_dE, de_ld, d_e, e_ld, d_sum, d_coef, d_dy, d_vals, B, F, K, mode, st);
}


}  // extern "C"
