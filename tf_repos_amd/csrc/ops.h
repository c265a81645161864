// Internal (namespace-level) declarations shared by the kernels' translation units and the engine.
#pragma once
#include "common.h"

namespace dctr {

// ---- optimizer scalars; a copy lives in device memory inside StepState so graphs replay correctly
struct Hyper {
    float lr, beta1, beta2, eps, lr_t, momentum;
    float pad[2];
};

constexpr int LR_HIST = 32;     // per-step Adam lr_t of the last LR_HIST steps (lag.h: rows that lag replay their missed steps with them)
struct StepState {
    int64_t t;          // global_step (number of optimizer steps applied so far)
    uint64_t seed;      // base dropout seed
    uint64_t seed_t;    // seed of the current step
    uint64_t row0;      // MUST follow seed_t (dropout_row0, common.h): index of this rank's first example in the GLOBAL batch of the step --
                        // a dropout mask is a function of the global example row, so N data-parallel ranks draw what one rank would
    Hyper hyper;
    Hyper hyper_lin;    // canned-estimator models: the linear side's optimizer (wide_n_deep.py:144-149); else a copy of hyper
    float lr_hist[LR_HIST];     // lr_hist[s % LR_HIST] = hyper.lr_t of step s
    int32_t lag_overflow;       // set (and kept: the next step's state is a copy) by a replay that met a row further behind than LAG_MAX_PERIOD steps --
    int32_t pad_;               // a broken invariant (stamp wrap, a direct caller of the owner-side API): dctr_check_ids reports it
};

// sum-of-squares outputs (the l2_loss part of the reported loss) are SUMSQ_SHARDS-way sharded by block index: thousands of
// blocks adding to ONE word serialise at ~90 atomics/us and used to cost more than the streaming pass itself
constexpr int SUMSQ_SHARDS = 64;
constexpr int OPT_BLOCK = 1024;   // elements of the dense arena handled by one optimizer block
struct OptBlockMeta {
    int64_t part_off;     // offset (floats) into the partial-gradient workspace for this block's first element
    int64_t part_stride;  // distance (floats) between consecutive partial slabs
    int32_t n_part;       // number of partial slabs to sum
    float l2;             // l2_reg coefficient if this variable is in the loss via l2_loss (DCN.py:199), else 0
};

// ---- K8 grouping state (group.hip)
struct Group {
    int64_t rows = 0;
    int64_t max_entries = 0;
    int K = 0;
    int32_t* slot = nullptr;      // [rows]   0 = untouched; count during grouping; u+1 afterwards
    int32_t* uniq = nullptr;      // [max_entries]
    int32_t* cnt = nullptr;       // [max_entries]
    int32_t* seg_start = nullptr; // [max_entries]
    int32_t* cursor = nullptr;    // [max_entries]
    int32_t* perm = nullptr;      // [max_entries]
    int32_t* seg_of = nullptr;    // [max_entries]
    int32_t* counters = nullptr;  // [8]: 0 = U (distinct ids), 1 = total grouped entries, 2 = long segments, 3 = reset ticket, 4 = medium segments
    int32_t* long_list = nullptr; // [long_cap] distinct-id indices u of the segments with >= long_segment(K/4) entries (group.hip)
    int64_t long_cap = 0;
    float* gemb = nullptr;        // [max_entries, K] compact gradient rows
    float* glin = nullptr;        // [max_entries]
    int32_t* medium_list = nullptr; // [medium_cap] distinct-id indices of the segments with SHORT_SEGMENT < entries < long_segment(K/4)
    int64_t medium_cap = 0;
    int32_t* done = nullptr;      // [max_entries] entries of each segment folded so far (embed_scatter_apply's completion tickets)
    bool slots_clean = false;     // host-side: every slot word is 0 (embed_scatter_apply clears the words of the rows it visits)
    bool gemb_clean = false;      // host-side: every compact gradient row is zero (kept so by embed_scatter_apply, broken by embed_scatter_bwd)
};
inline int64_t group_capacity(const Group* g) { return g->max_entries; }

int group_create(int64_t rows, int64_t max_entries, int K, Group** out);
int group_destroy(Group* g);
int group_ids(Group* g, const int32_t* ids, int B, int F, hipStream_t st, bool zero_gemb = true);
int embed_scatter_apply(Group* g, int kind, const Hyper* hdev, const Hyper& hval, float* emb, float* e0, float* e1, float* lin,
                        float* l0, float* l1, float l2, float* sumsq_emb, float* sumsq_lin, const float* dE, int de_ld,
                        const float* e, int e_ld, const float* S, const float* coef, const float* dy, const float* vals, int B, int F,
                        int K, int mode, hipStream_t st, int dy_ld = 1, const int32_t* entry_row = nullptr, uint8_t* lag_ts = nullptr,
                        const StepState* lag_state = nullptr,
                        int tab_ld = 0, int tab_lin_ld = 1);     // row strides (floats) of emb / e0 / e1 and lin / l0 / l1: 0 / 1 = dense [rows, K] / [rows] arrays
int embed_scatter_bwd(Group* g, const float* dE, int de_ld, const float* e, int e_ld, const float* S,
                      const float* coef, const float* dy, const float* vals, int B, int F, int K, int mode,
                      float* gemb, float* glin, hipStream_t st, int dy_ld = 1,     // dy_ld: stride (floats) between examples in dy
                      const int32_t* entry_row = nullptr);                       // CSR batches: example of every entry (F == 1, vals per entry)

// ---- K2 (gather.hip)
struct LagView;     // lag.h
int embed_gather_fwd(const float* emb, const float* lin, int64_t rows, const int32_t* ids, const float* vals,
                     int B, int F, int K, int mode, float* e, int e_ld, float* yw, float* sum, float* red,
                     int32_t* status, hipStream_t st, const LagView* lag = nullptr);

int embed_gather_strided(const float* emb, int emb_ld, const float* lin, int lin_ld, int64_t rows, const int32_t* ids,
                         const float* vals, int B, int F, int K, int mode, float* e, int e_ld, float* yw, float* sum, float* red,
                         int32_t* status, hipStream_t st, const LagView* lag = nullptr);

// ---- shard.hip: packed [K+4]-float row records for the row-sharded exchange
int pack_table_rows(const float* emb, const float* lin, int64_t rows, int K, const int32_t* rows_idx, int n, float* out,
                    int32_t* status, hipStream_t st, const LagView* lag = nullptr, int tab_ld = 0, int tab_lin_ld = 1);
int pack_unique_grads(const Group* g, const float* glin, const int32_t* upos, float* out, hipStream_t st);

// ---- K6 (gemm.hip)
// dctr_config.gemm_mode = 1 (gemm_dr3.hip): the product may take the split-precision kernels; w_fwd / w_dgr are the layer's weight
// pre-split into three bf16 planes in the two forms the forward and the dgrad product read (null: that product stays exact)
struct GemmOpt {
    int mode = 0;
    const unsigned* w_fwd = nullptr;
    int64_t fwd_plane = 0;
    const unsigned* w_dgr = nullptr;
    int64_t dgr_plane = 0;
    int wgrad_low_prio = 0;     // the weight gradient's waves keep the default issue priority (it runs beside the table step the next gather waits for)
};
int fc_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N,
           int relu, float keep, const uint64_t* seed_ptr, uint64_t seed, hipStream_t st, int over = 0, const GemmOpt* go = nullptr);
int fc_bwd_data(const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int K, int N,
                const float* act, int ldact, float keep_prev, hipStream_t st, int over = 0, const GemmOpt* go = nullptr);
int fc_bwd_weights_partials(const float* x, int ldx, const float* dy, int lddy, float* dw_part, int64_t dw_stride,
                            float* db_part, int64_t db_stride, int M, int K, int N, int splits, hipStream_t st, int over = 0, const GemmOpt* go = nullptr);
// gemm_dr3.hip: *done = false -> not taken
bool dr3_shape_ok(int M, int K, int N);
int64_t dr3_fwd_plane_bytes(int K, int N);
int64_t dr3_dgr_plane_bytes(int K, int N);
int dr3_wsplit(const float* w, int ldw, int K, int N, unsigned* fwd, unsigned* dgr, hipStream_t st);
struct WsplitJob { const float* w; int ldw, K, N; unsigned* fwd; unsigned* dgr; };
int dr3_wsplit_multi(const WsplitJob* jobs, int n, hipStream_t st);
int dr3_fc_fwd(const float* x, int ldx, const unsigned* wp, int64_t plane, const float* b, float* y, int ldy, int M, int K, int N, int relu, float keep,
               const uint64_t* seed_ptr, uint64_t seed, hipStream_t st, bool* done);
int dr3_fc_bwd_data(const float* dy, int lddy, const unsigned* wp, int64_t plane, float* dx, int lddx, int M, int K, int N, const float* act, int ldact,
                    float keep_prev, hipStream_t st, bool* done);
int dr3_fc_bwd_weights_partials(const float* x, int ldx, const float* dy, int lddy, float* dw_part, int64_t dw_stride, float* db_part,
                                int64_t db_stride, int M, int K, int N, int splits, hipStream_t st, bool* done, int low_prio = 0);
// `over` = 1: both operands are followed by GEMM_SLACK_ROWS rows (64 * ld floats) of readable memory, edge tiles may over-read
constexpr int GEMM_SLACK_ROWS = 64;
int sum_partials(const float* part, int64_t stride, int splits, int64_t n, float* out, hipStream_t st);
int choose_wgrad_splits(int M, int K, int N);
// gemm_dr.hip: the direct-to-register wave-split-K kernel family; *done = false -> not taken, the caller runs the LDS-tiled kernel
int dr_fc_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N, int relu, float keep,
              const uint64_t* seed_ptr, uint64_t seed, hipStream_t st, bool* done);
int dr_fc_bwd_data(const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int K, int N, const float* act, int ldact,
                   float keep_prev, hipStream_t st, bool* done);
int dr_fc_bwd_weights_partials(const float* x, int ldx, const float* dy, int lddy, float* dw_part, int64_t dw_stride, float* db_part,
                               int64_t db_stride, int M, int K, int N, int splits, hipStream_t st, bool* done);
int dr_fc_bwd_weights_partials_gate(const float* x, int ldx, const float* h, int ldh, const float* rowscale, const float* colscale,
                                    float* dw_part, int64_t dw_stride, float* db_part, int64_t db_stride, float* dwo_part, int64_t dwo_stride,
                                    int M, int K, int N, int splits, hipStream_t st, bool* done);
// tall operands, the small one resident in LDS (gemm_ws.hip; DCTR_GEMM_WS=0 turns it off)
bool ws_takes(int64_t M, int R, int N);
int ws_fc_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N, int relu, float keep,
              const uint64_t* seed_ptr, uint64_t seed, hipStream_t st, bool* done);
int ws_fc_bwd_data(const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int K, int N, const float* act, int ldact,
                   float keep_prev, hipStream_t st, bool* done);
int ws_fc_fwd_dot(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N, int relu,
                  const float* dot_w, float* dot_parts, int64_t dot_stride, int* n_parts, hipStream_t st, bool* done);
int ws_fc_bwd_data_gate(const float* h, int ldh, const float* rowscale, const float* kscale, const float* w, float* dx, int lddx, int M, int K,
                        int N, hipStream_t st, bool* done);
// tall operands in split precision (gemm_ts.hip; DCTR_GEMM_TS=0 turns it off): *done = false -> not taken.  planes_ws: ts_plane_bytes(R, N)
bool ts_takes(int64_t M, int R, int N);
size_t ts_plane_bytes(int R, int N);
// the tall operand as pair products of gathered embeddings (AFM.py:130-139), never stored: row b P + p = e[b, pair_i[p], :] . e[b, pair_j[p], :]
// with e [examples, e_ld], field f at f * K
struct TsPairs { const float* e; int e_ld; int examples; const int16_t* pair_i; const int16_t* pair_j; int P; };
bool ts_pairs_ok(const TsPairs* g, int64_t M, int K);
int ts_fc_fwd_dot(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int64_t M, int K, int N, const float* dot_w,
                  float* dot_out, void* planes_ws, bool split_here, hipStream_t st, bool* done, const TsPairs* pairs = nullptr, void* sign_bits_out = nullptr);
int ts_fc_bwd_data_gate(const float* h, int ldh, const float* rowscale, const float* kscale, const float* w, float* dx, int lddx, int64_t M, int K,
                        int N, void* planes_ws, bool split_here, hipStream_t st, bool* done, const void* sign_bits = nullptr);
inline size_t ts_sign_bytes(int64_t M) { return (size_t)M * 32; }      // the forward's sign words: one u64 per row and lane quarter
int ts_fc_bwd_weights_gate(const float* x, int ldx, const float* h, int ldh, const float* rowscale, const float* colscale, float* dw_part,
                           int64_t dw_stride, float* db_part, int64_t db_stride, float* dwo_part, int64_t dwo_stride, int64_t M, int K, int N, int splits,
                           hipStream_t st, bool* done, const TsPairs* pairs = nullptr, const void* sign_bits = nullptr, const float* w = nullptr,
                           const float* bias = nullptr);
bool ts_bits_enabled();
constexpr int TS_WGRAD_SLABS = 256;         // one block per slab and CU
int ts_prepare(const float* w, int K, int N, const float* kscale, void* fwd_planes, void* dgr_planes, hipStream_t st);
int dr_wgrad_splits(int M, int K, int N);
// Outer-PNN first layer with the pair products formed in the MFMA fragments (gemm_dr.hip)
bool opnn_fused_ok(int F, int K, int H);
int64_t opnn_fwd_ws_floats_max(int max_batch, int H);
int opnn_outer_fwd(const float* e, int e_ld, int B, int F, int K, const int* pairs, const float* w_outer, const float* bias, float* y, int ldy,
                   int H, int relu, float keep, const uint64_t* seed_ptr, uint64_t seed, float* ws, hipStream_t st);
int opnn_outer_wgrad(const float* e, int e_ld, int B, int F, int K, const int* pairs, const float* dy, int lddy, int H, float* dw_outer, hipStream_t st);
// ... and its backward to the embeddings without the [B, P K K] gradient tensor (opnn_dgrad.hip)
bool opnn_dgrad_fused_ok(int K, int H);
int opnn_outer_dgrad_fused(const float* dh, int lddh, int H, const float* w_outer, const float* e, int e_ld, const int* pairs, int B, int F, int K,
                           float* dE, int de_ld, hipStream_t st);

// ---- dense_ops.hip
int rowdot(const float* x, int ldx, const float* w, const float* bias, int M, int n, float* y, int accumulate,
           hipStream_t st);
int rank1_bwd(const float* dy, const float* w, int M, int n, const float* act, int ldact, float keep, float* dx,
              int lddx, int accumulate, hipStream_t st);
int colsum_partials(const float* Y, int ldy, const float* rs, int M, int N, int splits, float* out,
                    int64_t split_stride, hipStream_t st);
// the same for up to COLSUM_MAX_JOBS (Y, rs, out) triples of ONE shape, in one launch
constexpr int COLSUM_MAX_JOBS = 16;
struct ColsumJobs { const float* Y[COLSUM_MAX_JOBS]; const float* rs[COLSUM_MAX_JOBS]; float* out[COLSUM_MAX_JOBS]; int n; };
int colsum_partials_batch(const ColsumJobs& J, int ldy, int M, int N, int splits, int64_t split_stride, hipStream_t st);
int out_layer_bwd(const float* x, int ldx, const float* dy, const float* w, int M, int n, int splits, int masked,
                  float keep, float* dx, int lddx, float* dw_part, int64_t dw_stride, float* db_part, int64_t db_stride,
                  hipStream_t st);
int loss_head(const float* bias, const float* yw, const float* yv, const float* yd, const float* labels, int B,
              float inv_batch, float* y, float* prob, float* dy, float* loss_sum, hipStream_t st);
int opt_dense_arena(int kind, const Hyper* hdev, const Hyper& hval, float* theta, float* s0, float* s1,
                    const float* parts, const OptBlockMeta* meta, int n_blocks, float* gout, int apply, float* sumsq,
                    hipStream_t st);
int opt_dense_flat(int kind, const Hyper* hdev, const Hyper& hval, float* theta, float* s0, float* s1, const float* grad,
                   int n_part, int64_t part_stride, int64_t n, float l2, hipStream_t st);
int opt_table(int kind, const Hyper* hdev, const Hyper& hval, int table_mode, int64_t rows, int K, float* emb, float* e0,
              float* e1, float* lin, float* l0, float* l1, const int32_t* slot, const int32_t* uniq,
              const int32_t* counters, int64_t max_entries, const float* gemb, const float* glin, float l2,
              float* sumsq_emb, float* sumsq_lin, hipStream_t st, hipStream_t st_lin = nullptr, int pass = 0,
              int tab_ld = 0, int tab_lin_ld = 1);      // row strides (floats) of emb / e0 / e1 and lin / l0 / l1: 0 / 1 = dense [rows, K] / [rows] arrays
enum { OPT_PASS_ALL = 0, OPT_PASS_UNTOUCHED = 1, OPT_PASS_TOUCHED = 2 };
int step_state_advance(StepState* s, float* zero, int n_zero, hipStream_t st, uint64_t row0 = 0);
int step_state_next(const StepState* cur, StepState* nxt, float* zero, int n_zero, hipStream_t st);
int head_out_bwd(const float* x1, int ld1, const float* w1, int n1, int masked1, float* dx1, int lddx1,
                 const float* x2, int ld2, const float* w2, int n2, int masked2, float* dx2, int lddx2,
                 const float* b_out, const float* bias, const float* yw, const float* yv, const float* labels, int B, float inv_batch,
                 float keep, int splits, float* yd, float* y, float* prob, float* dy, float* loss_shards, float* dw_part,
                 int64_t dw_stride, float* db_part, int64_t db_stride, hipStream_t st);
int head_fused(const float* x1, int ld1, const float* w1, int n1, const float* x2, int ld2, const float* w2, int n2,
               const float* b_out, const float* bias, const float* yw, const float* yv, const float* labels, int B, float inv_batch,
               float* yd, float* y, float* prob, float* dy, float* loss_shards, hipStream_t st);

// ---- wnd.hip: dense (numeric-column) inputs and the linear side of the canned-estimator models
int wnd_dense_fwd(const float* dense, int nd, const float* wd, int B, float* x_in, int ldx, int col0, float* yw, hipStream_t st);
int opt_lin_touched(int kind, const Hyper* hdev, const Hyper& hval, float* lin, float* l0, float* l1, const int32_t* uniq,
                    const int32_t* counters, int64_t max_entries, const float* glin, hipStream_t st);

// ---- bn.hip: contrib.layers.batch_norm after each hidden layer's ReLU, then dropout (DeepFM.py:159-162, 231-235)
// cross-rank sum of small device vectors (batch_norm's column sums under data-parallel ranks); the transport's all-reduce
struct BnSync {
    int (*all_reduce)(void* ctx, int channel, float* d_buf, int64_t n, void* stream) = nullptr;
    void* ctx = nullptr;
    int world = 1;
};
int bn_forward(const float* y, int ldy, int B, int H, bool train, float eps, float decay, const float* gamma, const float* beta,
               float* mm, float* mv, float keep, const uint64_t* seed_ptr, uint64_t salt, float* stats, float* scratch, float* out,
               int ldo, hipStream_t st, const BnSync* sync = nullptr, bool bessel = true);
int bn_backward(const float* dout, int ldd, const float* y, int ldy, int B, int H, const float* stats, const float* gamma, float keep,
                const uint64_t* seed_ptr, uint64_t salt, float* scratch, float* dbeta, float* dgamma, float* dpre, int ldp,
                hipStream_t st, const BnSync* sync = nullptr);
int bn_scratch_floats(int H);

// ---- interact.hip
int pnn_inner_fwd(const float* e, int e_ld, int B, int F, int K, float* ip, int ip_ld, hipStream_t st);
int pnn_inner_bwd(const float* e, int e_ld, const float* dip, int dip_ld, int B, int F, int K, float* dE, int de_ld,
                  hipStream_t st);
int pnn_outer_fwd(const float* e, int e_ld, int B, int F, int K, float* op, int64_t op_ld, hipStream_t st);
int pnn_outer_bwd(const float* e, int e_ld, const float* dop, int64_t dop_ld, int B, int F, int K, float* dE, int de_ld,
                  hipStream_t st);
int dcn_cross_fwd(const float* x0, int x0_ld, const float* w, const float* b, int B, int D, int L, float* xs,
                  float* xlw, hipStream_t st);
int dcn_cross_bwd(const float* xs, const float* xlw, const float* w, const float* dxL, int dxl_ld, int B, int D, int L,
                  float* dx0, int dx0_ld, float* dw_part, float* db_part, int splits, int64_t part_stride,
                  float* scratch, hipStream_t st);
int dcn_cross_param_grads(const float* xs, int B, int D, int L, float* dw_part, float* db_part, int splits, int64_t part_stride,
                          const float* scratch, hipStream_t st);
// the training step's pair (interact.hip): forward keeps x_L and s only, backward re-forms the x_l and sums the parameter gradients itself
bool dcn_cross_lean_ok(int D, int L);
int dcn_cross_bwd_rows(int B, int D);       // rows of the backward's two [rows, L D] outputs
int dcn_cross_fwd_lean(const float* x0, int x0_ld, const float* w, const float* b, int B, int D, int L, float* xL, float* xlw, hipStream_t st);
int dcn_cross_bwd_fused(const float* x0, int x0_ld, const float* xlw, const float* w, const float* bias, const float* dxL, int dxl_ld, int B,
                        int D, int L, float* dx0, int dx0_ld, float* dw_rows, float* db_rows, hipStream_t st);
int dcn_cross_param_slabs(const float* dw_rows, const float* db_rows, int B, int D, int L, float* dw_part, float* db_part, int n_part,
                          int64_t part_stride, hipStream_t st);

// sparse.hip / mtl.hip: the CSR (multi-hot) models DIN / ESMM
int lookup_sparse_slots_fwd(const float* emb, int64_t rows, int K, const int32_t* offsets, const int32_t* ids, const float* weights,
                            int n_seg, int S, float* out, int out_ld, int32_t* status, hipStream_t st, int64_t nnz_hint = -1,
                            const LagView* lag = nullptr);
int csr_entry_offsets(const int32_t* offsets, int n_seg, int nnz, int S, int ld, int K, int32_t* entry_off, hipStream_t st);
int esmm_head(const float* h_ctr, int ld_ctr, const float* w_ctr, const float* b_ctr, int n_ctr, const float* h_cvr, int ld_cvr,
              const float* w_cvr, const float* b_cvr, int n_cvr, const float* y, const float* z, int B, float inv_b, float wgt,
              float* y_ctr, float* y_cvr, float* pctr, float* pcvr, float* pctcvr, float* dy_ctr, float* dy_cvr, float* loss_shards,
              hipStream_t st);
int add_inplace(float* a, const float* b, int64_t n, hipStream_t st);

// din_att.hip: DIN attention pooling over CSR batches (DIN.py:152-172)
int att_build_x(const float* emb, int64_t rows, int K, const int32_t* ids, const float* weights, const int32_t* entry_off,
                const int32_t* pair_ad, int nnz, const float* x, int ld, float* X, hipStream_t st, const LagView* lag = nullptr);
int att_pool_fwd(const int32_t* offsets, const int32_t* ids, const int32_t* pair_ad, const float* sc, int n_seg, int S, int K,
                 const float* X, float* att, float* x, int ld, hipStream_t st);
int att_bwd_scores(const int32_t* ids, const int32_t* entry_off, const int32_t* pair_ad, int nnz, int K, const float* dx, int ld,
                   const float* X, const float* att, float* dsc, hipStream_t st);
int att_bwd_combine(const int32_t* offsets, const int32_t* ids, const int32_t* entry_off, const int32_t* pair_ad, int nnz, int n_seg,
                    int S, int K, float* dx, int ld, const float* dX, const float* att, float* dub, int32_t* goff, hipStream_t st);

// afm_fused.hip: AFM attention network fused over the pair rows (no [B*P, A] hidden activations in HBM)
bool afm_fused_supported(int K, int A);
int afm_att_fwd(const float* pp, const float* W, const float* ba, const float* wo, const float* bo, int64_t rows, int K, int A, float* sc,
                hipStream_t st);
constexpr int AFM_SLABS = 64;        // gradient slabs of the fused attention backward
int afm_att_bwd(const float* pp, const float* W, const float* ba, const float* wo, const float* dsc, int64_t rows, int K, int A, float* dpp2,
                float* dW_part, int64_t dW_stride, float* dba_part, int64_t dba_stride, float* dwo_part, int64_t dwo_stride, float* dbo_part,
                int64_t dbo_stride, int n_slabs, hipStream_t st);

// DeepMVM product layer (DeepMVM.py:144-150)
int mvm_fwd(const float* e, int e_ld, const float* mb, int B, int F, int K, float* xm, hipStream_t st);
int mvm_bwd(const float* e, int e_ld, const float* mb, const float* dxm, int B, int F, int K, float* dE, int de_ld, float* dmb_part,
            int64_t part_stride, int splits, hipStream_t st);

}  // namespace dctr
