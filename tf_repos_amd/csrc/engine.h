// Engine state shared by engine.hip (MLP-family models) and afm.hip (attention model).
#pragma once
#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "ops.h"

namespace dctr {

struct Param {
    std::string name;
    int rank = 1;
    int64_t dims[4] = {1, 1, 1, 1};
    int64_t n = 0;
    bool is_table = false;
    float* ptr = nullptr;        // device
    float* s0 = nullptr;
    float* s1 = nullptr;
    int64_t arena_off = 0;       // dense params: offset in the arena
    int64_t padded = 0;
    int64_t part_off = 0;        // offset of the first partial slab in `parts`
    int n_part = 1;
    float l2 = 0.f;
    bool frozen = false;         // not trainable (BN moving statistics): the optimizer skips its blocks
    // --embedding_size values the kernels do not take (K/4 not a power of two): the engine runs on K padded to the next such size,
    // the padded columns / rows held at zero (they stay zero: every gradient into them is a product with a zero), and the
    // parameter is shown to the caller in its logical shape.  kseg maps runs of `blk` logical rows (of row_elems floats) to their
    // physical place; empty = the layout does not depend on K.
    struct KSeg { int64_t log_off, phys_off, n_blocks, blk, stride; };
    std::vector<KSeg> kseg;
    int64_t k_row_elems = 1;
    int64_t log_n = 0;
    int64_t log_dims[4] = {1, 1, 1, 1};
};

struct Fc {
    int in = 0, out = 0;
    int w = -1, b = -1;          // indices into params
    float keep = 1.f;
    uint64_t salt = 0;           // dropout site of this layer's output (DCTR_DROPOUT_SITE_MLP / _MLP2 of include/deepctr_hip.h)
    int splits = 1;
    int bn_beta = -1, bn_gamma = -1, bn_mm = -1, bn_mv = -1;   // batch_norm after this layer's ReLU (DeepFM.py:159-160)
    int last = -1;               // last parameter index of this layer (biases, or the BN moving variance)
    // dctr_config.gemm_mode = 1: the weight pre-split into three bf16 planes in the forms the forward and the dgrad product read
    // (gemm_dr3.hip); rewritten behind every optimizer step of the layer.  w_epoch counts writes of the weight, p_epoch is the
    // epoch the planes were made from: a product that finds them different refreshes first (parameter writes from the host).
    unsigned *wp_fwd = nullptr, *wp_dgr = nullptr;
    int64_t fwd_plane = 0, dgr_plane = 0;
    uint64_t w_epoch = 1, p_epoch = 0;
};

}  // namespace dctr

using dctr::Param;
using dctr::Fc;
using dctr::Group;
using dctr::OptBlockMeta;
using dctr::StepState;

struct dctr_engine {
    dctr_config cfg{};
    int F = 0, K = 0, P = 0, D = 0;      // D = F*K (K = the physical embedding width the kernels run on)
    int K_log = 0;                       // --embedding_size as the caller gave it (<= K; see Param::kseg)
    int64_t rows = 0;
    int MB = 0;
    int Din = 0, Din_ld = 0;
    std::vector<Param> params;
    std::map<std::string, int> index;
    std::vector<Fc> mlp;
    int p_out_w = -1, p_out_b = -1, p_bias = -1, p_cross_w = -1, p_cross_b = -1;
    int out_splits = 128;
    int gemm_mode = 0;                   // dctr_config.gemm_mode after the DCTR_GEMM_MODE override

    // tables
    float *emb = nullptr, *emb_s0 = nullptr, *emb_s1 = nullptr, *lin = nullptr, *lin_s0 = nullptr, *lin_s1 = nullptr;
    // Row strides (floats) of emb / emb_s0 / emb_s1 and of lin / lin_s0 / lin_s1.  Separate dense arrays (the default): (K, 1).
    // ROW RECORDS (table_rec != nullptr: DCTR_TABLE_RECORDS=1 in the environment when the handle is created, on an engine whose rows
    // may lag, unsharded, fixed-field batches): the six pointers point into ONE buffer of `rows` records
    //     [ theta K | w . . . | m K | w_m . . . | v K | w_v . . . ]        (the three 4-float linear groups only when the model has a linear table)
    // padded to a multiple of 16 floats (64 B) -- a row's step state is one or two adjacent 128-byte lines instead of six scattered
    // sectors.  Every kernel that touches a table row takes the two strides.  Measured in the step (profiles/r04_table_records.txt,
    // second part): the fused tail 40.7 -> 33.4 us, the catch-up gather unchanged (c2's table sits in the Infinity Cache), the sweep
    // 30.5 -> 32.2 us (+25 % bytes), the step 0.2655 -> 0.2667 ms: the tail is not what the step waits for.  Hence opt-in.
    int tab_ld = 0, lin_ld = 1;
    float* table_rec = nullptr;
    // time-blocked dense-exact sweep (lag.h): rows may lag behind global_step; row_ts stamps how far each has been advanced
    uint8_t* row_ts = nullptr;
    int lag_period = 1;             // 1 = classic sweep (every row every step)
    bool lag_suspended = false;     // dctr_time_kernel's single-stage replays do not advance global_step: they run the classic kernels
    bool lag_dirty = false;         // some rows may be behind the present: lag_flush before anything reads the tables as a whole
    bool want_loss = false;         // the step being enqueued reports its loss (needs sum theta^2 of every row)
    bool owner_lag_opt_in = false;  // the owner-side split API (dctr_table_*_packed) may let this shard's rows lag: set by the native step
                                    // driver (dist.hip), which tells every step whether its loss is read; direct callers get the classic sweep
    Group* group = nullptr;
    Group* group_alt = nullptr;     // second grouping state: owner side of the row-sharded path / the NEXT batch's ids grouped ahead
    // dctr_prefetch_ids: group_alt holds the grouping of `pre_ids` (an input slot), enqueued on s_group behind ev_tail
    const int32_t* pre_ids = nullptr;
    int pre_B = 0;
    bool pre_valid = false;
    int pre_slot = 0;               // ... of input slot `pre_slot` at generation `pre_gen`: a slot rewritten since (dctr_input_slot_rewrite,
    uint32_t pre_gen = 0;           // or a staging copy into it) no longer matches and the hint is dropped
    std::atomic<uint32_t> slot_gen[DCTR_INPUT_SLOTS] = {};
    // dctr_input_slot_fill / acquire / release: the library's own H2D leg (created on first use)
    int device = 0;
    hipStream_t s_main = nullptr;   // dctr_main_stream: a stream of the engine's own for the caller's train / predict calls (created on first use)
    hipStream_t s_copy = nullptr;
    hipEvent_t slot_filled[DCTR_INPUT_SLOTS] = {}, slot_released[DCTR_INPUT_SLOTS] = {};
    std::atomic<int> slot_fill_pending[DCTR_INPUT_SLOTS] = {}, slot_release_valid[DCTR_INPUT_SLOTS] = {};
    std::atomic<int> slot_feed_ready = {0};
    hipEvent_t ev_tail = nullptr;   // = the event of the main stream's last fork when the dense backward was enqueued (ring of 64: one step uses ~12)
    hipEvent_t last_fork_ev = nullptr;
    // The step's LAST cross-stream join, deferred (record_train): the dense side of step t -- the first layer's weight gradient, its
    // optimizer step, the re-split of its weight -- is awaited by step t + 1 behind its GATHER (forward_rest), not in front of it: the
    // gather needs the tables and the step state, not the MLP.  x_in alternates between two buffers so that the gather of step t + 1
    // does not overwrite what the weight gradient of step t is still reading.
    hipEvent_t ev_dense = nullptr;
    bool dense_pending = false;
    bool lean_step = false;         // record_train: a small-batch step with the fewest enqueue calls (one optimizer launch for every dense variable)
    bool capturing = false;         // a step is being captured into a graph by dctr_time_kernel: no deferred join (nothing may stay unjoined at EndCapture)
    float* x_in_alt = nullptr;
    hipEvent_t armed_ev = nullptr;  // stop_arm: the ring event armed for the next launch (engine.hip stop_arm / stop_fork)
    bool have_tail = false;
    // arena
    float *theta = nullptr, *as0 = nullptr, *as1 = nullptr, *gflat = nullptr, *parts = nullptr;
    int64_t arena_n = 0, parts_n = 0;
    OptBlockMeta* meta = nullptr;
    OptBlockMeta* meta_flat = nullptr;
    float* ones = nullptr;        // [max_batch*F*world] of 1.0f: the `vals` of raw row gathers / gradient segment sums
    int n_blocks = 0;
    // state
    StepState* state = nullptr;
    StepState* state_alt = nullptr;   // the NEXT step's state, prepared under the tail of the step in flight (record_train)
    float* scalars_alt = nullptr;
    bool state_ready = false;
    StepState h_state{};
    float* scalars = nullptr;     // [0] xent sum, [1] sumsq emb, [2] sumsq linear, [3] sumsq dense-l2 params
    int32_t* status = nullptr;    // [2]
    // activations
    int32_t* ids = nullptr;       // current input slot (aliases slot_*[cur_slot])
    int32_t* slot_ids[DCTR_INPUT_SLOTS] = {};
    float* slot_vals[DCTR_INPUT_SLOTS] = {};
    float* slot_labels[DCTR_INPUT_SLOTS] = {};
    int cur_slot = 0;
    bool head_did_out_bwd = false;   // the fused head kernel already produced the output layer's backward
    float *vals = nullptr, *labels = nullptr;
    float *x_in = nullptr, *dx_in = nullptr, *e_buf = nullptr, *S = nullptr, *yw = nullptr, *yv = nullptr;
    dctr::BnSync bn_sync;            // batch_norm under data-parallel ranks: cross-rank sum of the column sums (dctr_set_stat_sync)
    // Outer-PNN with the pair products formed inside the first layer's GEMMs (gemm_dr.h DR_AGEN_*)
    bool opnn_fused = false;
    int* opnn_pairs = nullptr;       // [P] i << 16 | j
    float* opnn_ws = nullptr;        // forward split-K slabs
    float* opnn_dop = nullptr;       // [MB, P K K] d(pair products)
    float *yd = nullptr, *y = nullptr, *prob = nullptr, *dy = nullptr;
    std::vector<float*> h, dh;
    bool bn = false;
    std::vector<float*> hbn, bn_stats;   // batch_norm: normalised (+dropout) layer outputs, batch mean / invstd per layer
    float* bn_scratch = nullptr;
    float *xs = nullptr, *xlw = nullptr, *dxL = nullptr, *cross_scratch = nullptr;
    float* e = nullptr;           // alias: where the scaled embeddings live
    int e_ld = 0;
    // graphs
    std::map<int, hipGraphExec_t> train_graphs, predict_graphs;
    int last_B = 0;
    // in-step timing of the first MLP GEMM (dctr_step_timer): event pairs recorded around its launch inside the step
    bool timer_on = false;
    std::vector<hipEvent_t> timer_ev;   // pairs
    std::vector<int> timer_layer;       // MLP layer of every recorded pair
    int timer_mode = 1;                 // 1: two records around layer 0 (a bracket), 2: every forward layer's own dispatch events
    bool timer_step = false;            // this step is a timed one (every 32nd)
    size_t timer_n = 0;
    uint64_t timer_tick = 0;
    hipStream_t s_group = nullptr, s_wgrad = nullptr;   // side streams of the step DAG
    hipStream_t s_opt = nullptr;                          // dense optimizer steps behind the deferred weight gradients (wgrad_late)
    hipStream_t s_afm = nullptr;                          // AFM's unfused path: the second lane of its chunked passes (afm.hip)
    bool opt_pending = false;                             // s_opt holds work of this step: joined at the end of record_train
    int64_t* auc_counts = nullptr;   // [4*200] tp,fn,tn,fp per threshold (tf.metrics.auc)
    float* eval_scalars = nullptr;   // [0] sum xent over the eval set, [1..] scratch
    int64_t eval_examples = 0;
    std::vector<hipEvent_t> events;
    size_t ev_next = 0;

    // canned-estimator models (wide_n_deep.py): dense numeric inputs + a linear side with its own optimizer
    bool wnd = false, wnd_wide = false, wnd_deep = false;
    int n_dense = 0;
    const float* dense = nullptr;    // [B, n_dense] inputs of the next call (dctr_set_dense_input), caller-owned
    int p_lin_dense = -1;

    // DeepMVM (DeepMVM.py:144-150): x_mvm = prod_f (e_f + mvm_b_f)
    int p_mvm_b = -1;
    float *xmvm = nullptr, *dxmvm = nullptr;      // [MB, K]

    // CSR (multi-hot) models DIN / ESMM (dctr_train_step_csr); ESMM's second tower (CVR) lives in the *2 members and is
    // swapped with the primary members (CTR) around the shared MLP forward / backward code
    bool csr = false;
    int64_t max_entries = 0;
    int32_t* entry_off = nullptr;    // [max_entries] float4 offset of every entry's slot in x_in / dx_in
    std::vector<Fc> mlp2;
    std::vector<float*> h2, dh2, hbn2, bn_stats2;
    int p_out2_w = -1, p_out2_b = -1;
    float *dx_in2 = nullptr, *dy2 = nullptr, *y2 = nullptr, *prob2 = nullptr, *prob3 = nullptr;
    // DIN attention pooling (din_att.hip): the attention MLP is the *2 tower, run over the batch's nnz entry rows
    bool att_on = false;
    int32_t* pair_ad = nullptr;      // [F] ad slot paired with each user slot, -1 elsewhere
    float *x_att = nullptr, *att_sc = nullptr, *att_w = nullptr;     // X [max_entries, 3K]; scores, sigmoid(scores) [max_entries]
    float* dub = nullptr;            // [max_entries, K] per-entry gradient rows; lives behind dx_in in ONE allocation
    int32_t* entry_goff = nullptr;   // [max_entries] where the table backward reads each entry's gradient row
    int x_att_ld = 0;

    // AFM (afm.hip)
    int A = 0;                       // attention layer width
    int p_att_w = -1, p_att_b = -1, p_ao_w = -1, p_ao_b = -1;
    int att_splits = 1, ao_splits = 1024;
    std::vector<Fc> att_fc;          // the attention network's hidden layers (AFM.py:143-145); [0] = p_att_w / p_att_b
    std::vector<float*> ahs, dahs;   // their activations / gradients over the B*P pair rows (unfused path); ah / dah = the last ones
    bool afm_fused = false;          // attention network fused over the pair rows (afm_fused.hip)
    float keep_att = 1.f, keep_emb = 1.f;
    float* sc_parts = nullptr;      // AFM: per-column-slab partial score dots out of the last attention product's epilogue [2][MB * P]
    void* ts_planes = nullptr;      // AFM, gemm_mode split: the attention weight's bf16 planes for the tall products (gemm_ts.hip), forward | input gradient
    void* ts_sign = nullptr;        // ... the sign words of the attention layer's output (gemm_ts.h bits_out), [MB * P] x 32 bytes
    bool ts_sign_ready = false;     // ... written by this step's forward
    bool afm_ts_wgrad = false;      // ... and the gated weight gradient is gemm_ts.hip's (the attention layer declared TS_WGRAD_SLABS slabs)
    bool afm_ah_skipped = false;    // ... nor the attention layer's output: both gradient products work from the sign words
    bool afm_pp_skipped = false;    // ... this step's forward did not write the pair tensor (its readers form the rows from the embeddings)
    bool ts_dgr_ready = false;      // ... the input gradient's planes were written by this step's forward (ts_prepare)
    bool afm_gate_slabs = false;    // AFM: attention_out's gradient slabs are laid out for the gated weight gradient (one per batch split)
    float *pairp = nullptr, *dpairp2 = nullptr, *ah = nullptr, *dah = nullptr, *sc = nullptr, *dsc = nullptr,
          *att = nullptr, *dE_buf = nullptr;
    int16_t *pair_i = nullptr, *pair_j = nullptr;
    // where dL/de lives for the table backward (dx_in for the MLP-family models, dE_buf for AFM)
    float* dE = nullptr;
    int dE_ld = 0;

    float* pp(int i) { return params[i].ptr; }
    float* part(int i) { return parts + params[i].part_off; }
};


// shared between engine.hip and afm.hip
int engine_add_param(dctr_engine* E, const std::string& name, std::initializer_list<int64_t> dims, bool table, int n_part, float l2);
int fork(dctr_engine* E, hipStream_t from, hipStream_t to);
void stop_arm(dctr_engine* E);
int stop_record(dctr_engine* E, hipStream_t from, hipEvent_t* ev);
int stop_fork(dctr_engine* E, hipStream_t from, hipStream_t to);
// row-sharded path (engine.hip), used by the native step driver (dist.hip)
int sharded_forward_backward(dctr_engine* E, const float* d_rows, int n_rows, const int32_t* d_idx, const float* d_vals,
                             const float* d_labels, int B, int global_batch, bool train, bool join_wgrad, hipStream_t st);
int afm_declare_params(dctr_engine* E);
int afm_alloc(dctr_engine* E);
void afm_free(dctr_engine* E);
int afm_forward(dctr_engine* E, int B, bool train, hipStream_t st);
int afm_backward(dctr_engine* E, int B, hipStream_t st, hipStream_t sw);
int afm_interaction_backward(dctr_engine* E, int B, hipStream_t st, hipStream_t sw);
