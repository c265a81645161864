// The engine: owns the tables, the dense-parameter arena, optimizer slots and activations of ONE rank, and runs
// the whole model_fn step (DeepFM.py:125-213 and the PNN/NFM/DCN variants) as a fixed sequence of HIP kernels,
// captured once per batch size into a hipGraph so a step costs one graph launch.
//
// HBM layout (all f32):
//   tables    emb [rows,K] + 2 optimizer slots, linear [rows] + 2 slots      (rows = ceil((V - rank)/world))
//   arena     every dense variable padded to OPT_BLOCK floats, one flat theta/slot0/slot1 triple
//   parts     gradient partial slabs (split-K wgrad / column sums); the optimizer kernel sums them
//   acts      x_in [B, Din_ld] (the scaled embeddings e are its first F*K columns), h_i [B,H_i], dh_i, ...
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "ops.h"
#include <functional>

#include "engine.h"
#include "lag.h"

using namespace dctr;

int engine_add_param(dctr_engine* E, const std::string& name, std::initializer_list<int64_t> dims, bool table, int n_part, float l2) {
    Param p;
    p.name = name;
    p.rank = (int)dims.size();
    int i = 0;
    p.n = 1;
    for (auto d : dims) { p.dims[i++] = d; p.n *= d; }
    p.is_table = table;
    p.n_part = n_part;
    p.l2 = l2;
    E->index[name] = (int)E->params.size();
    E->params.push_back(p);
    return (int)E->params.size() - 1;
}

// records "to waits for everything enqueued on from so far" (works eagerly and under stream capture)
int fork(dctr_engine* E, hipStream_t from, hipStream_t to) {
    if (from == to) return DCTR_OK;
    hipEvent_t ev = E->events[E->ev_next++ % E->events.size()];
    E->last_fork_ev = ev;
    DCTR_HIP_CHECK(hipEventRecord(ev, from));
    DCTR_HIP_CHECK(hipStreamWaitEvent(to, ev, 0));
    return DCTR_OK;
}

// ---- records that ride on a kernel launch (common.h arm_stop_event).  Usage in the step:
//     stop_arm(E);  <enqueue the op whose LAST kernel the fork depends on>;  stop_fork(E, st, to)  /  stop_record(E, st, &ev)
// If the op's launch site took the armed event (the direct GEMM kernels and the fused head do), the event completes with that
// kernel and no barrier packet is enqueued on `st`; otherwise it is recorded the ordinary way.  A/B knob DCTR_STOP_EVENTS=0.
static bool stop_events_on() {
    static const bool off = [] { const char* v = getenv("DCTR_STOP_EVENTS"); return v != nullptr && v[0] == '0'; }();
    return !off;
}
void stop_arm(dctr_engine* E) {
    if (!stop_events_on() || E->cfg.use_graph) { E->armed_ev = nullptr; return; }
    E->armed_ev = E->events[E->ev_next++ % E->events.size()];
    arm_stop_event(E->armed_ev);
}
// -> the event that covers everything enqueued on `from` up to and including the armed op
int stop_record(dctr_engine* E, hipStream_t from, hipEvent_t* ev) {
    if (E->armed_ev != nullptr && !stop_event_pending()) {      // taken: bound to the op's kernel
        *ev = E->armed_ev;
    } else {
        disarm_stop_event();
        *ev = E->armed_ev != nullptr ? E->armed_ev : E->events[E->ev_next++ % E->events.size()];
        DCTR_HIP_CHECK(hipEventRecord(*ev, from));
    }
    E->armed_ev = nullptr;
    return DCTR_OK;
}
int stop_fork(dctr_engine* E, hipStream_t from, hipStream_t to) {
    if (from == to) { disarm_stop_event(); E->armed_ev = nullptr; return DCTR_OK; }
    hipEvent_t ev = nullptr;
    DCTR_TRY(stop_record(E, from, &ev));
    E->last_fork_ev = ev;
    DCTR_HIP_CHECK(hipStreamWaitEvent(to, ev, 0));
    return DCTR_OK;
}

// one record, several waiters (an event record is a barrier packet on the recording stream: ~5 us of the critical path each)
int record_on(dctr_engine* E, hipStream_t from, hipEvent_t* ev) {
    *ev = E->events[E->ev_next++ % E->events.size()];
    DCTR_HIP_CHECK(hipEventRecord(*ev, from));
    return DCTR_OK;
}

namespace {

static int add_param(dctr_engine* E, const std::string& name, std::initializer_list<int64_t> dims, bool table, int n_part, float l2) {
    return engine_add_param(E, name, dims, table, n_part, l2);
}

// DCN's cross network in a step: the lean forward / fused backward pair of interact.hip (A/B knob DCTR_DCN_LEAN=0: the op-level
// kernels, which keep every x_l and leave the parameter gradients to column-sum launches)
bool dcn_lean(const dctr_engine* E) {
    static const bool off = [] { const char* v = getenv("DCTR_DCN_LEAN"); return v != nullptr && v[0] == '0'; }();
    return !off && E->cfg.model == DCTR_MODEL_DCN && dcn_cross_lean_ok(E->D, E->cfg.cross_layers) && E->Din_ld % 4 == 0;
}

int gather_mode(const dctr_engine* E) {
    switch (E->cfg.model) {
        case DCTR_MODEL_DEEPFM: return DCTR_GATHER_FM;
        case DCTR_MODEL_NFM: return DCTR_GATHER_BI;
        default: return DCTR_GATHER_RAW;
    }
}

template <typename T>
int dmalloc(T** p, size_t n_elems, bool zero = true) {
    DCTR_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n_elems, 4) * sizeof(T)));
    if (zero) DCTR_HIP_CHECK(hipMemset(*p, 0, std::max<size_t>(n_elems, 4) * sizeof(T)));
    return DCTR_OK;
}

// in-place dropout on an arbitrary-sign tensor (NFM's bi-interaction, NFM.py:136-137): mask recomputed from the counter RNG
// (x is [rows, width] contiguous, n = rows * width; the mask index counts from this rank's first GLOBAL row, common.h dropout_row0)
__global__ void dropout_inplace_kernel(float* __restrict__ x, int64_t n, int width, float keep, const uint64_t* __restrict__ seed_ptr, uint64_t salt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i] *= dropout_scale(*seed_ptr ^ salt, dropout_row0(seed_ptr) * (uint64_t)width + (uint64_t)i, keep);
}

int dropout_inplace(float* x, int64_t n, int width, float keep, const uint64_t* seed_ptr, uint64_t salt, hipStream_t st) {
    if (n <= 0 || keep >= 1.f) return DCTR_OK;
    DCTR_LAUNCH_RIDE(dropout_inplace_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0u, st, x, n, width, keep, seed_ptr, salt);
    DCTR_LAUNCH_CHECK();
    return DCTR_OK;
}

int build(dctr_engine* E) {
    const dctr_config& c = E->cfg;
    DCTR_REQUIRE(c.model >= DCTR_MODEL_DEEPFM && c.model <= DCTR_MODEL_ESMM, "unknown model %d", c.model);
    E->csr = c.model == DCTR_MODEL_DIN || c.model == DCTR_MODEL_ESMM;
    if (E->csr) {
        DCTR_REQUIRE(c.shard_world == 1, "the CSR models are not row-sharded");
        DCTR_REQUIRE(c.max_entries >= 0, "max_entries must be >= 0");
        DCTR_REQUIRE(c.model != DCTR_MODEL_ESMM || (c.ctr_task_wgt >= 0.f && c.ctr_task_wgt <= 1.f), "ctr_task_wgt must be in [0,1]");
        E->max_entries = c.max_entries > 0 ? c.max_entries : (int64_t)c.max_batch * c.field_size * 8;
        E->att_on = c.model == DCTR_MODEL_DIN && c.n_att_pairs > 0;
        // (DIN.py:165 applies batch_norm_layer inside attention_unit with an undefined `train_phase`: a NameError in the script)
        DCTR_REQUIRE(!(E->att_on && c.batch_norm), "batch_norm together with DIN attention pooling is not implemented");
        if (E->att_on) {
            DCTR_REQUIRE(c.n_att_pairs <= 8, "at most 8 attention pairs");
            DCTR_REQUIRE(c.n_attention_layers >= 1 && c.n_attention_layers <= DCTR_MAX_LAYERS, "attention pooling needs 1..%d attention layers", DCTR_MAX_LAYERS);
            DCTR_REQUIRE(c.embedding_size <= 64, "attention pooling: embedding_size <= 64");
            for (int p = 0; p < c.n_att_pairs; ++p) {
                DCTR_REQUIRE(c.att_user_slot[p] >= 0 && c.att_user_slot[p] < c.field_size && c.att_ad_slot[p] >= 0 &&
                             c.att_ad_slot[p] < c.field_size && c.att_ad_slot[p] != c.att_user_slot[p], "attention pair %d: bad slots", p);
                for (int q = 0; q < c.n_att_pairs; ++q)
                    DCTR_REQUIRE(q == p || (c.att_user_slot[q] != c.att_user_slot[p] && c.att_ad_slot[q] != c.att_ad_slot[p] &&
                                            c.att_ad_slot[q] != c.att_user_slot[p]),
                                 "attention pairs must use distinct user slots and distinct ad slots, and no slot in both roles");
            }
        }
    }
    E->wnd = c.model >= DCTR_MODEL_WIDE && c.model <= DCTR_MODEL_WND;
    E->wnd_wide = c.model == DCTR_MODEL_WIDE || c.model == DCTR_MODEL_WND;
    E->wnd_deep = c.model == DCTR_MODEL_DEEP || c.model == DCTR_MODEL_WND;
    E->n_dense = E->wnd ? c.dense_size : 0;
    if (E->wnd) {
        DCTR_REQUIRE(c.dense_size >= 0 && c.dense_size <= 4096, "dense_size %d out of range", c.dense_size);
        DCTR_REQUIRE(c.table_mode == DCTR_TABLE_TOUCHED_ROWS, "canned-estimator models apply sparse (touched-rows) table updates");
        DCTR_REQUIRE(c.shard_world == 1, "canned-estimator models are not row-sharded");
        DCTR_REQUIRE(c.lin_optimizer >= DCTR_OPT_ADAM && c.lin_optimizer <= DCTR_OPT_FTRL, "bad lin_optimizer %d", c.lin_optimizer);
    }
    DCTR_REQUIRE(c.field_size > 0 && c.feature_size > 0 && c.max_batch > 0, "field_size, feature_size, max_batch must be > 0");
    DCTR_REQUIRE(c.embedding_size % 4 == 0 && c.embedding_size >= 4, "embedding_size must be a multiple of 4");
    const bool afm = c.model == DCTR_MODEL_AFM;
    const bool no_mlp = afm || c.model == DCTR_MODEL_WIDE;
    DCTR_REQUIRE(no_mlp || (c.n_deep_layers >= 1 && c.n_deep_layers <= DCTR_MAX_LAYERS), "1..%d deep layers supported", DCTR_MAX_LAYERS);
    DCTR_REQUIRE(c.shard_world >= 1 && c.shard_rank >= 0 && c.shard_rank < c.shard_world, "bad shard rank/world");
    E->bn = c.batch_norm != 0;
    if (E->bn) {
        DCTR_REQUIRE(c.model != DCTR_MODEL_AFM && (c.model < DCTR_MODEL_WIDE || E->csr),
                     "batch_norm is implemented for the MLP-family models");
        // (row-sharded tables = data-parallel ranks: the batch statistics are synchronised through dctr_set_stat_sync, which the
        //  sharded drivers install; sharded_forward_backward refuses to run without it)
        DCTR_REQUIRE(c.batch_norm_decay >= 0.f && c.batch_norm_decay <= 1.f, "batch_norm_decay must be in [0,1]");
    }
    for (int i = 0; i < (afm ? 2 : (no_mlp ? 0 : c.n_deep_layers)); ++i)
        DCTR_REQUIRE(c.keep_prob[i] > 0.f && c.keep_prob[i] <= 1.f, "dropout keep_prob[%d]=%f must be in (0,1]", i, c.keep_prob[i]);
    E->F = c.field_size; E->K = c.embedding_size; E->D = E->F * E->K; E->P = E->F * (E->F - 1) / 2; E->MB = c.max_batch;
    E->rows = (c.feature_size - c.shard_rank + c.shard_world - 1) / c.shard_world;
    const int F = E->F, K = E->K, D = E->D, P = E->P, MB = E->MB;
    switch (c.model) {
        case DCTR_MODEL_IPNN: E->Din = D + P; break;
        case DCTR_MODEL_OPNN:
            // fused first layer (gemm_dr.h DR_AGEN_*): the [B, P K K] product tensor is never written -- x_in is the flat part only
            E->opnn_fused = !no_mlp && opnn_fused_ok(F, K, c.deep_layers[0]);
            E->Din = E->opnn_fused ? D : D + P * K * K;
            break;
        case DCTR_MODEL_NFM: E->Din = K; break;
        case DCTR_MODEL_AFM: E->Din = K; break;
        default: E->Din = D + E->n_dense; break;     // canned DNN: [embeddings | numeric columns]
    }
    E->Din_ld = (int)round_up(E->Din, 4);
    const bool mvm = c.model == DCTR_MODEL_MVM;
    const bool has_lin = E->wnd ? E->wnd_wide : (c.model != DCTR_MODEL_DCN && !mvm && !E->csr);

    // ---- parameters (engine names, tf_repos_amd.checkpoint maps them to TF names)
    // A/B knob DCTR_OUT_SPLITS: slabs of the output layer's gradient = blocks of the fused head kernel (rows per block = batch / slabs)
    if (const char* v = getenv("DCTR_OUT_SPLITS")) { const int n = atoi(v); if (n >= 32 && n <= 1024) E->out_splits = n; }
    if (c.model == DCTR_MODEL_DCN) {
        E->p_cross_b = add_param(E, "cross_b", {c.cross_layers, D}, false, 32, c.l2_reg);
        E->p_cross_w = add_param(E, "cross_w", {c.cross_layers, D}, false, 32, c.l2_reg);
    } else if (mvm) {
        E->p_mvm_b = add_param(E, "mvm_b", {F, K}, false, 256, c.l2_reg);       // DeepMVM.py:118, in the loss (:197-199); 256 slabs = 256 blocks in mvm_bwd
    } else if (has_lin) {
        E->p_bias = add_param(E, "bias", {1}, false, E->out_splits, 0.f);
        add_param(E, "linear", {E->rows}, true, 1, c.l2_reg);
        if (E->wnd && E->n_dense > 0) E->p_lin_dense = add_param(E, "linear_dense", {E->n_dense}, false, E->out_splits, 0.f);
    }
    if (!E->wnd || E->wnd_deep) add_param(E, "emb", {E->rows, K}, true, 1, c.l2_reg);   // (wide-only: the table exists, unused)
    int d = E->Din;
    if (afm) DCTR_TRY(afm_declare_params(E));
    // ESMM declares two towers over the same input: "ctr_" (primary members) and "cvr_" (the *2 members)
    const char* tower_prefix[2] = {c.model == DCTR_MODEL_ESMM ? "ctr_" : "", "cvr_"};
    for (int t = 0; t < (c.model == DCTR_MODEL_ESMM ? 2 : 1); ++t) {
        std::vector<Fc>& tower = t == 0 ? E->mlp : E->mlp2;
        d = E->Din;
        for (int i = 0; i < (no_mlp ? 0 : c.n_deep_layers); ++i) {
            Fc fc;
            fc.in = d; fc.out = c.deep_layers[i]; fc.keep = c.keep_prob[i];
            fc.salt = t == 0 ? DCTR_DROPOUT_SITE_MLP(i) : DCTR_DROPOUT_SITE_MLP2(i);   // (independent draws per tower, as two nn.dropout ops are)
            DCTR_REQUIRE(fc.out > 0, "deep layer widths must be positive (got %d)", fc.out);
            fc.splits = choose_wgrad_splits(MB, fc.in, fc.out);
            if (i == 0 && E->opnn_fused) { fc.in = D + P * K * K; fc.splits = 1; }    // (the weight keeps the reference's [F K + P K K, H] shape)
            char nm[64];
            snprintf(nm, sizeof(nm), "%smlp%d/weights", tower_prefix[t], i);
            fc.w = add_param(E, nm, {fc.in, fc.out}, false, fc.splits, 0.f);
            snprintf(nm, sizeof(nm), "%smlp%d/biases", tower_prefix[t], i);
            fc.b = add_param(E, nm, {fc.out}, false, fc.splits, 0.f);
            fc.last = fc.b;
            if (E->bn) {        // variable names of contrib.layers.batch_norm under scope bn_%d (DeepFM.py:160,231-235)
                // (ESMM: scopes cvr_bn_%d / ctr_bn_%d, DeepCvrMTL.py:178,199)
                snprintf(nm, sizeof(nm), "%sbn_%d/beta", tower_prefix[t], i);            fc.bn_beta = add_param(E, nm, {fc.out}, false, 1, 0.f);
                snprintf(nm, sizeof(nm), "%sbn_%d/gamma", tower_prefix[t], i);           fc.bn_gamma = add_param(E, nm, {fc.out}, false, 1, 0.f);
                snprintf(nm, sizeof(nm), "%sbn_%d/moving_mean", tower_prefix[t], i);     fc.bn_mm = add_param(E, nm, {fc.out}, false, 1, 0.f);
                snprintf(nm, sizeof(nm), "%sbn_%d/moving_variance", tower_prefix[t], i); fc.bn_mv = add_param(E, nm, {fc.out}, false, 1, 0.f);
                E->params[fc.bn_mm].frozen = E->params[fc.bn_mv].frozen = true;
                fc.last = fc.bn_mv;
            }
            tower.push_back(fc);
            d = fc.out;
        }
        if (c.model == DCTR_MODEL_ESMM) {       // ctr_out / cvr_out (DeepCvrMTL.py:182,203)
            char nm[64];
            snprintf(nm, sizeof(nm), "%sout/weights", tower_prefix[t]);
            (t == 0 ? E->p_out_w : E->p_out2_w) = add_param(E, nm, {d, 1}, false, E->out_splits, 0.f);
            snprintf(nm, sizeof(nm), "%sout/biases", tower_prefix[t]);
            (t == 0 ? E->p_out_b : E->p_out2_b) = add_param(E, nm, {1}, false, E->out_splits, 0.f);
        }
    }
    if (E->att_on) {
        // attention MLP over [ub | ub - ax | ax]: Field-wise-Pooling-layer/att_fc%d, att_out (DIN.py:163-168)
        int da = 3 * K;
        for (int i = 0; i < c.n_attention_layers; ++i) {
            Fc fc;
            fc.in = da; fc.out = c.attention_layers[i]; fc.keep = c.keep_prob[i];
            fc.salt = DCTR_DROPOUT_SITE_MLP2(i);
            DCTR_REQUIRE(fc.out > 0, "attention layer widths must be positive (got %d)", fc.out);
            fc.splits = choose_wgrad_splits((int)std::min<int64_t>(E->max_entries, 1 << 30), fc.in, fc.out);
            char nm[64];
            snprintf(nm, sizeof(nm), "att_fc%d/weights", i);
            fc.w = add_param(E, nm, {fc.in, fc.out}, false, fc.splits, 0.f);
            snprintf(nm, sizeof(nm), "att_fc%d/biases", i);
            fc.b = add_param(E, nm, {fc.out}, false, fc.splits, 0.f);
            fc.last = fc.b;
            E->mlp2.push_back(fc);
            da = fc.out;
        }
        E->p_out2_w = add_param(E, "att_out/weights", {da, 1}, false, 1024, 0.f);
        E->p_out2_b = add_param(E, "att_out/biases", {1}, false, 1024, 0.f);
    }
    if (afm || c.model == DCTR_MODEL_WIDE || c.model == DCTR_MODEL_ESMM) {
        // AFM: output layer declared by afm_declare_params; LinearClassifier: no DNN side at all; ESMM: declared per tower above
    } else if (c.model == DCTR_MODEL_DCN) {
        E->p_out_w = add_param(E, "out_layer/weights", {D + d, 1}, false, E->out_splits, 0.f);
        E->p_out_b = add_param(E, "out_layer/biases", {1}, false, E->out_splits, 0.f);
    } else if (mvm) {       // fc([x_mvm (K) || mlp_out]) -> 1, scope DeepMVM-out/deep_out (DeepMVM.py:185-188)
        E->p_out_w = add_param(E, "deep_out/weights", {K + d, 1}, false, E->out_splits, 0.f);
        E->p_out_b = add_param(E, "deep_out/biases", {1}, false, E->out_splits, 0.f);
    } else {
        E->p_out_w = add_param(E, "deep_out/weights", {d, 1}, false, E->out_splits, 0.f);
        E->p_out_b = add_param(E, "deep_out/biases", {1}, false, E->out_splits, 0.f);
    }

    // ---- tables
    // time-blocked dense-exact sweep (lag.h): Adam on the engine's own unsharded tables, eager steps, fixed-field batches
    int lag_period_cfg = 1;
    {
        int period = c.table_sweep_period;
        if (period == 0) { const char* v = getenv("DCTR_SWEEP_PERIOD"); period = v ? atoi(v) : 8; }
        DCTR_REQUIRE(period >= 1 && period <= LAG_MAX_PERIOD, "table_sweep_period %d outside [1, %d]", period, LAG_MAX_PERIOD);
        const bool can = c.table_mode == DCTR_TABLE_DENSE_EXACT && c.optimizer == DCTR_OPT_ADAM && !E->wnd && !c.use_graph;
        lag_period_cfg = can ? period : 1;
    }
    E->tab_ld = K; E->lin_ld = 1;
    {
        const char* v = getenv("DCTR_TABLE_RECORDS");
        const bool records = v != nullptr && v[0] == '1' && lag_period_cfg > 1 && !E->csr && c.shard_world == 1 && E->K == E->K_log;
        if (records) {          // row records (engine.h): one buffer, six views
            const int grp = has_lin ? K + 4 : K;
            const int R = (int)round_up(3 * grp, 16);
            DCTR_TRY(dmalloc(&E->table_rec, (size_t)E->rows * R));
            E->emb = E->table_rec; E->emb_s0 = E->table_rec + grp; E->emb_s1 = E->table_rec + 2 * grp;
            if (has_lin) { E->lin = E->emb + K; E->lin_s0 = E->emb_s0 + K; E->lin_s1 = E->emb_s1 + K; }
            E->tab_ld = R; E->lin_ld = R;
        } else {
            DCTR_TRY(dmalloc(&E->emb, (size_t)E->rows * K));
            DCTR_TRY(dmalloc(&E->emb_s0, (size_t)E->rows * K));
            DCTR_TRY(dmalloc(&E->emb_s1, (size_t)E->rows * K));
            if (has_lin) {
                DCTR_TRY(dmalloc(&E->lin, (size_t)E->rows));
                DCTR_TRY(dmalloc(&E->lin_s0, (size_t)E->rows));
                DCTR_TRY(dmalloc(&E->lin_s1, (size_t)E->rows));
            }
        }
    }
    {
        E->lag_period = lag_period_cfg;
        if (E->lag_period > 1) DCTR_TRY(dmalloc(&E->row_ts, (size_t)E->rows));
    }
    // owner side may receive rows from every rank; CSR models group up to max_entries ids per step
    DCTR_TRY(group_create(E->rows, E->csr ? E->max_entries : (int64_t)MB * F * c.shard_world, K, &E->group));

    // ---- dense arena + partial slabs + optimizer block metadata
    int64_t off = 0, poff = 0;
    for (auto& p : E->params) {
        if (p.is_table) continue;
        p.padded = round_up(p.n, OPT_BLOCK);
        p.arena_off = off; off += p.padded;
        p.part_off = poff; poff += p.padded * p.n_part;
    }
    // the global bias' gradient (sum_b dy) equals deep_out/biases' gradient: alias its slabs instead of recomputing
    if (E->p_bias >= 0 && E->p_out_b >= 0) { E->params[E->p_bias].part_off = E->params[E->p_out_b].part_off; E->params[E->p_bias].n_part = E->params[E->p_out_b].n_part; }
    E->arena_n = off; E->parts_n = poff;
    E->n_blocks = (int)(off / OPT_BLOCK);
    // readable slack behind the arena: the MLP GEMMs run their edge tiles unchecked (gemm.hip `over`) and may read up to
    // GEMM_SLACK_ROWS rows past a weight matrix -- normally the next parameters, past the last one this padding
    size_t wslack = 0;
    for (auto& fc : E->mlp) wslack = std::max(wslack, (size_t)GEMM_SLACK_ROWS * (size_t)std::max(fc.out, fc.in));
    for (auto& fc : E->mlp2) wslack = std::max(wslack, (size_t)GEMM_SLACK_ROWS * (size_t)std::max(fc.out, fc.in));
    DCTR_TRY(dmalloc(&E->theta, (size_t)off + wslack));
    DCTR_TRY(dmalloc(&E->as0, (size_t)off));
    DCTR_TRY(dmalloc(&E->as1, (size_t)off));
    DCTR_TRY(dmalloc(&E->gflat, (size_t)off));
    DCTR_TRY(dmalloc(&E->parts, (size_t)poff));
    std::vector<OptBlockMeta> hm((size_t)E->n_blocks);
    for (auto& p : E->params) {
        if (p.is_table) {
            const bool is_emb = p.name == "emb";
            p.ptr = is_emb ? E->emb : E->lin; p.s0 = is_emb ? E->emb_s0 : E->lin_s0; p.s1 = is_emb ? E->emb_s1 : E->lin_s1;
            continue;
        }
        p.ptr = E->theta + p.arena_off; p.s0 = E->as0 + p.arena_off; p.s1 = E->as1 + p.arena_off;
        for (int64_t j = 0; j < p.padded / OPT_BLOCK; ++j) {
            OptBlockMeta& m = hm[(size_t)(p.arena_off / OPT_BLOCK + j)];
            m.part_off = p.part_off + j * OPT_BLOCK;
            m.part_stride = p.padded;
            m.n_part = p.frozen ? -1 : p.n_part;        // n_part < 0: the optimizer leaves the block alone
            m.l2 = p.l2;
        }
    }
    DCTR_TRY(dmalloc(&E->meta, hm.size(), false));
    DCTR_HIP_CHECK(hipMemcpy(E->meta, hm.data(), hm.size() * sizeof(OptBlockMeta), hipMemcpyHostToDevice));
    // second metadata table: gradients already reduced into the flat arena (after the all-reduce of the sharded path)
    std::vector<OptBlockMeta> hf(hm);
    for (size_t j = 0; j < hf.size(); ++j) { hf[j].part_off = (int64_t)j * OPT_BLOCK; hf[j].part_stride = 0; hf[j].n_part = hf[j].n_part < 0 ? -1 : 1; }
    DCTR_TRY(dmalloc(&E->meta_flat, hf.size(), false));
    DCTR_HIP_CHECK(hipMemcpy(E->meta_flat, hf.data(), hf.size() * sizeof(OptBlockMeta), hipMemcpyHostToDevice));
    {
        std::vector<float> ones((size_t)MB * F * c.shard_world, 1.0f);
        DCTR_TRY(dmalloc(&E->ones, ones.size(), false));
        DCTR_HIP_CHECK(hipMemcpy(E->ones, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
    }

    // ---- step state
    StepState s{};
    s.t = 0; s.seed = c.seed; s.seed_t = c.seed;
    s.hyper.lr = c.learning_rate; s.hyper.beta1 = 0.9f; s.hyper.beta2 = 0.999f; s.hyper.eps = 1e-8f;   // DeepFM.py:205
    s.hyper.momentum = 0.95f;                                                                          // DeepFM.py:209
    s.hyper.lr_t = c.learning_rate;
    s.hyper_lin = s.hyper;
    if (E->wnd) { s.hyper_lin.lr = c.lin_learning_rate; s.hyper_lin.lr_t = c.lin_learning_rate; }
    E->h_state = s;
    DCTR_TRY(dmalloc(&E->state, 1, false));
    DCTR_HIP_CHECK(hipMemcpy(E->state, &s, sizeof(s), hipMemcpyHostToDevice));
    DCTR_TRY(dmalloc(&E->state_alt, 1, false));
    DCTR_HIP_CHECK(hipMemcpy(E->state_alt, &s, sizeof(s), hipMemcpyHostToDevice));      // (step_state_next keeps the destination's lag_overflow: not uninitialised memory)
    DCTR_TRY(dmalloc(&E->scalars_alt, 4 * SUMSQ_SHARDS));
    DCTR_TRY(dmalloc(&E->scalars, 4 * SUMSQ_SHARDS));   // [0..63] xent shards; [64..127] emb^2 shards; [128..191] linear^2; [192..255] dense l2 params
    DCTR_TRY(dmalloc(&E->status, 2));

    // optimizer slot initial values (DeepFM.py:207 Adagrad 1e-8; Ftrl default accumulator 0.1 [TF-1.4])
    // canned estimators: tf.train.AdagradOptimizer / FtrlOptimizer defaults, accumulators start at 0.1 [TF-1.4]
    if (c.optimizer == DCTR_OPT_ADAGRAD || c.optimizer == DCTR_OPT_FTRL || E->wnd) {
        const float init = (c.optimizer == DCTR_OPT_ADAGRAD && !E->wnd) ? 1e-8f : 0.1f;
        auto fill = [&](float* p, size_t n) -> int {
            if (!p || !n) return DCTR_OK;
            std::vector<float> v(std::min<size_t>(n, (size_t)1 << 22), init);
            for (size_t o = 0; o < n; o += v.size())
                DCTR_HIP_CHECK(hipMemcpy(p + o, v.data(), std::min(v.size(), n - o) * sizeof(float), hipMemcpyHostToDevice));
            return DCTR_OK;
        };
        DCTR_TRY(fill(E->emb_s0, (size_t)E->rows * K));
        DCTR_TRY(fill(E->lin_s0, has_lin ? (size_t)E->rows : 0));
        DCTR_TRY(fill(E->as0, (size_t)E->arena_n));
    }

    {
        // The runtime multiplexes streams onto 4 hardware queues PER PRIORITY LEVEL, and which queue a stream lands on depends on
        // how many other streams the process (torch, RCCL) has touched before: side streams that share the main stream's queue
        // silently serialise behind it.  Giving each side stream its own priority level gives it a queue of its own:
        // grouping / routing = low (background work hidden under the GEMMs), weight gradients + dense update = high.
        int least = 0, greatest = 0;
        DCTR_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        auto prio = [&](const char* env, int dflt) {
            const char* v = getenv(env);
            if (v == nullptr) return dflt;
            return v[0] == 'h' ? greatest : (v[0] == 'l' ? least : 0);
        };
        // (use_graph: a graph captured across streams of different priorities replays 2.7x slower -- measured on c1 -- so the
        //  graph mode keeps plain streams)
        DCTR_HIP_CHECK(hipStreamCreateWithPriority(&E->s_group, hipStreamNonBlocking, c.use_graph ? 0 : prio("DCTR_PRIO_GROUP", least)));
        DCTR_HIP_CHECK(hipStreamCreateWithPriority(&E->s_wgrad, hipStreamNonBlocking, c.use_graph ? 0 : prio("DCTR_PRIO_WGRAD", greatest)));
        DCTR_HIP_CHECK(hipStreamCreateWithPriority(&E->s_opt, hipStreamNonBlocking, c.use_graph ? 0 : prio("DCTR_PRIO_OPT", least)));
    }
    E->events.resize(64);
    {
        // the step's events order streams of ONE device: no host ever inspects them, so the record need not flush to system scope
        // (A/B knob DCTR_EVENT_FLAGS: 0 = plain, 1 = DisableSystemFence (default), 2 = ReleaseToDevice)
        const char* ef = getenv("DCTR_EVENT_FLAGS");
        const int mode = ef ? atoi(ef) : 1;
        const unsigned flags = hipEventDisableTiming | (mode == 1 ? hipEventDisableSystemFence : (mode == 2 ? hipEventReleaseToDevice : 0u));
        for (auto& ev : E->events) DCTR_HIP_CHECK(hipEventCreateWithFlags(&ev, flags));
    }

    DCTR_TRY(dmalloc(&E->auc_counts, 3 * 800));      // (ESMM: CTR_AUC, CVR_AUC, CTCVR_AUC -- DeepCvrMTL.py:231-235)
    DCTR_TRY(dmalloc(&E->eval_scalars, 2 * SUMSQ_SHARDS));   // [0..63] xent shards over the eval set, [64] sum of squares scratch

    // ---- activations
    // DCTR_INPUT_SLOTS sets of input staging buffers: a caller that fills a slot directly (dctr_input_slot) pays no copy,
    // and each slot has its own captured graph, so batches can be staged while earlier steps run
    // (one allocation per slot, [ids | vals | labels]: a whole batch arrives in ONE host-to-device copy, dctr_input_slot_fill)
    for (int k = 0; k < DCTR_INPUT_SLOTS; ++k) {
        const size_t region = round_up((size_t)MB * F, 64);        // (floats; every array starts 256-byte aligned)
        float* base = nullptr;
        DCTR_TRY(dmalloc(&base, 2 * region + MB));
        E->slot_ids[k] = reinterpret_cast<int32_t*>(base);
        E->slot_vals[k] = base + region;
        E->slot_labels[k] = base + 2 * region;
    }
    E->ids = E->slot_ids[0]; E->vals = E->slot_vals[0]; E->labels = E->slot_labels[0];
    DCTR_TRY(dmalloc(&E->x_in, (size_t)(MB + GEMM_SLACK_ROWS) * E->Din_ld));      // (+ slack rows: see gemm.hip `over`)
    if (!c.use_graph && !E->csr) DCTR_TRY(dmalloc(&E->x_in_alt, (size_t)(MB + GEMM_SLACK_ROWS) * E->Din_ld));     // (engine.h: ev_dense)
    if (E->opnn_fused) {
        std::vector<int> pairs;
        for (int i = 0; i < F; ++i)
            for (int j = i + 1; j < F; ++j) pairs.push_back(i << 16 | j);           // PNN.py:142-146 loop order
        DCTR_TRY(dmalloc(&E->opnn_pairs, pairs.size()));
        DCTR_HIP_CHECK(hipMemcpy(E->opnn_pairs, pairs.data(), pairs.size() * sizeof(int), hipMemcpyHostToDevice));
        DCTR_TRY(dmalloc(&E->opnn_ws, (size_t)opnn_fwd_ws_floats_max(MB, c.deep_layers[0])));
        static const bool dop_env = getenv("DCTR_OPNN_DGRAD_MATERIALISE") != nullptr;       // A/B knob
        if (dop_env || !opnn_dgrad_fused_ok(K, c.deep_layers[0])) DCTR_TRY(dmalloc(&E->opnn_dop, (size_t)(MB + GEMM_SLACK_ROWS) * P * K * K, false));
    }
    {   // (attention pooling: the per-entry gradient rows `dub` sit behind dx_in so that one int32 float4 offset reaches both)
        const size_t dxn = (size_t)(MB + GEMM_SLACK_ROWS) * E->Din_ld;
        DCTR_TRY(dmalloc(&E->dx_in, dxn + (E->att_on ? (size_t)E->max_entries * K : 0)));
        if (E->att_on) E->dub = E->dx_in + dxn;
    }
    if (c.model == DCTR_MODEL_NFM || afm) {
        DCTR_TRY(dmalloc(&E->e_buf, (size_t)MB * D));
        E->e = E->e_buf; E->e_ld = D;
    } else {
        E->e = E->x_in; E->e_ld = E->Din_ld;
    }
    E->dE = E->dx_in; E->dE_ld = E->Din_ld;
    if (afm) {
        DCTR_TRY(afm_alloc(E));
        E->dE = E->dE_buf; E->dE_ld = D;
    }
    DCTR_TRY(dmalloc(&E->S, (size_t)MB * K));
    DCTR_TRY(dmalloc(&E->yw, (size_t)MB));
    DCTR_TRY(dmalloc(&E->yv, (size_t)MB));
    DCTR_TRY(dmalloc(&E->yd, (size_t)MB));
    DCTR_TRY(dmalloc(&E->y, (size_t)MB));
    DCTR_TRY(dmalloc(&E->prob, (size_t)MB));
    DCTR_TRY(dmalloc(&E->dy, (size_t)MB));
    for (auto& fc : E->mlp) {
        float *a = nullptr, *g = nullptr;
        DCTR_TRY(dmalloc(&a, (size_t)(MB + GEMM_SLACK_ROWS) * fc.out));
        DCTR_TRY(dmalloc(&g, (size_t)(MB + GEMM_SLACK_ROWS) * fc.out));
        E->h.push_back(a); E->dh.push_back(g);
        if (E->bn) {
            float *z = nullptr, *sx = nullptr;
            DCTR_TRY(dmalloc(&z, (size_t)(MB + GEMM_SLACK_ROWS) * fc.out));
            DCTR_TRY(dmalloc(&sx, (size_t)2 * fc.out));
            E->hbn.push_back(z); E->bn_stats.push_back(sx);
        }
    }
    const size_t rows2 = E->att_on ? (size_t)E->max_entries : (size_t)MB;        // the attention MLP runs over entry rows
    for (auto& fc : E->mlp2) {
        float *a = nullptr, *g = nullptr;
        DCTR_TRY(dmalloc(&a, (rows2 + GEMM_SLACK_ROWS) * fc.out));
        DCTR_TRY(dmalloc(&g, (rows2 + GEMM_SLACK_ROWS) * fc.out));
        E->h2.push_back(a); E->dh2.push_back(g);
        if (E->bn && !E->att_on) {
            float *z = nullptr, *sx = nullptr;
            DCTR_TRY(dmalloc(&z, (rows2 + GEMM_SLACK_ROWS) * fc.out));
            DCTR_TRY(dmalloc(&sx, (size_t)2 * fc.out));
            E->hbn2.push_back(z); E->bn_stats2.push_back(sx);
        }
    }
    if (E->att_on) {
        E->x_att_ld = 3 * K;
        DCTR_TRY(dmalloc(&E->x_att, (rows2 + GEMM_SLACK_ROWS) * E->x_att_ld));
        DCTR_TRY(dmalloc(&E->dx_in2, (rows2 + GEMM_SLACK_ROWS) * E->x_att_ld));
        DCTR_TRY(dmalloc(&E->att_sc, rows2));
        DCTR_TRY(dmalloc(&E->att_w, rows2));
        DCTR_TRY(dmalloc(&E->dy2, rows2));
        DCTR_TRY(dmalloc(&E->entry_goff, rows2));
        std::vector<int32_t> pa((size_t)F, -1);
        for (int p = 0; p < c.n_att_pairs; ++p) pa[(size_t)c.att_user_slot[p]] = c.att_ad_slot[p];
        DCTR_TRY(dmalloc(&E->pair_ad, pa.size(), false));
        DCTR_HIP_CHECK(hipMemcpy(E->pair_ad, pa.data(), pa.size() * 4, hipMemcpyHostToDevice));
    }
    if (E->csr) {
        DCTR_TRY(dmalloc(&E->entry_off, (size_t)E->max_entries));
        if (c.model == DCTR_MODEL_ESMM) {
            DCTR_TRY(dmalloc(&E->dx_in2, (size_t)(MB + GEMM_SLACK_ROWS) * E->Din_ld));
            DCTR_TRY(dmalloc(&E->dy2, (size_t)MB));
            DCTR_TRY(dmalloc(&E->y2, (size_t)MB));
            DCTR_TRY(dmalloc(&E->prob2, (size_t)MB));
            DCTR_TRY(dmalloc(&E->prob3, (size_t)MB));
        }
    }
    if (E->bn) {
        int hmax = 0;
        for (auto& fc : E->mlp) hmax = std::max(hmax, fc.out);
        for (auto& fc : E->mlp2) hmax = std::max(hmax, fc.out);
        DCTR_TRY(dmalloc(&E->bn_scratch, (size_t)bn_scratch_floats(hmax)));
    }
    if (mvm) {
        DCTR_TRY(dmalloc(&E->xmvm, (size_t)MB * K));
        DCTR_TRY(dmalloc(&E->dxmvm, (size_t)MB * K));
    }
    if (c.model == DCTR_MODEL_DCN) {
        const int L = c.cross_layers;
        DCTR_REQUIRE(L >= 1 && L <= 16, "cross_layers must be in [1,16]");
        DCTR_TRY(dmalloc(&E->xs, (size_t)(L + 1) * MB * D));
        DCTR_TRY(dmalloc(&E->xlw, (size_t)L * MB));
        DCTR_TRY(dmalloc(&E->dxL, (size_t)MB * D));
        // ([L, B, D] + [L, B] for the op-level backward; the step's fused backward wants two [512, L D] row blocks)
        DCTR_TRY(dmalloc(&E->cross_scratch, std::max((size_t)L * MB * D + (size_t)L * MB, (size_t)2 * 512 * L * D)));
    }
    DCTR_HIP_CHECK(hipDeviceSynchronize());     // the zero-fills above ran on the null stream; callers use non-blocking streams
    return DCTR_OK;
}

// ---- forward (train=true: dropout on, DeepFM.py:161-162) ------------------------------------------------
// the gather reads (emb, lin, rows, ids): the engine's own tables, or -- in the row-sharded path -- the buffer of rows
// received from their owners with ids = positions in that buffer
// ---- dctr_config.gemm_mode = 1: the MLP weights' pre-split planes (gemm_dr3.hip) ------------------------------------------------
// fresh planes of one layer on `st` (behind whatever wrote the weight on that stream)
int wplanes_refresh(dctr_engine* E, Fc& fc, hipStream_t st) {
    if (fc.wp_fwd == nullptr) return DCTR_OK;
    DCTR_TRY(dr3_wsplit(E->pp(fc.w), fc.out, fc.in, fc.out, fc.wp_fwd, fc.wp_dgr, st));
    fc.p_epoch = fc.w_epoch;
    return DCTR_OK;
}
// ... if the weight was written since they were made (parameter writes from the host; the step's own optimizer launches refresh eagerly)
int wplanes_ensure(dctr_engine* E, Fc& fc, hipStream_t st) {
    return fc.wp_fwd != nullptr && fc.p_epoch != fc.w_epoch ? wplanes_refresh(E, fc, st) : DCTR_OK;
}
// the parameters [p_first, p_last] were just written on `st` (an optimizer launch): their layers' planes follow on the same stream
int wplanes_written(dctr_engine* E, int p_first, int p_last, hipStream_t st) {
    WsplitJob jobs[8];
    int n = 0;
    for (std::vector<Fc>* tower : {&E->mlp, &E->mlp2})
        for (Fc& fc : *tower)
            if (fc.w >= p_first && fc.w <= p_last) {
                fc.w_epoch++;
                if (fc.wp_fwd == nullptr) continue;
                if (n == 8) { DCTR_TRY(dr3_wsplit_multi(jobs, n, st)); n = 0; }          // (one launch per 8 layers)
                jobs[n++] = WsplitJob{E->pp(fc.w), fc.out, fc.in, fc.out, fc.wp_fwd, fc.wp_dgr};
                fc.p_epoch = fc.w_epoch;
            }
    return dr3_wsplit_multi(jobs, n, st);
}
// the host wrote parameters: the next product that reads a layer's planes refreshes them first
void wplanes_invalidate(dctr_engine* E) {
    for (std::vector<Fc>* tower : {&E->mlp, &E->mlp2})
        for (Fc& fc : *tower) fc.w_epoch++;
}
GemmOpt gemm_opt(const dctr_engine* E, const Fc& fc) {
    GemmOpt g;
    if (E->gemm_mode == 1 && fc.wp_fwd != nullptr) { g.mode = 1; g.w_fwd = fc.wp_fwd; g.fwd_plane = fc.fwd_plane; g.w_dgr = fc.wp_dgr; g.dgr_plane = fc.dgr_plane; }
    else if (E->gemm_mode == 1) g.mode = 1;               // (no planes: the weight gradient may still take the split kernel)
    return g;
}
static int wplanes_alloc(dctr_engine* E) {
    if (E->gemm_mode != 1) return DCTR_OK;
    for (std::vector<Fc>* tower : {&E->mlp, &E->mlp2})
        for (size_t i = 0; i < tower->size(); ++i) {
            Fc& fc = (*tower)[i];
            if (tower == &E->mlp && i == 0 && E->opnn_fused) continue;          // (its products run over sub-ranges of the weight's rows)
            if (!dr3_shape_ok(E->MB, fc.in, fc.out)) continue;
            fc.fwd_plane = dr3_fwd_plane_bytes(fc.in, fc.out);
            fc.dgr_plane = dr3_dgr_plane_bytes(fc.in, fc.out);
            DCTR_HIP_CHECK(hipMalloc(&fc.wp_fwd, (size_t)(3 * fc.fwd_plane)));
            DCTR_HIP_CHECK(hipMalloc(&fc.wp_dgr, (size_t)(3 * fc.dgr_plane)));
        }
    return DCTR_OK;
}

int gather_from(dctr_engine* E, const float* emb, const float* lin, int64_t rows, const int32_t* ids, int B, hipStream_t st, const LagView* lag = nullptr) {
    const int F = E->F, K = E->K;
    const int mode = gather_mode(E);
    float* red = mode == DCTR_GATHER_FM ? E->yv : (mode == DCTR_GATHER_BI ? E->x_in : nullptr);
    const bool own = emb == E->emb;         // (the engine's own tables may be row records: engine.h tab_ld)
    DCTR_TRY(embed_gather_strided(emb, own ? E->tab_ld : K, lin, own ? E->lin_ld : 1, rows, ids, E->vals, B, F, K, mode, E->e, E->e_ld,
                                  lin ? E->yw : nullptr, E->S, red, E->status, st, lag));
    if (E->wnd && E->n_dense > 0)       // numeric columns: appended to the DNN input, and their linear_model term added to y_w
        DCTR_TRY(wnd_dense_fwd(E->dense, E->n_dense, E->p_lin_dense >= 0 ? E->pp(E->p_lin_dense) : nullptr, B,
                               E->wnd_deep ? E->x_in : nullptr, E->Din_ld, E->D, E->yw, st));
    return DCTR_OK;
}

// after_layer0: called once the first MLP layer has been enqueued (the step uses it to start the id grouping there)
// the deferred end-of-step join (engine.h ev_dense): whatever reads a dense variable on `st` waits here first
int join_deferred(dctr_engine* E, hipStream_t st) {
    if (E->dense_pending) {
        DCTR_HIP_CHECK(hipStreamWaitEvent(st, E->ev_dense, 0));
        E->dense_pending = false;
    }
    return DCTR_OK;
}

int forward_rest(dctr_engine* E, int B, bool train, hipStream_t st, const std::function<int()>* after_layer0 = nullptr, int after_idx = 0) {
    DCTR_TRY(join_deferred(E, st));
    const dctr_config& c = E->cfg;
    const int F = E->F, K = E->K, D = E->D;
    if (c.model == DCTR_MODEL_AFM) return afm_forward(E, B, train, st);
    const uint64_t* seedp = &E->state->seed_t;
    if (c.model == DCTR_MODEL_IPNN) DCTR_TRY(pnn_inner_fwd(E->e, E->e_ld, B, F, K, E->x_in + D, E->Din_ld, st));
    if (c.model == DCTR_MODEL_OPNN && !E->opnn_fused) DCTR_TRY(pnn_outer_fwd(E->e, E->e_ld, B, F, K, E->x_in + D, E->Din_ld, st));
    if (c.model == DCTR_MODEL_NFM && train) DCTR_TRY(dropout_inplace(E->x_in, (int64_t)B * K, K, c.keep_prob[0], seedp, DCTR_DROPOUT_SITE_NFM_BI, st));   // NFM.py:136-137
    if (c.model == DCTR_MODEL_DCN)
    {
        if (dcn_lean(E)) {
            // (interact.hip: x_L and s only)
            DCTR_TRY(dcn_cross_fwd_lean(E->x_in, E->Din_ld, E->pp(E->p_cross_w), E->pp(E->p_cross_b), B, D, c.cross_layers,
                                        E->xs + (size_t)c.cross_layers * B * D, E->xlw, st));
        } else
        DCTR_TRY(dcn_cross_fwd(E->x_in, E->Din_ld, E->pp(E->p_cross_w), E->pp(E->p_cross_b), B, D, c.cross_layers, E->xs, E->xlw, st));
    }
    if (c.model == DCTR_MODEL_MVM) DCTR_TRY(mvm_fwd(E->e, E->e_ld, E->pp(E->p_mvm_b), B, F, K, E->xmvm, st));
    const float* x = E->x_in;
    int ldx = E->Din_ld;
    for (size_t i = 0; i < E->mlp.size(); ++i) {
        Fc& fc = E->mlp[i];
        DCTR_TRY(wplanes_ensure(E, fc, st));
        const GemmOpt go = gemm_opt(E, fc);
        // relu(x W + b) [-> batch_norm] -> dropout (DeepFM.py:156-162): without BN the dropout rides in the GEMM epilogue
        // (every 32nd step only: a timed step costs ~15 us more -- event records, or launches with completion signals -- which the
        //  bench's `value` should not carry: 0.5 us per step on average)
        if (i == 0) E->timer_step = E->timer_on && train && (E->timer_tick++ % 32) == 3;      // (not the very first step after the enable: it fills the pipeline)
        const bool room = E->timer_n + 2 <= E->timer_ev.size();
        const bool timed = E->timer_step && room && E->timer_mode == 1 && i == 0;
        // mode 2: the launch carries its own start / stop events (common.h arm_timer_events) -- the interval is the dispatch alone
        const bool timed2 = E->timer_step && room && E->timer_mode == 2 && !E->bn && !(i == 0 && E->opnn_fused);
        if (timed) DCTR_HIP_CHECK(hipEventRecord(E->timer_ev[E->timer_n], st));
        const bool fork_here = (int)i == std::min(after_idx, (int)E->mlp.size() - 1) && after_layer0 != nullptr;
        if (fork_here && !E->bn && !(i == 0 && E->opnn_fused) && !timed2) stop_arm(E);      // (the fork behind this layer rides on its GEMM launch)
        if (timed2) arm_timer_events(E->timer_ev[E->timer_n], E->timer_ev[E->timer_n + 1]);
        if (i == 0 && E->opnn_fused) {
            // flat rows of W0 as an ordinary product (raw sums), then the pair-product rows with A formed in registers + bias/ReLU/dropout
            DCTR_TRY(fc_fwd(x, ldx, E->pp(fc.w), nullptr, E->h[0], fc.out, B, D, fc.out, 0, 1.f, nullptr, 0, st, 1));
            DCTR_TRY(opnn_outer_fwd(E->e, E->e_ld, B, F, K, E->opnn_pairs, E->pp(fc.w) + (size_t)D * fc.out, E->pp(fc.b), E->h[0], fc.out, fc.out, 1,
                                    (train && !E->bn) ? fc.keep : 1.f, seedp, fc.salt, E->opnn_ws, st));
        } else
        DCTR_TRY(fc_fwd(x, ldx, E->pp(fc.w), E->pp(fc.b), E->h[i], fc.out, B, fc.in, fc.out, 1, (train && !E->bn) ? fc.keep : 1.f,
                        seedp, fc.salt, st, 1, &go));
        if (timed) { DCTR_HIP_CHECK(hipEventRecord(E->timer_ev[E->timer_n + 1], st)); E->timer_layer.push_back(0); E->timer_n += 2; }
        if (timed2) {
            if (timer_events_pending()) disarm_timer_events();          // (a launch site that does not carry events: this launch is not timed)
            else { E->timer_layer.push_back((int)i); E->timer_n += 2; }
        }
        if (fork_here) DCTR_TRY((*after_layer0)());
        x = E->h[i]; ldx = fc.out;
        if (E->bn) {
            DCTR_TRY(bn_forward(E->h[i], fc.out, B, fc.out, train, 1e-3f, c.batch_norm_decay, E->pp(fc.bn_gamma), E->pp(fc.bn_beta),
                                E->pp(fc.bn_mm), E->pp(fc.bn_mv), fc.keep, seedp, fc.salt, E->bn_stats[i], E->bn_scratch,
                                E->hbn[i], fc.out, st, E->bn_sync.world > 1 ? &E->bn_sync : nullptr, c.batch_norm_biased_moving_variance == 0));
            x = E->hbn[i];
        }
    }
    return DCTR_OK;     // the [H -> 1] output layer is fused into the head kernel (head())
}

int forward_gather(dctr_engine* E, int B, hipStream_t st) { return gather_from(E, E->emb, E->lin, E->rows, E->ids, B, st); }

// the gather of a TRAINING step whose table rows may lag (lag.h): the step's state (t, lr history) must be in place on `st`.
// A step that reports its loss first brings every row to t-1 -- its l2 term needs sum theta^2 over the whole table, which the
// flush accumulates.
int forward_gather_train(dctr_engine* E, int B, hipStream_t st);
int lag_flush_tables(dctr_engine* E, hipStream_t st, int offset, bool with_sums);
bool lag_on(const dctr_engine* E);
bool owner_lag(const dctr_engine* E);
LagView lag_view(const dctr_engine* E);

int forward_gather_train(dctr_engine* E, int B, hipStream_t st) {
    if (!lag_on(E)) return forward_gather(E, B, st);
    if (E->want_loss) DCTR_TRY(lag_flush_tables(E, st, -1, true));
    const LagView L = lag_view(E);
    return gather_from(E, E->emb, E->lin, E->rows, E->ids, B, st, &L);
}

int forward(dctr_engine* E, int B, bool train, hipStream_t st) {
    // (predict / eval right behind a train step: the gather below writes x_in, which that step's deferred first-layer weight gradient may
    //  still be reading on the side stream -- record_train alternates the buffer, this path does not)
    DCTR_TRY(join_deferred(E, st));
    DCTR_TRY(forward_gather(E, B, st));
    return forward_rest(E, B, train, st);
}

// output layer ([H -> 1], DeepFM.py:165-167; DCN's fc([x_L || mlp_out]), DCN.py:179-183; AFM's fc(K -> 1), AFM.py:160-162) fused
// with logit / sigmoid / xent / dy.  `loss_shards`: SUMSQ_SHARDS floats receiving the xent sum.
int head(dctr_engine* E, int B, int global_batch, bool with_labels, hipStream_t st, float* loss_shards = nullptr, bool fuse_out_bwd = false) {
    const dctr_config& c = E->cfg;
    const float* bias = E->p_bias >= 0 ? E->pp(E->p_bias) : nullptr;
    const float* yw = E->lin ? E->yw : nullptr;
    const float* yv = c.model == DCTR_MODEL_DEEPFM ? E->yv : nullptr;
    const float* wout = E->p_out_w >= 0 ? E->pp(E->p_out_w) : nullptr;
    const float *x1, *x2 = nullptr, *w2 = nullptr;
    int ld1, n1, ld2 = 0, n2 = 0;
    if (c.loss_sum) global_batch = 1;           // canned heads: SUM over the batch, dy = prob - label
    if (c.model == DCTR_MODEL_WIDE) {           // LinearClassifier: logits = bias + y_w, no output layer
        E->head_did_out_bwd = false;
        float* ls = loss_shards ? loss_shards : E->scalars;
        return loss_head(bias, yw, nullptr, nullptr, with_labels ? E->labels : nullptr, B, 1.0f / (float)global_batch, E->y, E->prob,
                         with_labels ? E->dy : nullptr, with_labels ? ls : nullptr, st);
    }
    if (c.model == DCTR_MODEL_AFM) {
        x1 = E->x_in; ld1 = E->Din_ld; n1 = E->K;
    } else if (c.model == DCTR_MODEL_DCN) {
        x1 = E->xs + (size_t)c.cross_layers * B * E->D; ld1 = E->D; n1 = E->D;      // xs is laid out [L+1, B, D] for the current B
        x2 = E->bn ? E->hbn.back() : E->h.back(); ld2 = E->mlp.back().out; n2 = ld2; w2 = wout + E->D;
    } else if (c.model == DCTR_MODEL_MVM) {
        x1 = E->xmvm; ld1 = E->K; n1 = E->K;
        x2 = E->bn ? E->hbn.back() : E->h.back(); ld2 = E->mlp.back().out; n2 = ld2; w2 = wout + E->K;
    } else {
        x1 = E->bn ? E->hbn.back() : E->h.back(); ld1 = E->mlp.back().out; n1 = ld1;
    }
    const int mask_last = E->bn ? 0 : 1;        // with BN the layer's ReLU/dropout backward happens in bn_backward
    if (loss_shards == nullptr) loss_shards = E->scalars;
    E->head_did_out_bwd = false;

    if (fuse_out_bwd && with_labels && c.model != DCTR_MODEL_AFM) {
        // one launch for output layer forward + loss head + output layer backward (dh_last / dxL, dW and db partial slabs)
        const Param& pw = E->params[E->p_out_w];
        const Param& pb = E->params[E->p_out_b];
        const float keep_last = E->mlp.back().keep;
        int rc;
        if (c.model == DCTR_MODEL_DCN || c.model == DCTR_MODEL_MVM)
            rc = head_out_bwd(x1, ld1, wout, n1, 0, c.model == DCTR_MODEL_DCN ? E->dxL : E->dxmvm, ld1, x2, ld2, w2, n2, mask_last, E->dh.back(), ld2, E->pp(E->p_out_b), bias, yw, yv,
                              E->labels, B, 1.0f / (float)global_batch, keep_last, pw.n_part, E->yd, E->y, E->prob, E->dy, loss_shards,
                              E->part(E->p_out_w), pw.padded, E->part(E->p_out_b), pb.padded, st);
        else
            rc = head_out_bwd(x1, ld1, wout, n1, mask_last, E->dh.back(), ld1, nullptr, 0, nullptr, 0, 0, nullptr, 0, E->pp(E->p_out_b), bias, yw, yv,
                              E->labels, B, 1.0f / (float)global_batch, keep_last, pw.n_part, E->yd, E->y, E->prob, E->dy, loss_shards,
                              E->part(E->p_out_w), pw.padded, E->part(E->p_out_b), pb.padded, st);
        if (rc == DCTR_OK) { E->head_did_out_bwd = true; return DCTR_OK; }
        if (rc != DCTR_ERR_UNSUPPORTED) return rc;
    }
    return head_fused(x1, ld1, wout, n1, x2, ld2, w2, n2, E->pp(E->p_out_b), bias, yw, yv, with_labels ? E->labels : nullptr, B,
                      1.0f / (float)global_batch, E->yd, E->y, E->prob, with_labels ? E->dy : nullptr,
                      with_labels ? loss_shards : nullptr, st);
}

// ---- backward through head + MLP + interaction: leaves dL/de in dx_in (or the BI coefficient for NFM) ----------
// st: critical path (dgrad chain); sw: side stream for the weight gradients (independent of the dgrad chain)
// optimizer over the arena blocks of parameters [first, last] (inclusive, consecutive in the arena)
int opt_dense_range(dctr_engine* E, int p_first, int p_last, hipStream_t st, bool lin_side = false) {
    const Param& a = E->params[p_first];
    const Param& b = E->params[p_last];
    const int64_t off = a.arena_off;
    const int nb = (int)((b.arena_off + b.padded - off) / OPT_BLOCK);
    const int kind = lin_side ? E->cfg.lin_optimizer : E->cfg.optimizer;
    DCTR_TRY(opt_dense_arena(kind, lin_side ? &E->state->hyper_lin : &E->state->hyper, lin_side ? E->h_state.hyper_lin : E->h_state.hyper,
                             E->theta + off, E->as0 + off, E->as1 + off, E->parts, E->meta + off / OPT_BLOCK, nb, nullptr, 1,
                             E->scalars + 3 * SUMSQ_SHARDS, st));
    return E->gemm_mode == 1 ? wplanes_written(E, p_first, p_last, st) : DCTR_OK;
}

// d(pair products) -> dL/de for the fused Outer-PNN first layer: dOP[b][(p,a,c)] = sum_h dh0[b][h] W0[F K + (p,a,c)][h], then
// de_i[a] += sum_c dOP e_j[c], de_j[c] += sum_a dOP e_i[a].  dx_in already holds the flat rows' share.
int opnn_outer_dgrad(dctr_engine* E, int B, hipStream_t st) {
    const Fc& fc = E->mlp[0];
    const int F = E->F, K = E->K, D = E->D;
    const int64_t L = (int64_t)E->P * K * K;
    if (E->opnn_dop == nullptr)
        return opnn_outer_dgrad_fused(E->dh[0], fc.out, fc.out, E->pp(fc.w) + (size_t)D * fc.out, E->e, E->e_ld, E->opnn_pairs, B, F, K, E->dx_in, E->Din_ld, st);
    // (first layer wider than 256 or K > 64: d(pair products) materialised, then contracted)
    DCTR_TRY(fc_bwd_data(E->dh[0], fc.out, E->pp(fc.w) + (size_t)D * fc.out, E->opnn_dop, (int)L, B, (int)L, fc.out, nullptr, 0, 1.f, st, 1));
    return pnn_outer_bwd(E->e, E->e_ld, E->opnn_dop, L, B, F, K, E->dx_in, E->Din_ld, st);
}

// fused_opt: step each MLP layer's weights on the side stream as soon as BOTH its wgrad (same stream) and its dgrad (which
// still reads the old weights, other stream) are done -- the dense optimizer then costs nothing at the end of the step
// head_ev: an event recorded on st right after the head (nothing enqueued on st since): the first layer's wgrad waits on it
// instead of a record of its own
int backward_dense(dctr_engine* E, int B, hipStream_t st, hipStream_t sw, bool fused_opt = false, const hipEvent_t* head_ev = nullptr) {
    const dctr_config& c = E->cfg;
    if (c.model == DCTR_MODEL_AFM) return afm_backward(E, B, st, sw);
    const int F = E->F, K = E->K, D = E->D;
    const int H = E->mlp.back().out;
    const int nl = (int)E->mlp.size();
    const float keep_last = E->mlp.back().keep;
    const float* wout = E->pp(E->p_out_w);
    const Param& pw = E->params[E->p_out_w];
    const Param& pb = E->params[E->p_out_b];
    // output layer in one pass over h_last: dh = dy (x) w (masked), dW = h^T dy, db = sum dy (also the global bias' gradient)
    if (E->head_did_out_bwd) {
        // already done inside the fused head kernel
    } else if (c.model == DCTR_MODEL_MVM) {
        DCTR_TRY(out_layer_bwd(E->xmvm, K, E->dy, wout, B, K, pw.n_part, 0, 1.f, E->dxmvm, K, E->part(E->p_out_w), pw.padded, nullptr, 0, st));
        DCTR_TRY(out_layer_bwd(E->bn ? E->hbn.back() : E->h.back(), H, E->dy, wout + K, B, H, pw.n_part, E->bn ? 0 : 1, keep_last, E->dh.back(), H,
                               E->part(E->p_out_w) + K, pw.padded, E->part(E->p_out_b), pb.padded, st));
    } else if (c.model == DCTR_MODEL_DCN) {
        const float* xL = E->xs + (size_t)c.cross_layers * B * D;
        DCTR_TRY(out_layer_bwd(xL, D, E->dy, wout, B, D, pw.n_part, 0, 1.f, E->dxL, D, E->part(E->p_out_w), pw.padded, nullptr, 0, st));
        DCTR_TRY(out_layer_bwd(E->bn ? E->hbn.back() : E->h.back(), H, E->dy, wout + D, B, H, pw.n_part, E->bn ? 0 : 1, keep_last, E->dh.back(), H,
                               E->part(E->p_out_w) + D, pw.padded, E->part(E->p_out_b), pb.padded, st));
    } else {
        DCTR_TRY(out_layer_bwd(E->bn ? E->hbn.back() : E->h.back(), H, E->dy, wout, B, H, pw.n_part, E->bn ? 0 : 1, keep_last, E->dh.back(), H, E->part(E->p_out_w),
                               pw.padded, E->part(E->p_out_b), pb.padded, st));
    }
    const uint64_t* bn_seedp = &E->state->seed_t;
    // A/B knob DCTR_WGRAD_LATE=1: all weight gradients after the dgrad chain (measured SLOWER, 0.375 vs 0.344 ms/step at c2: beside
    // the scatter a one-block-per-CU GEMM and the scatter's many small blocks get in each other's way, 23 -> 42 us and 32 -> 47 us)
    static const bool wgrad_late_env = getenv("DCTR_WGRAD_LATE") != nullptr;
    const bool wgrad_late = wgrad_late_env && E->s_opt != nullptr && sw != st && !E->opnn_fused;
    // A/B knob DCTR_WGRAD_LATE_LAYERS=k: the weight gradients of layers 0 .. k-1 (and their optimizer steps) start behind the WHOLE dgrad
    // chain, beside the table step, with no per-layer record on st; the others stay beside their layer's dgrad
    static const int late_layers_env = getenv("DCTR_WGRAD_LATE_LAYERS") ? atoi(getenv("DCTR_WGRAD_LATE_LAYERS")) : 0;
    // (A/B knob DCTR_LEAN_WGRAD_LATE=1: a lean step -- record_train, small batches -- takes ALL its weight gradients behind the dgrad chain: one
    //  record on st and one wait on sw for the three of them instead of one pair per layer.  Measured at c1: 0.1047 vs 0.1009 ms/step -- four calls
    //  less, but 18 us of kernels more behind the table step; profiles/r06_ab_c1_lean.txt)
    static const bool lean_late = [] { const char* v = getenv("DCTR_LEAN_WGRAD_LATE"); return v != nullptr && v[0] == '1'; }();
    const int late_layers = (!wgrad_late && fused_opt && sw != st && !E->opnn_fused && !E->bn) ? ((E->lean_step && lean_late) ? nl : std::min(late_layers_env, nl)) : 0;
    // The MLP's optimizer steps as ONE launch behind the last weight gradient (and one re-split of the weights, gemm_mode 1) instead of
    // each layer's in front of the weight gradient of the layer below: with the step's last join deferred past the next gather
    // (record_train) the end of this stream is off the critical path, and two to four small launches leave the chain of full-chip
    // products.  A/B knob DCTR_OPT_TAIL=0 (the round-2 placement).  The layers' parameters are neighbours in the arena.
    static const bool opt_tail_off = [] { const char* v = getenv("DCTR_OPT_TAIL"); return v != nullptr && v[0] == '0'; }();
    const bool opt_tail = (!opt_tail_off || E->lean_step) && fused_opt && sw != st && !E->cfg.use_graph && !wgrad_late && late_layers == 0 && E->cfg.shard_world == 1;
    for (int i = nl - 1; i >= 0; --i) {
        Fc& fc = E->mlp[i];
        DCTR_TRY(wplanes_ensure(E, fc, st));
        const GemmOpt go = gemm_opt(E, fc);
        const float* x = i > 0 ? (E->bn ? E->hbn[i - 1] : E->h[i - 1]) : E->x_in;
        const int ldx = i > 0 ? E->mlp[i - 1].out : E->Din_ld;
        const Param& w = E->params[fc.w];
        const Param& b = E->params[fc.b];
        if (E->bn)      // dh[i] holds dL/d(layer output): dropout mask, BN backward, ReLU mask -> dL/d(pre-activation), in place
            DCTR_TRY(bn_backward(E->dh[i], fc.out, E->h[i], fc.out, B, fc.out, E->bn_stats[i], E->pp(fc.bn_gamma), fc.keep, bn_seedp,
                                 fc.salt, E->bn_scratch, E->part(fc.bn_beta), E->part(fc.bn_gamma), E->dh[i], fc.out, st,
                                 E->bn_sync.world > 1 ? &E->bn_sync : nullptr));
        if (!wgrad_late && i >= late_layers) {
            // dh[i] is complete on st -- and so is dgrad_{i+1}, the last reader of W_{i+1}: ONE record serves the weight gradient of
            // this layer and (fused_opt) the optimizer step of the layer above, whose wgrad is already queued on sw
            if (i == nl - 1 && head_ev != nullptr && !E->bn && E->head_did_out_bwd) DCTR_HIP_CHECK(hipStreamWaitEvent(sw, *head_ev, 0));
            else DCTR_TRY(stop_fork(E, st, sw));        // (rides on the dgrad launch of the layer above when that one was armed)
            // A/B knob DCTR_OPT_SIDE=1: the optimizer step of the layer above on a stream of its own instead of in front of this
            // layer's weight gradient (measured at c2: 0.3052 vs 0.3062 ms/step -- nothing; every kernel here fills the chip, the
            // step is the sum of their solo times whatever the order)
            static const bool opt_side = getenv("DCTR_OPT_SIDE") != nullptr;
            if (fused_opt && i < nl - 1 && !opt_tail && !E->lean_step) {
                if (opt_side && E->s_opt != nullptr && sw != st) {
                    DCTR_TRY(fork(E, sw, E->s_opt));
                    DCTR_TRY(opt_dense_range(E, E->mlp[i + 1].w, E->mlp[i + 1].last, E->s_opt));
                    E->opt_pending = true;
                } else {
                    DCTR_TRY(opt_dense_range(E, E->mlp[i + 1].w, E->mlp[i + 1].last, sw));
                }
            }
            // (dctr_step_timer mode 2: the backward products carry their own dispatch events too -- layer ids nl + i dgrad, 2 nl + i wgrad)
            const bool tw = E->timer_step && E->timer_mode == 2 && E->timer_n + 2 <= E->timer_ev.size() && !E->bn && !(i == 0 && E->opnn_fused);
            if (tw) arm_timer_events(E->timer_ev[E->timer_n], E->timer_ev[E->timer_n + 1]);
            // A/B knob DCTR_WGRAD0_LOW_PRIO=1: the first layer's weight gradient -- the one that runs beside the table step the NEXT gather
            // waits for -- does not raise its waves' issue priority
            static const bool w0_low = [] { const char* v = getenv("DCTR_WGRAD0_LOW_PRIO"); return v != nullptr && v[0] == '1'; }();
            GemmOpt gow = go;
            gow.wgrad_low_prio = (i == 0 && w0_low) ? 1 : 0;
            DCTR_TRY(fc_bwd_weights_partials(x, ldx, E->dh[i], fc.out, E->part(fc.w), w.padded, E->part(fc.b), b.padded, B,
                                             (i == 0 && E->opnn_fused) ? D : fc.in, fc.out, fc.splits, sw, 1, &gow));
            if (tw) {
                if (timer_events_pending()) disarm_timer_events();
                else { E->timer_layer.push_back(2 * nl + i); E->timer_n += 2; }
            }
            if (i == 0 && E->opnn_fused)
                DCTR_TRY(opnn_outer_wgrad(E->e, E->e_ld, B, F, K, E->opnn_pairs, E->dh[0], fc.out, fc.out, E->part(fc.w) + (size_t)D * fc.out, sw));
        }
        // the NEXT cross-stream record on st -- the fork of the layer below, or the one that ends this function -- depends on this dgrad's
        // kernel and on nothing enqueued after it: it rides on the launch (not with batch_norm, whose backward kernels follow; not
        // when an interaction backward follows the last dgrad)
        const bool tail_plain = c.model == DCTR_MODEL_DEEPFM || c.model == DCTR_MODEL_FNN || c.model == DCTR_MODEL_WIDE || c.model == DCTR_MODEL_DEEP ||
                                c.model == DCTR_MODEL_WND;
        const bool td = E->timer_step && E->timer_mode == 2 && E->timer_n + 2 <= E->timer_ev.size() && !E->bn && !E->opnn_fused;
        if (td) arm_timer_events(E->timer_ev[E->timer_n], E->timer_ev[E->timer_n + 1]);       // (a launch has ONE stop event: a timed dgrad does not carry the fork's record)
        else if (!wgrad_late && late_layers == 0 && !E->bn && !E->opnn_fused && sw != st && (i > 0 || (fused_opt && tail_plain))) stop_arm(E);
        if (i > 0)
            DCTR_TRY(fc_bwd_data(E->dh[i], fc.out, E->pp(fc.w), E->dh[i - 1], E->mlp[i - 1].out, B, fc.in, fc.out,
                                 E->bn ? nullptr : E->h[i - 1], E->mlp[i - 1].out, E->bn ? 1.f : E->mlp[i - 1].keep, st, 1, &go));
        else if (E->opnn_fused) {
            DCTR_TRY(fc_bwd_data(E->dh[0], fc.out, E->pp(fc.w), E->dx_in, E->Din_ld, B, D, fc.out, nullptr, 0, 1.f, st, 1));
            DCTR_TRY(opnn_outer_dgrad(E, B, st));
        } else
            DCTR_TRY(fc_bwd_data(E->dh[0], fc.out, E->pp(fc.w), E->dx_in, E->Din_ld, B, fc.in, fc.out, nullptr, 0, 1.f, st, 1, &go));
        if (td) {
            if (timer_events_pending()) disarm_timer_events();
            else { E->timer_layer.push_back(nl + i); E->timer_n += 2; }
        }
    }
    if (wgrad_late) {
        // (experiment) the weight gradients AFTER the whole dgrad chain, beside the interaction backward, the scatter and the
        // table step, with ONE cross-stream record on st instead of one per layer; each layer's optimizer step follows its
        // gradient on a stream of its own.
        DCTR_TRY(fork(E, st, sw));
        for (int i = nl - 1; i >= 0; --i) {
            const Fc& fc = E->mlp[i];
            const float* x = i > 0 ? (E->bn ? E->hbn[i - 1] : E->h[i - 1]) : E->x_in;
            const int ldx = i > 0 ? E->mlp[i - 1].out : E->Din_ld;
            const Param& w = E->params[fc.w];
            const Param& b = E->params[fc.b];
            DCTR_TRY(fc_bwd_weights_partials(x, ldx, E->dh[i], fc.out, E->part(fc.w), w.padded, E->part(fc.b), b.padded, B, fc.in,
                                             fc.out, fc.splits, sw, 1));
            if (fused_opt) {
                DCTR_TRY(fork(E, sw, E->s_opt));
                DCTR_TRY(opt_dense_range(E, fc.w, fc.last, E->s_opt));
                E->opt_pending = true;
            }
        }
    }
    const uint64_t* seedp = &E->state->seed_t;
    // the interaction backward is the LAST kernel on st the fork below depends on: where its launch site can carry the record
    // (common.h DCTR_LAUNCH_RIDE), the record rides on it like the MLP models' on their last dgrad
    static const bool ride_off = [] { const char* v = getenv("DCTR_RIDE_LAST"); return v != nullptr && v[0] == '0'; }();      // A/B knob
    const bool ride_last = !ride_off && fused_opt && sw != st && !wgrad_late && late_layers == 0 && !E->bn &&
                           (c.model == DCTR_MODEL_IPNN || c.model == DCTR_MODEL_NFM || (c.model == DCTR_MODEL_DCN && dcn_lean(E)));
    if (ride_last) stop_arm(E);
    if (c.model == DCTR_MODEL_IPNN) DCTR_TRY(pnn_inner_bwd(E->e, E->e_ld, E->dx_in + D, E->Din_ld, B, F, K, E->dx_in, E->Din_ld, st));
    if (c.model == DCTR_MODEL_OPNN && !E->opnn_fused) DCTR_TRY(pnn_outer_bwd(E->e, E->e_ld, E->dx_in + D, E->Din_ld, B, F, K, E->dx_in, E->Din_ld, st));
    if (c.model == DCTR_MODEL_NFM) DCTR_TRY(dropout_inplace(E->dx_in, (int64_t)B * K, K, c.keep_prob[0], seedp, DCTR_DROPOUT_SITE_NFM_BI, st));
    if (c.model == DCTR_MODEL_MVM) {         // after the MLP's dgrad wrote dx_in: the product layer adds its share of dL/de
        const Param& pm = E->params[E->p_mvm_b];
        DCTR_TRY(mvm_bwd(E->e, E->e_ld, E->pp(E->p_mvm_b), E->dxmvm, B, F, K, E->dx_in, E->Din_ld, E->part(E->p_mvm_b), pm.padded, pm.n_part, st));
    }
    if (c.model == DCTR_MODEL_DCN) {
        const Param& cw = E->params[E->p_cross_w];
        // (fused_opt: the cross parameters' 2 L column-sum launches go to the side stream behind the fork below -- 37 us of small
        //  kernels that stood between the cross backward and the table step at c3)
        if (dcn_lean(E)) {
            // (the blocks' rows of parameter-gradient sums land in the cross scratch -- two [512, L D] blocks, build() sizes it for
            //  them --, folded into the slabs on the side stream, or here when there is none)
            float* rows_w = E->cross_scratch;
            float* rows_b = E->cross_scratch + (size_t)512 * c.cross_layers * D;
            DCTR_TRY(dcn_cross_bwd_fused(E->x_in, E->Din_ld, E->xlw, E->pp(E->p_cross_w), E->pp(E->p_cross_b), E->dxL, D, B, D, c.cross_layers,
                                         E->dx_in, E->Din_ld, rows_w, rows_b, st));
            if (!(fused_opt && sw != st))
                DCTR_TRY(dcn_cross_param_slabs(rows_w, rows_b, B, D, c.cross_layers, E->part(E->p_cross_w), E->part(E->p_cross_b), cw.n_part, cw.padded, st));
        } else
        DCTR_TRY(dcn_cross_bwd(E->xs, E->xlw, E->pp(E->p_cross_w), E->dxL, D, B, D, c.cross_layers, E->dx_in, E->Din_ld,
                               fused_opt && sw != st ? nullptr : E->part(E->p_cross_w), E->part(E->p_cross_b), cw.n_part, cw.padded, E->cross_scratch, st));
    }
    if (fused_opt) {
        // everything the dense side reads or writes on st is enqueued: the first layer's step (its dgrad is done) and, for the
        // caller, the cross-network / output-layer partial slabs
        DCTR_TRY(stop_fork(E, st, sw));
        if (c.model == DCTR_MODEL_DCN && sw != st && dcn_lean(E)) {
            const Param& cw = E->params[E->p_cross_w];
            DCTR_TRY(dcn_cross_param_slabs(E->cross_scratch, E->cross_scratch + (size_t)512 * c.cross_layers * D, B, D, c.cross_layers,
                                           E->part(E->p_cross_w), E->part(E->p_cross_b), cw.n_part, cw.padded, sw));
        } else if (c.model == DCTR_MODEL_DCN && sw != st) {
            const Param& cw = E->params[E->p_cross_w];
            DCTR_TRY(dcn_cross_param_grads(E->xs, B, D, c.cross_layers, E->part(E->p_cross_w), E->part(E->p_cross_b), cw.n_part, cw.padded, E->cross_scratch, sw));
        }
        for (int i = late_layers - 1; i >= 0; --i) {        // (DCTR_WGRAD_LATE_LAYERS: every dgrad has read its weights by now)
            const Fc& fc = E->mlp[i];
            const float* x = i > 0 ? E->h[i - 1] : E->x_in;
            const int ldx = i > 0 ? E->mlp[i - 1].out : E->Din_ld;
            if (i + 1 < nl && !E->lean_step) DCTR_TRY(opt_dense_range(E, E->mlp[i + 1].w, E->mlp[i + 1].last, sw));
            const GemmOpt gol = gemm_opt(E, fc);
            DCTR_TRY(fc_bwd_weights_partials(x, ldx, E->dh[i], fc.out, E->part(fc.w), E->params[fc.w].padded, E->part(fc.b), E->params[fc.b].padded, B,
                                             fc.in, fc.out, fc.splits, sw, 1, &gol));
        }
        if (!wgrad_late && !E->lean_step) DCTR_TRY(opt_dense_range(E, E->mlp[0].w, opt_tail ? E->mlp[nl - 1].last : E->mlp[0].last, sw));
    }
    return DCTR_OK;
}

// the step's tail as ONE launch (embed_scatter_apply): whenever the table step after the scatter visits only the batch's distinct
// rows -- dense-exact with the untouched rows stepped in the background, or the lazy touched_rows mode.  DCTR_FUSED_TAIL=0 keeps
// the two-launch tail (A/B knob)
bool split_table_on(const dctr_engine* E) {
    static const bool no_split = getenv("DCTR_NO_SPLIT_TABLE") != nullptr;      // A/B knob
    return E->cfg.table_mode == DCTR_TABLE_DENSE_EXACT && !no_split;
}
bool tail_fused(const dctr_engine* E) {
    static const bool off = [] { const char* v = getenv("DCTR_FUSED_TAIL"); return v != nullptr && v[0] == '0'; }();
    return !off && !E->wnd && E->cfg.shard_world == 1 && (split_table_on(E) || E->cfg.table_mode != DCTR_TABLE_DENSE_EXACT);
}

// time-blocked sweep (lag.h): is this handle's table allowed to lag in training steps?
bool lag_on(const dctr_engine* E) { return E->lag_period > 1 && !E->lag_suspended && split_table_on(E) && tail_fused(E); }
// the owner side of the row-sharded path (dctr_table_gather_packed / dctr_table_apply_packed) under the same scheme: its
// shard's rows lag and are replayed exactly like the unsharded table's
bool owner_lag(const dctr_engine* E) {
    static const bool off = [] { const char* v = getenv("DCTR_OWNER_LAG"); return v != nullptr && v[0] == '0'; }();     // A/B knob
    return !off && E->owner_lag_opt_in && E->lag_period > 1 && !E->lag_suspended && split_table_on(E);
}
LagView lag_view(const dctr_engine* E) {
    return LagView{E->row_ts, E->state, reinterpret_cast<float4*>(E->emb_s0), reinterpret_cast<float4*>(E->emb_s1), E->lin_s0, E->lin_s1, E->cfg.l2_reg,
                   E->tab_ld / 4, E->lin_ld};
}
// every row to step state->t + offset (0: the present, between steps; -1: inside a step whose state has already advanced);
// with_sums: sum theta^2 of all rows (at that step) into the step's loss scalars
int lag_flush_tables(dctr_engine* E, hipStream_t st, int offset, bool with_sums) {
    if (E->lag_period <= 1) return DCTR_OK;
    if (!E->lag_dirty && !with_sums) return DCTR_OK;
    DCTR_TRY(lag_flush(E->K, E->rows, E->emb, E->emb_s0, E->emb_s1, E->lin, E->lin_s0, E->lin_s1, E->row_ts, E->state, E->cfg.l2_reg, offset,
                       with_sums ? E->scalars + SUMSQ_SHARDS : nullptr, (with_sums && E->lin) ? E->scalars + 2 * SUMSQ_SHARDS : nullptr, st,
                       E->tab_ld, E->lin_ld));
    E->lag_dirty = false;
    return DCTR_OK;
}

// ---- table side of the backward: segment-sum the row gradients (ids already grouped), step the tables ------------
int scatter_and_step_tables(dctr_engine* E, int B, hipStream_t st, hipStream_t st_lin = nullptr, int pass = OPT_PASS_ALL) {
    const dctr_config& c = E->cfg;
    const int mode = gather_mode(E);
    const float* dE = mode == DCTR_GATHER_BI ? nullptr : E->dE;
    const float* coef = mode == DCTR_GATHER_FM ? E->dy : (mode == DCTR_GATHER_BI ? E->dx_in : nullptr);
    if (tail_fused(E) && (pass == OPT_PASS_TOUCHED || c.table_mode != DCTR_TABLE_DENSE_EXACT)) {
        const bool lag = lag_on(E) && pass == OPT_PASS_TOUCHED;
        // (lagging rows: sum theta^2 of the visited rows alone means nothing -- a loss-reporting step takes it from the flush)
        return embed_scatter_apply(E->group, c.optimizer, &E->state->hyper, E->h_state.hyper, E->emb, E->emb_s0, E->emb_s1, E->lin, E->lin_s0,
                                   E->lin_s1, c.l2_reg, lag ? nullptr : E->scalars + SUMSQ_SHARDS, lag ? nullptr : E->scalars + 2 * SUMSQ_SHARDS,
                                   dE, E->dE_ld, E->e, E->e_ld, E->S, coef, E->lin ? E->dy : nullptr, E->vals, B, E->F, E->K, mode, st, 1, nullptr,
                                   lag ? E->row_ts : nullptr, lag ? E->state : nullptr, E->tab_ld, E->lin_ld);
    }
    DCTR_TRY(embed_scatter_bwd(E->group, dE, E->dE_ld, E->e, E->e_ld, E->S, coef, E->lin ? E->dy : nullptr, E->vals, B, E->F,
                               E->K, mode, E->group->gemb, E->lin ? E->group->glin : nullptr, st));
    if (pass == OPT_PASS_TOUCHED) st_lin = nullptr;     // the touched-rows kernel steps the linear weights in the same launch
    if (st_lin != nullptr && st_lin != st) DCTR_TRY(fork(E, st, st_lin));        // the compact gradients are complete on st
    DCTR_TRY(opt_table(c.optimizer, &E->state->hyper, E->h_state.hyper, c.table_mode, E->rows, E->K, E->emb, E->emb_s0, E->emb_s1,
                       E->lin, E->lin_s0, E->lin_s1, E->group->slot, E->group->uniq, E->group->counters, E->group->max_entries,
                       E->group->gemb, E->group->glin, c.l2_reg, E->scalars + SUMSQ_SHARDS, E->scalars + 2 * SUMSQ_SHARDS, st, st_lin,
                       pass, E->tab_ld, E->lin_ld));
    if (st_lin != nullptr && st_lin != st) DCTR_TRY(fork(E, st_lin, st));
    return DCTR_OK;
}

// dense-exact tables: the rows the batch does NOT touch have gradient l2*theta, which depends on nothing this step computes --
// their optimizer step (90 % of the table pass) runs right after the grouping, under the MLP GEMMs, instead of at the end
int step_untouched_rows(dctr_engine* E, hipStream_t st) {
    const dctr_config& c = E->cfg;
    if (lag_on(E)) {       // one block of the table per step, its rows advanced through every step they missed (lag.h)
        E->lag_dirty = true;
        return lag_sweep(E->K, E->rows, E->emb, E->emb_s0, E->emb_s1, E->lin, E->lin_s0, E->lin_s1, E->group->slot, E->row_ts, E->state,
                         c.l2_reg, E->lag_period, st, E->tab_ld, E->lin_ld);
    }
    return opt_table(c.optimizer, &E->state->hyper, E->h_state.hyper, c.table_mode, E->rows, E->K, E->emb, E->emb_s0, E->emb_s1,
                     E->lin, E->lin_s0, E->lin_s1, E->group->slot, E->group->uniq, E->group->counters, E->group->max_entries,
                     E->group->gemb, E->group->glin, c.l2_reg, E->scalars + SUMSQ_SHARDS, E->scalars + 2 * SUMSQ_SHARDS, st, nullptr,
                     OPT_PASS_UNTOUCHED, E->tab_ld, E->lin_ld);
}

// The step as a small DAG over three streams (captured into one hipGraph):
//   st : state -> gather -> MLP fwd -> head -> dgrad chain -> interaction bwd -> [join grouping] scatter -> table optimizer
//   sg : grouping of the batch's ids (depends only on the inputs; hidden under the MLP)
//   sw : weight gradients (each waits for its layer's dY) -> dense optimizer (runs beside scatter + table optimizer)
// Canned estimators (wide_n_deep.py:113-151): DNN side through the same MLP kernels with `optimizer`, linear side
// (table + numeric weights + bias) with `lin_optimizer`; sparse gradients, so only the batch's distinct rows move.
int record_train_wnd(dctr_engine* E, int B, hipStream_t st) {
    const dctr_config& c = E->cfg;
    hipStream_t sg = E->s_group, sw = E->s_wgrad;
    DCTR_TRY(step_state_advance(E->state, E->scalars, 4 * SUMSQ_SHARDS, st));
    DCTR_TRY(fork(E, st, sg));
    DCTR_TRY(group_ids(E->group, E->ids, B, E->F, sg));        // beside the forward pass
    DCTR_TRY(forward(E, B, true, st));
    DCTR_TRY(head(E, B, B, true, st, nullptr, E->wnd_deep));
    if (E->wnd_deep) DCTR_TRY(backward_dense(E, B, st, sw, false));
    if (E->wnd_wide) {
        // d linear_dense[j] = sum_b dy[b] x[b,j];  d bias = sum_b dy[b] (aliases the output bias' slabs when there is a DNN side)
        if (E->p_lin_dense >= 0) {
            const Param& pd = E->params[E->p_lin_dense];
            DCTR_TRY(colsum_partials(E->dense, E->n_dense, E->dy, B, E->n_dense, pd.n_part, E->part(E->p_lin_dense), pd.padded, st));
        }
        if (!E->wnd_deep) {
            const Param& pb = E->params[E->p_bias];
            DCTR_TRY(colsum_partials(E->ones, 1, E->dy, B, 1, pb.n_part, E->part(E->p_bias), pb.padded, st));
        }
    }
    DCTR_TRY(fork(E, sw, st));
    // dense parameters, each side with its own optimizer
    for (int i = 0; i < (int)E->params.size(); ++i) {
        if (E->params[i].is_table) continue;
        DCTR_TRY(opt_dense_range(E, i, i, st, i == E->p_bias || i == E->p_lin_dense));
    }
    // tables: segment-sum of the row gradients, then sparse apply per side
    DCTR_TRY(fork(E, sg, st));
    DCTR_TRY(embed_scatter_bwd(E->group, E->wnd_deep ? E->dx_in : nullptr, E->Din_ld, nullptr, 0, nullptr, nullptr,
                               E->wnd_wide ? E->dy : nullptr, E->vals, B, E->F, E->K, DCTR_GATHER_RAW, E->group->gemb,
                               E->wnd_wide ? E->group->glin : nullptr, st));
    if (E->wnd_deep)
        DCTR_TRY(opt_table(c.optimizer, &E->state->hyper, E->h_state.hyper, DCTR_TABLE_TOUCHED_ROWS, E->rows, E->K, E->emb, E->emb_s0,
                           E->emb_s1, nullptr, nullptr, nullptr, E->group->slot, E->group->uniq, E->group->counters,
                           E->group->max_entries, E->group->gemb, nullptr, 0.f, E->scalars + SUMSQ_SHARDS, E->scalars + 2 * SUMSQ_SHARDS, st));
    if (E->wnd_wide)
        DCTR_TRY(opt_lin_touched(c.lin_optimizer, &E->state->hyper_lin, E->h_state.hyper_lin, E->lin, E->lin_s0, E->lin_s1,
                                 E->group->uniq, E->group->counters, E->group->max_entries, E->group->glin, st));
    return DCTR_OK;
}

int record_train(dctr_engine* E, int B, hipStream_t st) {
    // (an event armed for a launch that an error path never reached must not ride on this step's first launch: common.h arm_stop_event)
    if (stop_event_pending()) disarm_stop_event();
    if (timer_events_pending()) disarm_timer_events();
    E->armed_ev = nullptr;
    if (E->wnd) return record_train_wnd(E, B, st);
    hipStream_t sg = E->s_group, sw = E->s_wgrad;
    // per-step state (loss scalars, global_step, Adam lr_t, dropout seed) off the critical path: the gather does not need it
    // the per-step state kernel (5 us) runs on st: handing it to a side stream costs st a record AND a wait on a fresh
    // dependency -- two cross-queue hops of ~10 us each, measured 0.374 -> 0.351 ms/step.  DCTR_STATE_ON_SIDE=1 is the old placement
    static const bool state_on_main = getenv("DCTR_STATE_ON_SIDE") == nullptr;
    // ids grouped ahead (dctr_prefetch_ids, during the tail of the previous step)
    const bool pregrouped = E->pre_valid && E->pre_ids == E->ids && E->pre_B == B && E->group_alt != nullptr && !E->cfg.use_graph &&
                            E->pre_gen == E->slot_gen[E->pre_slot].load();     // (the slot still holds what was grouped)
    if (E->x_in_alt != nullptr && !E->cfg.use_graph) {       // (engine.h ev_dense: the step before may still be reading its x_in on the side stream)
        const bool e_is_x = E->e == E->x_in;
        std::swap(E->x_in, E->x_in_alt);
        if (e_is_x) E->e = E->x_in;
    }
    // the next step's state is prepared EARLY in this step, on the grouping stream in front of the record the table step waits for
    // (it only READS the live state): the step's last join can then be deferred past the next gather (below)
    static const bool no_state_ahead = getenv("DCTR_NO_STATE_AHEAD") != nullptr;       // A/B knob
    static const bool defer_off = [] { const char* v = getenv("DCTR_DEFER_JOIN"); return v != nullptr && v[0] == '0'; }();     // A/B knob
    // (not while the step is being CAPTURED -- dctr_time_kernel("train_step"): the deferred join would leave sw unjoined at EndCapture and
    //  dense_pending pointing at an event of the capture)
    const bool state_early_ok = !defer_off && !E->cfg.use_graph && !E->capturing && !no_state_ahead && sw != st && sg != st && E->cfg.shard_world == 1;
    bool state_early = false;
    if (E->state_ready) {
        // prepared under the tail of the previous step (below): the two states / scalar sets change roles
        std::swap(E->state, E->state_alt);
        std::swap(E->scalars, E->scalars_alt);
        E->state_ready = false;
        DCTR_TRY(forward_gather_train(E, B, st));
    } else if (state_on_main || lag_on(E)) {
        DCTR_TRY(step_state_advance(E->state, E->scalars, 4 * SUMSQ_SHARDS, st));
        DCTR_TRY(forward_gather_train(E, B, st));
    } else {
        DCTR_TRY(fork(E, st, sw));
        DCTR_TRY(step_state_advance(E->state, E->scalars, 4 * SUMSQ_SHARDS, sw));
        DCTR_TRY(forward_gather(E, B, st));
        DCTR_TRY(fork(E, sw, st));          // (before the fork below: sg's table pass needs this step's lr_t and zeroed scalars)
    }
    const bool split_table = split_table_on(E);
    static const bool bg_late = getenv("DCTR_BG_LATE") != nullptr;              // A/B knob: background table pass beside the backward
    // where the grouping stream starts (id grouping unless prefetched, then the background table pass): 0 = after the gather,
    // i = after MLP layer i - 1.  Round 3: behind a first layer worth waiting for (c2: 624 x 400) -- the background pass is ALU-bound
    // now (lagging rows replayed in registers) and beside layer 0 it cost that product 7 us of its 36 (in-step 0.36 -> 0.45 of the
    // MFMA peak by the hipEvent bracket; step 0.2668 -> 0.2646 ms, c3 0.312 -> 0.307, c4 inner 0.502 -> 0.494); NFM's K-wide
    // first layer is over before the pass could start, and the table step would wait for it (0.233 -> 0.247): there, 0.
    // With the ids grouped ahead (next-batch hint) only the background pass is left for this stream and nothing but the table step
    // waits for it: it starts behind the LAST forward layer (n = number of MLP layers) -- c2 0.2650 -> 0.2605, NFM 0.232 -> 0.221,
    // c3 / c4 inner / DeepMVM -1 %; without the hint that is too late for the grouping itself at c4's sizes (0.493 -> 0.512).
    static const int group_after_env = getenv("DCTR_GROUP_AFTER") ? atoi(getenv("DCTR_GROUP_AFTER")) : -1;   // A/B knob
    // the id grouping (and the background table pass behind it) on the grouping stream
    // ids grouped ahead (dctr_prefetch_ids, during the tail of the previous step): the two grouping states change roles
    E->pre_valid = false;
    if (pregrouped) std::swap(E->group, E->group_alt);
    const int group_after = group_after_env >= 0 ? group_after_env
                            : E->mlp.empty() ? 0
                            : pregrouped ? (int)E->mlp.size()
                            : ((int64_t)E->mlp[0].in * E->mlp[0].out >= (1 << 16) ? 1 : 0);
    hipEvent_t tables_ev = nullptr;
    bool have_tables_ev = false;
    bool sg_joined_by_tables_ev = false;
    hipEvent_t sg_done_ev = nullptr;
    bool have_sg_done = false;
    // A/B knob DCTR_SWEEP_AFTER_HEAD=1: with the ids grouped ahead the grouping stream has only the background sweep to start, and the
    // fork that starts it behind the last forward layer is a record on st -- a barrier packet between that GEMM and the head (~5 us in
    // the timeline).  Here the sweep starts behind the record the head needs anyway (head_ev below).
    static const bool sweep_after_head_env = getenv("DCTR_SWEEP_AFTER_HEAD") != nullptr;
    const bool late_sweep = sweep_after_head_env && pregrouped && split_table && !bg_late && E->cfg.model != DCTR_MODEL_AFM && !E->mlp.empty();
    const std::function<int()> start_grouping = [&]() -> int {
        if (late_sweep) return DCTR_OK;
        DCTR_TRY(stop_fork(E, st, sg));     // not before the gather (its atomics slow a concurrent gather 4x); rides on the layer's launch when armed
        // (a captured step is replayed from states this enqueue cannot see: it always carries the slot-reset kernel)
        if (E->cfg.use_graph) E->group->slots_clean = false;
        if (!pregrouped) DCTR_TRY(group_ids(E->group, E->ids, B, E->F, sg, !tail_fused(E)));
        if (split_table && !bg_late) DCTR_TRY(step_untouched_rows(E, sg));
        // what the scatter needs from this stream ends here: it waits for THIS record, not for the output layer's optimizer
        // launches that follow on sg (two latency-bound kernels, ~80 us in the step: they used to hold the scatter back ~10 us)
        if (!bg_late) {
            if (state_early_ok) {
                // (started BEFORE forward_rest's join -- a small first layer, ids not grouped ahead: state_alt / scalars_alt are the live
                //  state of the step before, which that step's deferred optimizer tail on sw may still read and add its l2 sums to)
                if (E->dense_pending) DCTR_HIP_CHECK(hipStreamWaitEvent(sg, E->ev_dense, 0));
                DCTR_TRY(step_state_next(E->state, E->state_alt, E->scalars_alt, 4 * SUMSQ_SHARDS, sg));
                E->state_ready = state_early = true;
            }
            DCTR_TRY(record_on(E, sg, &tables_ev));
            have_tables_ev = true;
        }
        return DCTR_OK;
    };
    const bool have_mlp = !E->mlp.empty() && E->cfg.model != DCTR_MODEL_AFM;
    if (!(group_after >= 1 && have_mlp)) DCTR_TRY(start_grouping());
    DCTR_TRY(forward_rest(E, B, true, st, (group_after >= 1 && have_mlp) ? &start_grouping : nullptr, group_after - 1));
    const bool fused_opt = E->cfg.model != DCTR_MODEL_AFM;
    // Small batches (c1, the reference's own B = 128 / 256: run.sh:11-22) run at the pace of the HOST's enqueue calls -- 34 per step at ~3 us
    // (profiles/r05_c1_hip_api_stats.txt), the GPU idle between them.  A LEAN step keeps the same kernels on the same three streams but
    // drops the placements that exist to overlap full-chip kernels: the output layer's optimizer launches on the grouping stream behind a
    // record of their own, and one optimizer launch per group of parameters -- every dense variable is stepped by ONE launch at the tail
    // of the weight-gradient stream (the arena is contiguous).  Six enqueue calls less per step.  A/B knob DCTR_LEAN_BATCH=<rows> (0 = off).
    static const int lean_rows = getenv("DCTR_LEAN_BATCH") ? atoi(getenv("DCTR_LEAN_BATCH")) : 512;
    const bool lean = fused_opt && B <= lean_rows && !E->cfg.use_graph && sw != st && sg != st && !E->bn && !E->opnn_fused && E->cfg.shard_world == 1 &&
                      getenv("DCTR_WGRAD_LATE") == nullptr && getenv("DCTR_WGRAD_LATE_LAYERS") == nullptr;
    E->lean_step = lean;
    if (fused_opt && !lean) stop_arm(E);    // (the record behind the head rides on the fused head kernel's launch when that path is taken)
    DCTR_TRY(head(E, B, B, true, st, nullptr, true));
    hipEvent_t head_ev = nullptr;
    bool have_head_ev = false;
    if (fused_opt && E->head_did_out_bwd && !lean) {
        // the output layer (and the global bias, whose gradient aliases the output bias' slabs) is final right after the fused
        // head kernel: step it now, beside the MLP backward, instead of at the end of the step beside the scatter
        // (on the grouping stream, idle since the MLP forward: on the wgrad stream these two latency-bound launches -- 128 slabs
        //  summed by one block each, ~35 us -- would delay the whole weight-gradient chain behind them)
        DCTR_TRY(stop_record(E, st, &head_ev));
        have_head_ev = true;
        DCTR_HIP_CHECK(hipStreamWaitEvent(sg, head_ev, 0));
        if (late_sweep) {
            DCTR_TRY(step_untouched_rows(E, sg));
            DCTR_TRY(record_on(E, sg, &tables_ev));
            have_tables_ev = true;
        }
        if (split_table && bg_late) DCTR_TRY(step_untouched_rows(E, sg));
        DCTR_TRY(opt_dense_range(E, E->p_out_w, E->p_out_b, sg));
        if (E->p_bias >= 0) DCTR_TRY(opt_dense_range(E, E->p_bias, E->p_bias, sg));
        // The end of the step used to join BOTH side streams on st: two waits in front of the next gather.  This stream's work of
        // the step ends here, long before the step does: the weight-gradient stream waits for it (in front of its last kernel) and st
        // joins that stream only -- c2 0.2648 -> 0.2624 ms/step, c4 NFM 0.2188 -> 0.2154, PNN-inner 0.500 -> 0.495.  (=1: the table
        // step waits for it instead -- as good at c2 / c3, but NFM's short backward reaches the table step before this stream is done:
        // 0.221 -> 0.252.  =0: both joins, as before.)
        static const int one_join = getenv("DCTR_ONE_SG_JOIN") ? atoi(getenv("DCTR_ONE_SG_JOIN")) : 2;
        if (one_join == 1 && have_tables_ev && sg != st) { DCTR_TRY(record_on(E, sg, &tables_ev)); sg_joined_by_tables_ev = true; }
        if (one_join == 2 && have_tables_ev && sg != st && sw != st) { DCTR_TRY(record_on(E, sg, &sg_done_ev)); have_sg_done = true; }
    }
    if (late_sweep && !have_tables_ev) {      // (no fused head on this model: the ordinary fork)
        DCTR_TRY(fork(E, st, sg));
        DCTR_TRY(step_untouched_rows(E, sg));
        DCTR_TRY(record_on(E, sg, &tables_ev));
        have_tables_ev = true;
    }
    if (E->armed_ev != nullptr) { disarm_stop_event(); E->armed_ev = nullptr; }      // (armed for the head, not used)
    const bool out_done = fused_opt && E->head_did_out_bwd && !lean;
    if (lean && have_tables_ev) sg_joined_by_tables_ev = true;      // (nothing follows that record on the grouping stream in a lean step: st's wait for it joins the stream)
    // A/B knob DCTR_WGRAD_SERIAL=1: the weight gradients (and the per-layer optimizer steps) on the main stream, right behind their
    // layer's dgrad, instead of beside it on sw
    static const bool wgrad_serial = getenv("DCTR_WGRAD_SERIAL") != nullptr;
    static const bool lean_serial = [] { const char* v = getenv("DCTR_LEAN_SERIAL"); return v != nullptr && v[0] == '1'; }();      // A/B knob (measured slower: 0.115 vs 0.101)
    const bool serial = wgrad_serial || (lean && lean_serial);
    DCTR_TRY(backward_dense(E, B, st, serial ? st : sw, fused_opt, have_head_ev ? &head_ev : nullptr));
    // (fused_opt: backward_dense ends with that fork -- unless its weight gradients ran on st itself: then the optimizer launches below, on
    //  sw, are ordered behind them here)
    if (!fused_opt || (serial && sw != st)) DCTR_TRY(fork(E, st, sw));
    // where a prefetched grouping of the next batch may start (beside scatter + table step): the record of that last st -> sw
    // fork serves -- one more record here is one more barrier packet (~5 us) in front of the scatter
    if (!E->cfg.use_graph && E->last_fork_ev != nullptr) { E->ev_tail = E->last_fork_ev; E->have_tail = true; }
    if (fused_opt) {
        // what is left of the dense arena: cross_w / cross_b (DCN), and the output layer + bias if they were not stepped above --
        // neighbours in the arena in one launch (c3: cross_w, cross_b were two 6-us launches at the very end of the step)
        int run_first = -1, run_last = -1;
        auto flush_run = [&]() -> int {
            if (run_first >= 0) DCTR_TRY(opt_dense_range(E, run_first, run_last, sw));
            run_first = run_last = -1;
            return DCTR_OK;
        };
        for (int i = 0; i < (int)E->params.size(); ++i) {
            const Param& p = E->params[i];
            if (p.is_table) continue;
            bool is_mlp = false;
            for (auto& fc : E->mlp) is_mlp = is_mlp || (i >= fc.w && i <= fc.last);
            const bool skip = (is_mlp && !lean) || (out_done && (i == E->p_out_w || i == E->p_out_b || i == E->p_bias));
            if (skip) { DCTR_TRY(flush_run()); continue; }
            if (run_first >= 0 && E->params[run_last].arena_off + E->params[run_last].padded != p.arena_off) DCTR_TRY(flush_run());
            if (run_first < 0) run_first = i;
            run_last = i;
        }
        DCTR_TRY(flush_run());
    } else {
        DCTR_TRY(opt_dense_arena(E->cfg.optimizer, &E->state->hyper, E->h_state.hyper, E->theta, E->as0, E->as1, E->parts, E->meta,
                                 E->n_blocks, nullptr, 1, E->scalars + 3 * SUMSQ_SHARDS, sw));
        if (E->gemm_mode == 1) DCTR_TRY(wplanes_written(E, 0, (int)E->params.size() - 1, sw));
    }
    if (have_tables_ev) DCTR_HIP_CHECK(hipStreamWaitEvent(st, tables_ev, 0));
    else DCTR_TRY(fork(E, sg, st));
    if (split_table) DCTR_TRY(scatter_and_step_tables(E, B, st, nullptr, OPT_PASS_TOUCHED));
    else DCTR_TRY(scatter_and_step_tables(E, B, st, sg));  // the grouping stream is idle by now: linear table beside the embedding table
    // A/B knob DCTR_PREGROUP_WAIT=end: the next batch's grouping (dctr_prefetch_ids) starts BEHIND the table step -- beside the next gather and
    // first forward product -- instead of beside it: one more record on st (in front of the next gather)
    static const bool pregroup_end = [] { const char* v = getenv("DCTR_PREGROUP_WAIT"); return v != nullptr && strcmp(v, "end") == 0; }();
    if (pregroup_end && !E->cfg.use_graph) {
        hipEvent_t ev_end = nullptr;
        DCTR_TRY(record_on(E, st, &ev_end));
        E->ev_tail = ev_end; E->have_tail = true;
    }
    if (!E->cfg.use_graph && !no_state_ahead && sw != st) {
        // the next step's state (global_step + 1, Adam's lr_t, dropout seed, zeroed loss scalars) into the second StepState, on
        // the weight-gradient stream beside scatter / table step: it only READS the live state, and the join below orders it
        // (state_early: already enqueued on the grouping stream, in front of the record the table step waited for)
        if (have_sg_done) { DCTR_HIP_CHECK(hipStreamWaitEvent(sw, sg_done_ev, 0)); sg_joined_by_tables_ev = true; }
        if (!state_early) {
            DCTR_TRY(step_state_next(E->state, E->state_alt, E->scalars_alt, 4 * SUMSQ_SHARDS, sw));
            E->state_ready = true;
        }
    }
    // The step's last join.  What the weight-gradient stream still holds -- the first layer's weight gradient, its optimizer step, the
    // re-split of its weight (gemm_mode 1) and the joined grouping stream's output-layer steps -- is needed by the next FORWARD
    // PRODUCTS, not by the next gather (tables: this stream; step state: prepared early): with the state in place the join is
    // left to the next reader of a dense variable (join_deferred, behind the next step's gather).  A step that reports its loss
    // reads the l2 sums those optimizer launches add, so it joins here.  A/B knob DCTR_DEFER_JOIN=0.
    const bool defer = state_early && E->x_in_alt != nullptr && !E->want_loss && !E->opt_pending && fused_opt &&
                       (!have_tables_ev || sg_joined_by_tables_ev) && E->bn_sync.world <= 1;
    if (defer) {
        if (E->ev_dense == nullptr) DCTR_HIP_CHECK(hipEventCreateWithFlags(&E->ev_dense, hipEventDisableTiming | hipEventDisableSystemFence));
        DCTR_HIP_CHECK(hipEventRecord(E->ev_dense, sw));
        E->dense_pending = true;
    } else {
        DCTR_TRY(fork(E, sw, st));
    }
    if (have_tables_ev && !sg_joined_by_tables_ev) DCTR_TRY(fork(E, sg, st));      // (the output layer's step on sg: long finished, joined for the next forward)
    if (E->opt_pending) { DCTR_TRY(fork(E, E->s_opt, st)); E->opt_pending = false; }
    return DCTR_OK;
}

int record_predict(dctr_engine* E, int B, hipStream_t st) {
    DCTR_TRY(forward(E, B, false, st));
    DCTR_TRY(head(E, B, B, false, st));
    return DCTR_OK;
}

int run_graph(dctr_engine* E, std::map<int, hipGraphExec_t>& cache, int B, bool train, hipStream_t st) {
    if (!E->cfg.use_graph) return train ? record_train(E, B, st) : record_predict(E, B, st);
    const int key = B * DCTR_INPUT_SLOTS + E->cur_slot;        // input pointers are baked into the graph
    auto it = cache.find(key);
    if (it == cache.end()) {
        hipGraph_t graph = nullptr;
        hipStream_t cs = nullptr;
        DCTR_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        DCTR_HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int rc = train ? record_train(E, B, cs) : record_predict(E, B, cs);
        hipError_t e = hipStreamEndCapture(cs, &graph);
        hipStreamDestroy(cs);
        if (rc != DCTR_OK) { if (graph) hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return DCTR_ERR_HIP; }
        hipGraphExec_t exec = nullptr;
        DCTR_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        hipGraphDestroy(graph);
        it = cache.emplace(key, exec).first;
    }
    DCTR_HIP_CHECK(hipGraphLaunch(it->second, st));
    return DCTR_OK;
}

int stage_inputs(dctr_engine* E, const int32_t* ids, const float* vals, const float* labels, int B, hipStream_t st) {
    DCTR_REQUIRE(B > 0 && B <= E->MB, "batch %d outside (0, max_batch=%d]", B, E->MB);
    const size_t n = (size_t)B * E->F;
    // the caller's buffers ARE one of the input slots: select it, nothing to copy
    int slot = 0;
    for (int k = 0; k < DCTR_INPUT_SLOTS; ++k)
        if (ids == E->slot_ids[k] && vals == E->slot_vals[k] && (labels == nullptr || labels == E->slot_labels[k])) slot = k;
    E->cur_slot = slot;
    E->ids = E->slot_ids[slot]; E->vals = E->slot_vals[slot]; E->labels = E->slot_labels[slot];
    if (ids != E->ids) {
        E->slot_gen[slot]++;                // the staging copy rewrites the slot: a grouping prefetched from it is stale
        DCTR_HIP_CHECK(hipMemcpyAsync(E->ids, ids, n * 4, hipMemcpyDeviceToDevice, st));
    }
    if (vals != E->vals) DCTR_HIP_CHECK(hipMemcpyAsync(E->vals, vals, n * 4, hipMemcpyDeviceToDevice, st));
    if (labels != nullptr && labels != E->labels)
        DCTR_HIP_CHECK(hipMemcpyAsync(E->labels, labels, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    return DCTR_OK;
}

// [0] xent sum, [1] sum emb^2, [2] sum linear^2, [3] sum of l2-regularised dense params^2 (shards summed on the host); syncs
int read_scalars(dctr_engine* E, float out[4], hipStream_t st) {
    float raw[4 * SUMSQ_SHARDS];
    DCTR_HIP_CHECK(hipMemcpyAsync(raw, E->scalars, sizeof(raw), hipMemcpyDeviceToHost, st));
    DCTR_HIP_CHECK(hipStreamSynchronize(st));
    for (int k = 0; k < 4; ++k) {
        double s = 0.0;
        for (int j = 0; j < SUMSQ_SHARDS; ++j) s += raw[k * SUMSQ_SHARDS + j];
        out[k] = (float)s;
    }
    return DCTR_OK;
}

Param* find_param(dctr_engine* E, const char* name) {
    auto it = E->index.find(name ? name : "");
    if (it == E->index.end()) { set_error("no parameter named '%s'", name ? name : "(null)"); return nullptr; }
    return &E->params[it->second];
}

}  // namespace


// ---- parameters of a padded-K engine in their logical shape (Param::kseg) -----------------------------------------------------
struct KSegs { Param::KSeg s[4]; int n; };
__global__ __launch_bounds__(256) void kpad_copy_kernel(float* __restrict__ phys, float* __restrict__ logi, KSegs S, int64_t row_elems,
                                                       int64_t log_n, int to_phys) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < log_n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / row_elems, c = i - r * row_elems;
        int64_t pr = -1;
        for (int k = 0; k < S.n; ++k) {
            const Param::KSeg& g = S.s[k];
            const int64_t q = r - g.log_off;
            if (q >= 0 && q < g.n_blocks * g.blk) { pr = g.phys_off + (q / g.blk) * g.stride + q % g.blk; break; }
        }
        if (pr < 0) continue;
        if (to_phys) phys[pr * row_elems + c] = logi[i];
        else logi[i] = phys[pr * row_elems + c];
    }
}

// which parameters' layouts depend on K, per model: runs of K logical rows sit at stride Kp
static int build_k_layouts(dctr_engine* E) {
    const int K = E->K_log, Kp = E->K, F = E->F;
    const dctr_config& c = E->cfg;
    for (auto& p : E->params) {
        p.log_n = p.n;
        for (int i = 0; i < 4; ++i) p.log_dims[i] = p.dims[i];
        if (K == Kp) continue;
        auto blocks = [&](int64_t n_blocks, int64_t rest, int64_t row_elems) {
            p.k_row_elems = row_elems;
            p.kseg.push_back({0, 0, n_blocks, K, Kp});
            if (rest > 0) p.kseg.push_back({n_blocks * K, n_blocks * Kp, 1, rest, rest});
            p.log_n = (n_blocks * K + rest) * row_elems;
        };
        const std::string& nm = p.name;
        const bool first_layer = nm == "mlp0/weights" || nm == "ctr_mlp0/weights" || nm == "cvr_mlp0/weights";
        if (nm == "emb") { blocks(p.dims[0], 0, 1); p.log_dims[1] = K; }
        else if (nm == "cross_w" || nm == "cross_b") { blocks(p.dims[0] * F, 0, 1); p.log_dims[1] = (int64_t)F * K; }
        else if (nm == "mvm_b") { blocks(F, 0, 1); p.log_dims[1] = K; }
        else if (first_layer) {
            const int64_t H = p.dims[1];
            if (c.model == DCTR_MODEL_NFM) { blocks(1, 0, H); p.log_dims[0] = K; }
            else {
                const int64_t rest = p.dims[0] - (int64_t)F * Kp;          // IPNN's P inner products behind the flat embeddings
                blocks(F, rest, H); p.log_dims[0] = (int64_t)F * K + rest;
            }
        }
        else if (nm == "out_layer/weights" && c.model == DCTR_MODEL_DCN) { const int64_t rest = p.dims[0] - (int64_t)F * Kp; blocks(F, rest, 1); p.log_dims[0] = (int64_t)F * K + rest; }
        else if (nm == "deep_out/weights" && c.model == DCTR_MODEL_MVM) { const int64_t rest = p.dims[0] - Kp; blocks(1, rest, 1); p.log_dims[0] = K + rest; }
        else if (nm == "deep_out/weights" && c.model == DCTR_MODEL_AFM) { blocks(1, 0, 1); p.log_dims[0] = K; }
        else if (nm == "att_mlp0/weights") { blocks(1, 0, p.dims[1]); p.log_dims[0] = K; }
        else if (nm == "att_fc0/weights") { blocks(3, 0, p.dims[1]); p.log_dims[0] = 3 * (int64_t)K; }
    }
    return DCTR_OK;
}

extern "C" {

int dctr_create(const dctr_config* cfg, dctr_handle* h) {
    DCTR_REQUIRE(cfg != nullptr && h != nullptr, "null argument");
    dctr_engine* E = new dctr_engine();
    (void)hipGetDevice(&E->device);          // the device current at creation: the input thread's copies (dctr_input_slot_fill) select it
    E->cfg = *cfg;
    if (E->cfg.shard_world <= 0) { E->cfg.shard_world = 1; E->cfg.shard_rank = 0; }
    // the reference takes any --embedding_size (DeepFM.py:43); the kernels take K/4 a power of two: other sizes run on the next
    // such size with the extra columns held at zero (Param::kseg), at that size's cost
    E->K_log = cfg->embedding_size;
    if (cfg->embedding_size < 1 || cfg->embedding_size > 256) {
        set_error("embedding_size %d outside [1, 256]", cfg->embedding_size);
        delete E;
        return DCTR_ERR_INVALID_ARG;
    }
    {
        int kp = 4;
        while (kp < cfg->embedding_size) kp *= 2;
        if (kp != cfg->embedding_size) {
            const bool ok = cfg->model != DCTR_MODEL_OPNN && !(cfg->model >= DCTR_MODEL_WIDE && cfg->model <= DCTR_MODEL_WND) && E->cfg.shard_world == 1;
            if (!ok) {
                set_error("embedding_size %d: Outer-PNN, the canned wide_n_deep models and row-sharded tables take K/4 a power of two (4..256)", cfg->embedding_size);
                delete E;
                return DCTR_ERR_UNSUPPORTED;
            }
            E->cfg.embedding_size = kp;
        }
    }
    // dctr_config.gemm_mode: 0 = the library's default (DCTR_GEMM_MODE, else split), 1 = split, 2 = exact.  Internally 1 = split, 0 = exact.
    E->gemm_mode = E->cfg.gemm_mode == 2 ? 0 : 1;
    if (E->cfg.gemm_mode == 0) {
        if (const char* gm = getenv("DCTR_GEMM_MODE")) {   // (runs a whole test suite in the other mode: handles that do not name one)
            if (!strcmp(gm, "split") || !strcmp(gm, "1")) E->gemm_mode = 1;
            else if (!strcmp(gm, "exact") || !strcmp(gm, "2")) E->gemm_mode = 0;
        }
    }
    int rc = build(E);
    if (rc == DCTR_OK) rc = build_k_layouts(E);
    if (rc == DCTR_OK) rc = wplanes_alloc(E);
    if (rc != DCTR_OK) { dctr_destroy(E); return rc; }
    *h = E;
    return DCTR_OK;
}

int dctr_destroy(dctr_handle E) {
    if (!E) return DCTR_OK;
    for (auto& kv : E->train_graphs) hipGraphExecDestroy(kv.second);
    for (auto& kv : E->predict_graphs) hipGraphExecDestroy(kv.second);
    if (E->table_rec != nullptr) {          // row records: the six table pointers are views into this one buffer
        hipFree(E->table_rec);
        E->emb = E->emb_s0 = E->emb_s1 = E->lin = E->lin_s0 = E->lin_s1 = nullptr;
    }
    float* fl[] = {E->emb, E->emb_s0, E->emb_s1, E->lin, E->lin_s0, E->lin_s1, E->theta, E->as0, E->as1, E->gflat, E->parts,
                   E->scalars, E->x_in, E->dx_in, E->e_buf, E->S, E->yw, E->yv, E->yd, E->y, E->prob, E->dy,
                   E->xs, E->xlw, E->dxL, E->cross_scratch};
    for (float* p : fl) if (p) hipFree(p);
    for (float* p : E->h) hipFree(p);
    for (float* p : E->dh) hipFree(p);
    for (float* p : E->h2) hipFree(p);
    for (float* p : E->dh2) hipFree(p);
    { float* f2[] = {E->dx_in2, E->dy2, E->y2, E->prob2, E->prob3}; for (float* p : f2) if (p) hipFree(p); }
    for (std::vector<Fc>* tower : {&E->mlp, &E->mlp2})
        for (Fc& fc : *tower) { if (fc.wp_fwd) hipFree(fc.wp_fwd); if (fc.wp_dgr) hipFree(fc.wp_dgr); }
    if (E->opnn_pairs) hipFree(E->opnn_pairs);
    if (E->opnn_ws) hipFree(E->opnn_ws);
    if (E->opnn_dop) hipFree(E->opnn_dop);
    if (E->entry_off) hipFree(E->entry_off);
    if (E->entry_goff) hipFree(E->entry_goff);
    if (E->pair_ad) hipFree(E->pair_ad);
    { float* f3[] = {E->x_att, E->att_sc, E->att_w}; for (float* p : f3) if (p) hipFree(p); }
    if (E->s_copy) { hipStreamSynchronize(E->s_copy); hipStreamDestroy(E->s_copy); }
    if (E->s_main) { hipStreamSynchronize(E->s_main); hipStreamDestroy(E->s_main); }
    for (int k = 0; k < DCTR_INPUT_SLOTS; ++k) { if (E->slot_filled[k]) hipEventDestroy(E->slot_filled[k]); if (E->slot_released[k]) hipEventDestroy(E->slot_released[k]); }
    for (int k = 0; k < DCTR_INPUT_SLOTS; ++k) if (E->slot_ids[k]) hipFree(E->slot_ids[k]);       // (vals and labels live in the same block)
    if (E->status) hipFree(E->status);
    if (E->state) hipFree(E->state);
    if (E->state_alt) hipFree(E->state_alt);
    if (E->scalars_alt) hipFree(E->scalars_alt);
    if (E->meta) hipFree(E->meta);
    if (E->meta_flat) hipFree(E->meta_flat);
    if (E->auc_counts) hipFree(E->auc_counts);
    if (E->eval_scalars) hipFree(E->eval_scalars);
    if (E->ones) hipFree(E->ones);
    if (E->group_alt) group_destroy(E->group_alt);
    if (E->row_ts) hipFree(E->row_ts);
    if (E->xmvm) hipFree(E->xmvm);
    if (E->dxmvm) hipFree(E->dxmvm);
    for (auto& ev : E->timer_ev) if (ev) hipEventDestroy(ev);
    for (float* p : E->hbn) hipFree(p);
    for (float* p : E->bn_stats) hipFree(p);
    for (float* p : E->hbn2) hipFree(p);
    for (float* p : E->bn_stats2) hipFree(p);
    if (E->bn_scratch) hipFree(E->bn_scratch);
    group_destroy(E->group);
    afm_free(E);
    for (auto& ev : E->events) if (ev) hipEventDestroy(ev);
    if (E->ev_dense) hipEventDestroy(E->ev_dense);
    if (E->x_in_alt) hipFree(E->x_in_alt);
    if (E->s_group) hipStreamDestroy(E->s_group);
    if (E->s_wgrad) hipStreamDestroy(E->s_wgrad);
    if (E->s_opt) hipStreamDestroy(E->s_opt);
    delete E;
    return DCTR_OK;
}

int dctr_param_count(dctr_handle E, int* n) {
    DCTR_REQUIRE(E && n, "null argument");
    *n = (int)E->params.size();
    return DCTR_OK;
}

int dctr_param_info(dctr_handle E, int index, const char** name, int* rank, int64_t dims[4]) {
    DCTR_REQUIRE(E && index >= 0 && index < (int)E->params.size(), "bad parameter index %d", index);
    const Param& p = E->params[index];
    if (name) *name = p.name.c_str();
    if (rank) *rank = p.rank;
    if (dims) for (int i = 0; i < 4; ++i) dims[i] = p.log_dims[i];
    return DCTR_OK;
}

// rows of `width` floats, `ld` apart in `strided`, back to back in `dense` (to_strided: dense -> strided)
__global__ void strided_copy_kernel(float* __restrict__ strided, float* __restrict__ dense, int64_t n, int width, int ld, int to_strided) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float* s = strided + (i / width) * ld + (i % width);
        if (to_strided) *s = dense[i]; else dense[i] = *s;
    }
}

static int copy_param_impl(dctr_handle E, const char* name, int which, void* host, size_t nbytes, bool to_device) {
    DCTR_REQUIRE(E && host, "null argument");
    Param* p = find_param(E, name);
    if (!p) return DCTR_ERR_NOT_FOUND;
    DCTR_REQUIRE(nbytes == (size_t)p->log_n * sizeof(float), "parameter '%s' holds %lld floats, caller passed %zu bytes", name,
                 (long long)p->log_n, nbytes);
    float* d = which < 0 ? p->ptr : (which == 0 ? p->s0 : p->s1);
    if (to_device && which < 0 && !p->is_table) wplanes_invalidate(E);       // (gemm_mode 1: the next product re-splits the layer's weight)
    if (which == 2) {           // the gradient of a dense variable: its partial slabs of the last backward pass, summed
        DCTR_REQUIRE(!p->is_table && !to_device, "dctr_param_grad_get: dense variables only (a table's gradient lives in the compact rows of the grouping)");
        DCTR_HIP_CHECK(hipDeviceSynchronize());
        DCTR_TRY(opt_dense_arena(E->cfg.optimizer, &E->state->hyper, E->h_state.hyper, E->theta, E->as0, E->as1, E->parts, E->meta, E->n_blocks,
                                 E->gflat, 0, nullptr, nullptr));
        d = E->gflat + p->arena_off;
    }
    DCTR_REQUIRE(d != nullptr, "parameter '%s' has no such slot", name);
    DCTR_HIP_CHECK(hipDeviceSynchronize());
    if (p->is_table && E->lag_dirty) {           // lagging rows (lag.h): the table as of global_step is what is read -- and what a write replaces
        DCTR_TRY(lag_flush_tables(E, nullptr, 0, false));
        DCTR_HIP_CHECK(hipDeviceSynchronize());
    }
    if (!p->kseg.empty()) {       // padded K: through a staging buffer in the logical layout
        float* stage = nullptr;
        DCTR_HIP_CHECK(hipMalloc(&stage, nbytes));
        KSegs S{};
        S.n = (int)p->kseg.size();
        for (int k = 0; k < S.n; ++k) S.s[k] = p->kseg[k];
        const int grid = (int)std::min<int64_t>(ceil_div(p->log_n, 256), 4096);
        hipError_t e = hipSuccess;
        if (to_device) {
            e = hipMemcpy(stage, host, nbytes, hipMemcpyHostToDevice);
            if (e == hipSuccess) { kpad_copy_kernel<<<grid, 256>>>(d, stage, S, p->k_row_elems, p->log_n, 1); e = hipDeviceSynchronize(); }
        } else {
            kpad_copy_kernel<<<grid, 256>>>(d, stage, S, p->k_row_elems, p->log_n, 0);
            e = hipMemcpy(host, stage, nbytes, hipMemcpyDeviceToHost);
        }
        hipFree(stage);
        if (e != hipSuccess) { set_error("parameter copy failed: %s", hipGetErrorString(e)); return DCTR_ERR_HIP; }
        return DCTR_OK;
    }
    if (p->is_table && E->table_rec != nullptr) {       // row records (engine.h): the variable is a strided view -- through a dense staging buffer
        const int width = p->name == "emb" ? E->K : 1;
        const int64_t n = p->log_n;
        float* stage = nullptr;
        DCTR_HIP_CHECK(hipMalloc(&stage, nbytes));
        const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 8192);
        hipError_t e = hipSuccess;
        if (to_device) {
            e = hipMemcpy(stage, host, nbytes, hipMemcpyHostToDevice);
            if (e == hipSuccess) { strided_copy_kernel<<<grid, 256>>>(d, stage, n, width, E->tab_ld, 1); e = hipDeviceSynchronize(); }
        } else {
            strided_copy_kernel<<<grid, 256>>>(d, stage, n, width, E->tab_ld, 0);
            e = hipMemcpy(host, stage, nbytes, hipMemcpyDeviceToHost);
        }
        hipFree(stage);
        if (e != hipSuccess) { set_error("parameter copy failed: %s", hipGetErrorString(e)); return DCTR_ERR_HIP; }
        return DCTR_OK;
    }
    if (to_device) DCTR_HIP_CHECK(hipMemcpy(d, host, nbytes, hipMemcpyHostToDevice));
    else DCTR_HIP_CHECK(hipMemcpy(host, d, nbytes, hipMemcpyDeviceToHost));
    return DCTR_OK;
}

static int copy_param(dctr_handle E, const char* name, int which, void* host, size_t nbytes, bool to_device) {
    DCTR_TRY(copy_param_impl(E, name, which, host, nbytes, to_device));
    // gemm_mode 1 with captured steps: a replayed graph runs no host logic, so the "re-split before the next product" of the eager path
    // (wplanes_ensure) would never happen -- the planes are refreshed here, once per write
    if (to_device && which < 0 && E->gemm_mode == 1 && E->cfg.use_graph) {
        for (std::vector<Fc>* tower : {&E->mlp, &E->mlp2})
            for (Fc& fc : *tower) DCTR_TRY(wplanes_ensure(E, fc, nullptr));
        DCTR_HIP_CHECK(hipDeviceSynchronize());
    }
    return DCTR_OK;
}
int dctr_param_set(dctr_handle h, const char* name, const float* h_src, size_t nbytes) {
    return copy_param(h, name, -1, const_cast<float*>(h_src), nbytes, true);
}
int dctr_param_get(dctr_handle h, const char* name, float* h_dst, size_t nbytes) { return copy_param(h, name, -1, h_dst, nbytes, false); }
int dctr_param_grad_get(dctr_handle h, const char* name, float* h_dst, size_t nbytes) { return copy_param(h, name, 2, h_dst, nbytes, false); }
int dctr_slot_get(dctr_handle h, const char* name, int which, float* h_dst, size_t nbytes) {
    DCTR_REQUIRE(which == 0 || which == 1, "slot index must be 0 or 1");
    return copy_param(h, name, which, h_dst, nbytes, false);
}
int dctr_slot_set(dctr_handle h, const char* name, int which, const float* h_src, size_t nbytes) {
    DCTR_REQUIRE(which == 0 || which == 1, "slot index must be 0 or 1");
    return copy_param(h, name, which, const_cast<float*>(h_src), nbytes, true);
}
static int param_device_view(dctr_handle E, const char* name, float** d_ptr, int64_t* row_stride);
int dctr_param_device_ptr(dctr_handle E, const char* name, float** d_ptr) {
    DCTR_REQUIRE(E && d_ptr, "null argument");
    Param* p0 = find_param(E, name);
    if (p0 != nullptr && p0->is_table && E->table_rec != nullptr) {
        set_error("parameter '%s' lives in row records (rows %d floats apart): use dctr_param_device_view, or dctr_param_get / _set", name, E->tab_ld);
        return DCTR_ERR_UNSUPPORTED;
    }
    return param_device_view(E, name, d_ptr, nullptr);
}
int dctr_param_device_view(dctr_handle E, const char* name, float** d_ptr, int64_t* row_stride) {
    DCTR_REQUIRE(row_stride != nullptr, "null argument");
    return param_device_view(E, name, d_ptr, row_stride);
}
static int param_device_view(dctr_handle E, const char* name, float** d_ptr, int64_t* row_stride) {
    DCTR_REQUIRE(E && d_ptr, "null argument");
    Param* p = find_param(E, name);
    if (!p) return DCTR_ERR_NOT_FOUND;
    if (!p->kseg.empty()) { set_error("parameter '%s' lives in a padded layout (embedding_size %d runs as %d): use dctr_param_get / _set", name, E->K_log, E->K); return DCTR_ERR_UNSUPPORTED; }
    if (p->is_table && E->lag_dirty) {
        DCTR_HIP_CHECK(hipDeviceSynchronize());
        DCTR_TRY(lag_flush_tables(E, nullptr, 0, false));
        DCTR_HIP_CHECK(hipDeviceSynchronize());
    }
    if (!p->is_table) {
        // (the step's deferred tail -- last weight gradient, the MLP's optimizer launch -- may still be running on the side stream: the
        //  caller reads or writes through the pointer on a stream of its own)
        if (E->dense_pending) { DCTR_HIP_CHECK(hipEventSynchronize(E->ev_dense)); E->dense_pending = false; }
        wplanes_invalidate(E);                      // (gemm_mode 1: the caller may write through the pointer)
    }
    *d_ptr = p->ptr;
    if (row_stride != nullptr)      // floats between consecutive rows of the variable's first dimension
        *row_stride = p->is_table ? (p->name == "emb" ? E->tab_ld : E->lin_ld) : (p->rank > 0 ? p->log_n / std::max<int64_t>(p->log_dims[0], 1) : 1);
    return DCTR_OK;
}

int dctr_set_global_step(dctr_handle E, int64_t step) {
    DCTR_REQUIRE(E && step >= 0, "bad argument");
    DCTR_HIP_CHECK(hipDeviceSynchronize());
    if (E->lag_period > 1) {                // rows are stamped relative to global_step: bring them to the present under the old one,
        DCTR_TRY(lag_flush_tables(E, nullptr, 0, false));
        DCTR_HIP_CHECK(hipDeviceSynchronize());
    }
    E->h_state.t = step;
    E->state_ready = false;                 // a state prepared ahead was derived from the old global_step
    DCTR_HIP_CHECK(hipMemcpy(&E->state->t, &step, sizeof(step), hipMemcpyHostToDevice));
    if (E->lag_period > 1) {                // ... and stamp them with the new one
        DCTR_TRY(lag_stamp(E->row_ts, E->rows, E->state, nullptr));
        DCTR_HIP_CHECK(hipDeviceSynchronize());
    }
    return DCTR_OK;
}
int dctr_get_global_step(dctr_handle E, int64_t* step) {
    DCTR_REQUIRE(E && step, "null argument");
    DCTR_HIP_CHECK(hipDeviceSynchronize());
    DCTR_HIP_CHECK(hipMemcpy(step, &E->state->t, sizeof(*step), hipMemcpyDeviceToHost));
    return DCTR_OK;
}

int dctr_train_step(dctr_handle E, const int32_t* d_ids, const float* d_vals, const float* d_labels, int B, float* h_loss,
                    void* stream) {
    DCTR_REQUIRE(E && d_ids && d_vals && d_labels, "null argument");
    DCTR_REQUIRE(!E->csr, "this handle's model takes CSR batches (dctr_train_step_csr)");
    hipStream_t st = as_stream(stream);
    DCTR_TRY(stage_inputs(E, d_ids, d_vals, d_labels, B, st));
    E->want_loss = h_loss != nullptr;
    DCTR_TRY(run_graph(E, E->train_graphs, B, true, st));
    E->last_B = B;
    if (h_loss) {
        float sc[4];
        DCTR_TRY(read_scalars(E, sc, st));
        // DeepFM.py:188-190: mean xent + l2_reg*(l2_loss(W) + l2_loss(V)), evaluated with the pre-update weights
        *h_loss = sc[0] / (float)B + E->cfg.l2_reg * 0.5f * (sc[1] + sc[2] + sc[3]);
        if (E->cfg.loss_sum) *h_loss = sc[0];       // canned heads report the batch SUM, no regulariser
    }
    return DCTR_OK;
}

int dctr_prefetch_ids(dctr_handle E, const int32_t* d_ids_next, int B) {
    DCTR_REQUIRE(E && d_ids_next, "null argument");
    if (E->csr || E->wnd || E->cfg.use_graph || E->cfg.shard_world > 1) return DCTR_OK;       // not a path that groups per step: no-op
    DCTR_REQUIRE(B > 0 && B <= E->MB, "batch %d outside (0, max_batch=%d]", B, E->MB);
    int slot = -1;
    for (int k = 0; k < DCTR_INPUT_SLOTS; ++k) if (d_ids_next == E->slot_ids[k]) slot = k;
    if (slot < 0) return DCTR_OK;           // foreign buffers are staged by copy: their address says nothing about their contents
    if (E->group_alt == nullptr) DCTR_TRY(group_create(E->rows, E->group->max_entries, E->K, &E->group_alt));
    // where the grouping of the next batch may start: behind the step's last st -> sw fork (beside scatter + table step, default),
    // or as soon as the grouping stream has drained its own work of the step (DCTR_PREGROUP_WAIT=none: beside the dense backward)
    static const int wait_mode = [] { const char* v = getenv("DCTR_PREGROUP_WAIT"); return v == nullptr ? -1 : (strcmp(v, "none") == 0 ? 0 : 1); }();
    hipStream_t sh = E->s_group;
    if (E->have_tail && wait_mode != 0) DCTR_HIP_CHECK(hipStreamWaitEvent(sh, E->ev_tail, 0));
    // (a slot filled by dctr_input_slot_fill whose copies may still be in flight: the grouping stream waits for them on the device)
    if (E->slot_fill_pending[slot].load(std::memory_order_acquire)) DCTR_HIP_CHECK(hipStreamWaitEvent(sh, E->slot_filled[slot], 0));
    DCTR_TRY(group_ids(E->group_alt, d_ids_next, B, E->F, sh, !tail_fused(E)));
    E->pre_ids = d_ids_next; E->pre_B = B; E->pre_valid = true;
    E->pre_slot = slot; E->pre_gen = E->slot_gen[slot].load();
    return DCTR_OK;
}

int dctr_tables_sync(dctr_handle E, void* stream) {
    DCTR_REQUIRE(E, "null argument");
    return lag_flush_tables(E, as_stream(stream), 0, false);
}

int dctr_prefetch_cancel(dctr_handle E) {
    DCTR_REQUIRE(E, "null argument");
    E->pre_valid = false;
    return DCTR_OK;
}

int dctr_input_slot_rewrite(dctr_handle E, int slot) {
    DCTR_REQUIRE(E && slot >= 0 && slot < DCTR_INPUT_SLOTS, "input slot %d outside [0, %d)", slot, DCTR_INPUT_SLOTS);
    E->slot_gen[slot]++;                    // (atomic: the input pipeline's thread calls this while the training thread enqueues steps)
    return DCTR_OK;
}

// ---- the H2D leg of the input pipeline (see the header): events and the copy stream are made on first use, under a lock -- the input
// thread and the training thread may both arrive first
static int slot_feed_init(dctr_engine* E) {
    if (E->slot_feed_ready.load(std::memory_order_acquire)) return DCTR_OK;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (E->slot_feed_ready.load(std::memory_order_acquire)) return DCTR_OK;
    DCTR_HIP_CHECK(hipSetDevice(E->device));
    {
        // the copy stream at the LOW priority level (with the grouping stream): streams share 4 hardware queues per priority level, and a
        // copy stream that lands on the training stream's queue serialises the H2D copies with the steps (A/B knob DCTR_PRIO_COPY)
        int least = 0, greatest = 0;
        DCTR_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const char* v = getenv("DCTR_PRIO_COPY");
        const int pr = v == nullptr ? least : (v[0] == 'h' ? greatest : (v[0] == 'l' ? least : 0));
        DCTR_HIP_CHECK(hipStreamCreateWithPriority(&E->s_copy, hipStreamNonBlocking, pr));
    }
    // no system-scope fence on either record: "filled" orders a copy before kernels of the same device, "released" tells the host
    // that kernels have finished READING -- nothing they wrote is for the host to see.  (A plain event's record writes the L2s
    // back: with the tables' dirty lines in them that was worth 60 us per step through the feeder.)
    for (int k = 0; k < DCTR_INPUT_SLOTS; ++k) {
        DCTR_HIP_CHECK(hipEventCreateWithFlags(&E->slot_filled[k], hipEventDisableTiming | hipEventDisableSystemFence));
        DCTR_HIP_CHECK(hipEventCreateWithFlags(&E->slot_released[k], hipEventDisableTiming | hipEventDisableSystemFence));
    }
    E->slot_feed_ready.store(1, std::memory_order_release);
    return DCTR_OK;
}

int dctr_input_slot_fill(dctr_handle E, int slot, const int32_t* h_ids, const float* h_vals, const float* h_labels, int B) {
    DCTR_REQUIRE(E && h_ids && h_vals, "null argument");
    DCTR_REQUIRE(slot >= 0 && slot < DCTR_INPUT_SLOTS, "input slot %d outside [0, %d)", slot, DCTR_INPUT_SLOTS);
    DCTR_REQUIRE(B > 0 && B <= E->MB, "batch %d outside (0, max_batch=%d]", B, E->MB);
    DCTR_REQUIRE(!E->csr, "CSR handles take their batches through dctr_train_step_csr");
    DCTR_TRY(slot_feed_init(E));
    DCTR_HIP_CHECK(hipSetDevice(E->device));                       // (the input thread's current device is its own affair)
    E->slot_gen[slot]++;                                           // a grouping prefetched from the slot's old contents is stale from here on
    const size_t n = (size_t)B * E->F;
    // a whole batch laid out like the slot ([ids | vals | labels], the first two padded to a multiple of 64 elements) is ONE copy: every host-to-device copy
    // brings its own system-scope acquire, and the step running beside it pays for each
    const size_t region = round_up((size_t)E->MB * E->F, 64);
    if (B == E->MB && h_labels != nullptr && reinterpret_cast<const char*>(h_vals) == reinterpret_cast<const char*>(h_ids) + region * 4 &&
        reinterpret_cast<const char*>(h_labels) == reinterpret_cast<const char*>(h_vals) + region * 4) {
        DCTR_HIP_CHECK(hipMemcpyAsync(E->slot_ids[slot], h_ids, (2 * region + (size_t)B) * 4, hipMemcpyHostToDevice, E->s_copy));
    } else {
        DCTR_HIP_CHECK(hipMemcpyAsync(E->slot_ids[slot], h_ids, n * 4, hipMemcpyHostToDevice, E->s_copy));
        DCTR_HIP_CHECK(hipMemcpyAsync(E->slot_vals[slot], h_vals, n * 4, hipMemcpyHostToDevice, E->s_copy));
        if (h_labels != nullptr) DCTR_HIP_CHECK(hipMemcpyAsync(E->slot_labels[slot], h_labels, (size_t)B * 4, hipMemcpyHostToDevice, E->s_copy));
    }
    DCTR_HIP_CHECK(hipEventRecord(E->slot_filled[slot], E->s_copy));
    E->slot_fill_pending[slot].store(1, std::memory_order_release);
    return DCTR_OK;
}

int dctr_input_slot_acquire(dctr_handle E, int slot, void* stream) {
    DCTR_REQUIRE(E && slot >= 0 && slot < DCTR_INPUT_SLOTS, "input slot %d outside [0, %d)", slot, DCTR_INPUT_SLOTS);
    if (E->slot_fill_pending[slot].load(std::memory_order_acquire))
        DCTR_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), E->slot_filled[slot], 0));
    return DCTR_OK;
}

int dctr_input_slot_release(dctr_handle E, int slot, void* stream) {
    DCTR_REQUIRE(E && slot >= 0 && slot < DCTR_INPUT_SLOTS, "input slot %d outside [0, %d)", slot, DCTR_INPUT_SLOTS);
    DCTR_TRY(slot_feed_init(E));
    DCTR_HIP_CHECK(hipEventRecord(E->slot_released[slot], as_stream(stream)));
    E->slot_release_valid[slot].store(1, std::memory_order_release);
    return DCTR_OK;
}

int dctr_input_slot_wait_released(dctr_handle E, int slot) {
    DCTR_REQUIRE(E && slot >= 0 && slot < DCTR_INPUT_SLOTS, "input slot %d outside [0, %d)", slot, DCTR_INPUT_SLOTS);
    if (E->slot_release_valid[slot].load(std::memory_order_acquire)) DCTR_HIP_CHECK(hipEventSynchronize(E->slot_released[slot]));
    return DCTR_OK;
}

// A stream of the engine's own for the caller's step calls.  Why: the legacy default stream (what a Python caller that never thinks about
// streams passes) pays for every cross-stream record / wait of the step and of the input slots' handshake -- the feeder-driven loop
// ran 312 us per step there against 272 us on a stream of its own (tools/feeder_breakdown.py); the Estimator trains on this one.
int dctr_main_stream(dctr_handle E, void** stream) {
    DCTR_REQUIRE(E && stream, "null argument");
    if (E->s_main == nullptr) {
        DCTR_HIP_CHECK(hipSetDevice(E->device));
        DCTR_HIP_CHECK(hipStreamCreateWithFlags(&E->s_main, hipStreamNonBlocking));
    }
    *stream = reinterpret_cast<void*>(E->s_main);
    return DCTR_OK;
}

int dctr_input_slot_ready(dctr_handle E, int slot, int* ready) {
    DCTR_REQUIRE(E && ready && slot >= 0 && slot < DCTR_INPUT_SLOTS, "input slot %d outside [0, %d)", slot, DCTR_INPUT_SLOTS);
    *ready = 1;
    if (E->slot_fill_pending[slot].load(std::memory_order_acquire)) {
        const hipError_t q = hipEventQuery(E->slot_filled[slot]);
        if (q == hipErrorNotReady) *ready = 0;
        else DCTR_HIP_CHECK(q);
    }
    return DCTR_OK;
}

int dctr_predict(dctr_handle E, const int32_t* d_ids, const float* d_vals, int B, float* d_prob, float* d_logit, void* stream) {
    DCTR_REQUIRE(E && d_ids && d_vals, "null argument");
    DCTR_REQUIRE(!E->csr, "this handle's model takes CSR batches (dctr_predict_csr)");
    hipStream_t st = as_stream(stream);
    DCTR_TRY(stage_inputs(E, d_ids, d_vals, nullptr, B, st));
    DCTR_TRY(lag_flush_tables(E, st, 0, false));        // (lagging rows, lag.h: the forward reads the tables as they are NOW)
    DCTR_TRY(run_graph(E, E->predict_graphs, B, false, st));
    E->last_B = B;
    if (d_prob) DCTR_HIP_CHECK(hipMemcpyAsync(d_prob, E->prob, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    if (d_logit) DCTR_HIP_CHECK(hipMemcpyAsync(d_logit, E->y, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    return DCTR_OK;
}

// ---- CSR (multi-hot) models: DIN (sum pooling) and ESMM ------------------------------------------------------------------------
}  // extern "C"

namespace {

// ESMM: exchange the primary (CTR) tower with the second (CVR) one, so that forward_rest / backward_dense run on either
void swap_tower(dctr_engine* E) {
    std::swap(E->mlp, E->mlp2);
    std::swap(E->h, E->h2);
    std::swap(E->dh, E->dh2);
    std::swap(E->hbn, E->hbn2);
    std::swap(E->bn_stats, E->bn_stats2);
    std::swap(E->p_out_w, E->p_out2_w);
    std::swap(E->p_out_b, E->p_out2_b);
    std::swap(E->dy, E->dy2);
    std::swap(E->dx_in, E->dx_in2);
}

// DIN attention pooling: exchange the main tower with the attention MLP, whose "batch" is the nnz entry rows X [nnz, 3K]
void swap_att(dctr_engine* E) {
    swap_tower(E);
    std::swap(E->x_in, E->x_att);
    std::swap(E->Din_ld, E->x_att_ld);
}

int csr_check(dctr_engine* E, const int32_t* off, const int32_t* ids, int nnz, int B) {
    DCTR_REQUIRE(E && off && ids, "null argument");
    DCTR_REQUIRE(E->csr, "this handle's model takes fixed-field batches (dctr_train_step), not CSR batches");
    DCTR_REQUIRE(B > 0 && B <= E->MB, "batch %d outside (0, max_batch=%d]", B, E->MB);
    DCTR_REQUIRE(nnz >= 0 && (int64_t)nnz <= E->max_entries, "nnz=%d exceeds max_entries=%lld", nnz, (long long)E->max_entries);
    return DCTR_OK;
}

// x_in[b, s*K:(s+1)*K] = sum of the slot's weighted rows, then the tower(s) and the head
int csr_forward(dctr_engine* E, const int32_t* off, const int32_t* ids, const float* wts, int nnz, int B, const float* y, const float* z,
                bool train, hipStream_t st, float* loss_shards = nullptr) {
    if (loss_shards == nullptr) loss_shards = E->scalars;
    const dctr_config& c = E->cfg;
    // a TRAINING step reads rows that may lag (lag.h) as of step t-1; everything else reads a flushed table
    const bool lag = train && lag_on(E);
    const LagView LV = lag_view(E);
    const LagView* Lp = lag ? &LV : nullptr;
    if (!train) DCTR_TRY(lag_flush_tables(E, st, 0, false));
    DCTR_TRY(lookup_sparse_slots_fwd(E->emb, E->rows, E->K, off, ids, wts, B * E->F, E->F, E->x_in, E->Din_ld, E->status, st, nnz, Lp));
    if (E->att_on && nnz > 0) {
        // attention units (DIN.py:152-177): the user slots' plain sums just written are replaced by attention-weighted sums
        DCTR_TRY(csr_entry_offsets(off, B * E->F, nnz, E->F, E->Din_ld, E->K, E->entry_off, st));
        DCTR_TRY(att_build_x(E->emb, E->rows, E->K, ids, wts, E->entry_off, E->pair_ad, nnz, E->x_in, E->Din_ld, E->x_att, st, Lp));
        swap_att(E);
        int rc = forward_rest(E, nnz, train, st);
        swap_att(E);
        DCTR_TRY(rc);
        const int wl = E->mlp2.back().out;
        DCTR_TRY(rowdot(E->h2.back(), wl, E->pp(E->p_out2_w), E->pp(E->p_out2_b), nnz, wl, E->att_sc, 0, st));
        DCTR_TRY(att_pool_fwd(off, ids, E->pair_ad, E->att_sc, B * E->F, E->F, E->K, E->x_att, E->att_w, E->x_in, E->Din_ld, st));
    }
    DCTR_TRY(forward_rest(E, B, train, st));
    if (c.model == DCTR_MODEL_DIN) {
        E->labels = const_cast<float*>(y);
        return head(E, B, B, y != nullptr, st, loss_shards);
    }
    swap_tower(E);
    int rc = forward_rest(E, B, train, st);
    swap_tower(E);
    DCTR_TRY(rc);
    return esmm_head(E->bn ? E->hbn.back() : E->h.back(), E->mlp.back().out, E->pp(E->p_out_w), E->pp(E->p_out_b), E->mlp.back().out,
                     E->bn ? E->hbn2.back() : E->h2.back(), E->mlp2.back().out, E->pp(E->p_out2_w), E->pp(E->p_out2_b), E->mlp2.back().out, y, z, B, 1.0f / (float)B,
                     c.ctr_task_wgt, E->y, E->y2, E->prob, E->prob2, E->prob3, E->dy, E->dy2, loss_shards, st);
}

}  // namespace

extern "C" {

int dctr_train_step_csr(dctr_handle E, const int32_t* d_offsets, const int32_t* d_ids, const float* d_weights, int nnz,
                        const float* d_y, const float* d_z, int B, float* h_loss, void* stream) {
    DCTR_TRY(csr_check(E, d_offsets, d_ids, nnz, B));
    const dctr_config& c = E->cfg;
    const bool esmm = c.model == DCTR_MODEL_ESMM;
    DCTR_REQUIRE(d_y != nullptr && (!esmm || d_z != nullptr), "labels missing (ESMM takes y and z)");
    hipStream_t st = as_stream(stream), sg = E->s_group, sw = E->s_wgrad;
    DCTR_TRY(step_state_advance(E->state, E->scalars, 4 * SUMSQ_SHARDS, st));
    E->want_loss = h_loss != nullptr;
    if (lag_on(E) && E->want_loss) DCTR_TRY(lag_flush_tables(E, st, -1, true));     // (the loss's l2 term: sum theta^2 of every row, as of t-1)
    // grouping of the batch's ids + the entries' slot offsets: beside the forward pass
    DCTR_TRY(fork(E, st, sg));
    const bool split_table = split_table_on(E);
    const bool fused = tail_fused(E);
    if (E->cfg.use_graph) E->group->slots_clean = false;
    DCTR_TRY(group_ids(E->group, d_ids, nnz, 1, sg, !fused));
    if (!E->att_on) DCTR_TRY(csr_entry_offsets(d_offsets, B * E->F, nnz, E->F, E->Din_ld, E->K, E->entry_off, sg));   // (attention: the forward computes them)
    // dense-exact table: the rows this batch does not touch step now, under the MLP (as in record_train)
    if (split_table) DCTR_TRY(step_untouched_rows(E, sg));
    DCTR_TRY(csr_forward(E, d_offsets, d_ids, d_weights, nnz, B, d_y, d_z, true, st));
    DCTR_TRY(backward_dense(E, B, st, sw, false));
    if (esmm) {
        swap_tower(E);
        int rc = backward_dense(E, B, st, sw, false);
        swap_tower(E);
        DCTR_TRY(rc);
        DCTR_TRY(add_inplace(E->dx_in, E->dx_in2, (int64_t)B * E->Din_ld, st));
    }
    const int32_t* goff = E->entry_off;
    if (E->att_on && nnz > 0) {
        // back through the attention units: scores -> attention MLP (the *2 tower over the entry rows) -> per-entry gradient rows
        DCTR_TRY(att_bwd_scores(d_ids, E->entry_off, E->pair_ad, nnz, E->K, E->dx_in, E->Din_ld, E->x_att, E->att_w, E->dy2, st));
        swap_att(E);
        int rc = backward_dense(E, nnz, st, sw, false);
        swap_att(E);
        DCTR_TRY(rc);
        DCTR_TRY(att_bwd_combine(d_offsets, d_ids, E->entry_off, E->pair_ad, nnz, B * E->F, E->F, E->K, E->dx_in, E->Din_ld, E->dx_in2,
                                 E->att_w, E->dub, E->entry_goff, st));
        goff = E->entry_goff;
    }
    // dense parameters on the weight-gradient stream (its last dW is done; the output layers' slabs were written on st)
    DCTR_TRY(fork(E, st, sw));
    DCTR_TRY(opt_dense_arena(c.optimizer, &E->state->hyper, E->h_state.hyper, E->theta, E->as0, E->as1, E->parts, E->meta, E->n_blocks,
                             nullptr, 1, E->scalars + 3 * SUMSQ_SHARDS, sw));
    if (E->gemm_mode == 1) DCTR_TRY(wplanes_written(E, 0, (int)E->params.size() - 1, sw));
    // table: per-entry gradient = weight * dL/dx[slot], segment-summed per distinct id, then the optimizer (DIN.py:222: the
    // l2_loss(Feat_Emb) term makes the table gradient dense, as for the fixed-field models)
    DCTR_TRY(fork(E, sg, st));
    if (fused) {
        if (nnz > 0)
            DCTR_TRY(embed_scatter_apply(E->group, c.optimizer, &E->state->hyper, E->h_state.hyper, E->emb, E->emb_s0, E->emb_s1, nullptr, nullptr,
                                         nullptr, c.l2_reg, lag_on(E) ? nullptr : E->scalars + SUMSQ_SHARDS, lag_on(E) ? nullptr : E->scalars + 2 * SUMSQ_SHARDS,
                                         E->dx_in, 4, nullptr, 0, nullptr, nullptr, nullptr, d_weights, nnz, 1, E->K, DCTR_GATHER_RAW, st, 1, goff,
                                         lag_on(E) ? E->row_ts : nullptr, lag_on(E) ? E->state : nullptr, E->tab_ld, E->lin_ld));
    } else {
    if (nnz > 0)
        DCTR_TRY(embed_scatter_bwd(E->group, E->dx_in, 4, nullptr, 0, nullptr, nullptr, nullptr, d_weights, nnz, 1, E->K, DCTR_GATHER_RAW,
                                   E->group->gemb, nullptr, st, 1, goff));
    DCTR_TRY(opt_table(c.optimizer, &E->state->hyper, E->h_state.hyper, c.table_mode, E->rows, E->K, E->emb, E->emb_s0, E->emb_s1,
                       nullptr, nullptr, nullptr, E->group->slot, E->group->uniq, E->group->counters, E->group->max_entries,
                       E->group->gemb, nullptr, c.l2_reg, E->scalars + SUMSQ_SHARDS, E->scalars + 2 * SUMSQ_SHARDS, st, nullptr,
                       split_table ? OPT_PASS_TOUCHED : OPT_PASS_ALL, E->tab_ld, E->lin_ld));
    }
    DCTR_TRY(fork(E, sw, st));
    E->last_B = B;
    if (h_loss) {
        float sc[4];
        DCTR_TRY(read_scalars(E, sc, st));
        *h_loss = sc[0] / (float)B + c.l2_reg * 0.5f * sc[1];       // DIN.py:222 / DeepCvrMTL.py:225 (the head sums the task-weighted terms)
    }
    return DCTR_OK;
}

int dctr_predict_csr(dctr_handle E, const int32_t* d_offsets, const int32_t* d_ids, const float* d_weights, int nnz, int B,
                     float* d_out0, float* d_out1, float* d_out2, void* stream) {
    DCTR_TRY(csr_check(E, d_offsets, d_ids, nnz, B));
    hipStream_t st = as_stream(stream);
    DCTR_TRY(csr_forward(E, d_offsets, d_ids, d_weights, nnz, B, nullptr, nullptr, false, st));
    E->last_B = B;
    const bool esmm = E->cfg.model == DCTR_MODEL_ESMM;
    const float* src[3] = {E->prob, esmm ? E->prob2 : E->y, esmm ? E->prob3 : nullptr};
    float* dst[3] = {d_out0, d_out1, d_out2};
    for (int k = 0; k < 3; ++k)
        if (dst[k] && src[k]) DCTR_HIP_CHECK(hipMemcpyAsync(dst[k], src[k], (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    return DCTR_OK;
}

// mode EVAL of the CSR models: loss + AUC counters accumulate exactly as dctr_eval_batch does (dctr_eval_reset / dctr_eval_result);
// ESMM also counts CVR_AUC = auc(z, pcvr) and CTCVR_AUC = auc(z, pctcvr) (DeepCvrMTL.py:231-235), read with dctr_eval_auc_extra
int dctr_eval_batch_csr(dctr_handle E, const int32_t* d_offsets, const int32_t* d_ids, const float* d_weights, int nnz,
                        const float* d_y, const float* d_z, int B, void* stream) {
    DCTR_TRY(csr_check(E, d_offsets, d_ids, nnz, B));
    const bool esmm = E->cfg.model == DCTR_MODEL_ESMM;
    DCTR_REQUIRE(d_y != nullptr && (!esmm || d_z != nullptr), "labels missing (ESMM takes y and z)");
    DCTR_TRY(csr_forward(E, d_offsets, d_ids, d_weights, nnz, B, d_y, d_z, false, as_stream(stream), E->eval_scalars));
    DCTR_TRY(dctr_auc_update(d_y, E->prob, B, E->auc_counts, stream));
    if (esmm) {
        DCTR_TRY(dctr_auc_update(d_z, E->prob2, B, E->auc_counts + 800, stream));
        DCTR_TRY(dctr_auc_update(d_z, E->prob3, B, E->auc_counts + 1600, stream));
    }
    E->eval_examples += B;
    E->last_B = B;
    return DCTR_OK;
}

int dctr_eval_auc_extra(dctr_handle E, int which, float* h_auc, void* stream) {
    DCTR_REQUIRE(E && h_auc && which >= 0 && which < 3, "which must be 0 (first output), 1 or 2");
    return dctr_auc_result(E->auc_counts + 800 * which, h_auc, stream);
}

// ---- mode EVAL (DeepFM.py:193-201): streaming loss + tf.metrics.auc over an eval set --------------------------------------
// (x: n floats in rows of `width`, `ld` apart -- a dense variable is one row)
__global__ void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out, int width = 0, int ld = 0) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = width > 0 ? x[(i / width) * ld + (i % width)] : x[i];
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

int dctr_eval_reset(dctr_handle E, void* stream) {
    DCTR_REQUIRE(E, "null handle");
    DCTR_HIP_CHECK(hipMemsetAsync(E->auc_counts, 0, 3 * 800 * sizeof(int64_t), as_stream(stream)));
    DCTR_HIP_CHECK(hipMemsetAsync(E->eval_scalars, 0, 2 * SUMSQ_SHARDS * sizeof(float), as_stream(stream)));
    E->eval_examples = 0;
    return DCTR_OK;
}

int dctr_eval_batch(dctr_handle E, const int32_t* d_ids, const float* d_vals, const float* d_labels, int B, void* stream) {
    DCTR_REQUIRE(E && d_ids && d_vals && d_labels, "null argument");
    hipStream_t st = as_stream(stream);
    DCTR_TRY(stage_inputs(E, d_ids, d_vals, d_labels, B, st));
    DCTR_TRY(lag_flush_tables(E, st, 0, false));
    DCTR_TRY(forward(E, B, false, st));
    DCTR_TRY(head(E, B, B, true, st, E->eval_scalars));      // dy is written too (unused in EVAL)
    DCTR_TRY(dctr_auc_update(E->labels, E->prob, B, E->auc_counts, stream));
    E->eval_examples += B;
    E->last_B = B;
    return DCTR_OK;
}

int dctr_eval_result(dctr_handle E, float* h_auc, float* h_loss, int64_t* h_examples, void* stream) {
    DCTR_REQUIRE(E, "null handle");
    hipStream_t st = as_stream(stream);
    if (h_auc) DCTR_TRY(dctr_auc_result(E->auc_counts, h_auc, stream));
    if (h_loss) {
        // loss of DeepFM.py:188-190 over the eval set: mean xent + l2_reg * sum l2_loss(regularised variables)
        DCTR_TRY(lag_flush_tables(E, st, 0, false));
        DCTR_HIP_CHECK(hipMemsetAsync(E->eval_scalars + SUMSQ_SHARDS, 0, sizeof(float), st));
        for (auto& p : E->params)
            if (p.l2 != 0.f) {
                const bool rec = p.is_table && E->table_rec != nullptr;
                sumsq_kernel<<<256, 256, 0, st>>>(p.ptr, p.n, E->eval_scalars + SUMSQ_SHARDS, rec ? (p.name == "emb" ? E->K : 1) : 0,
                                                  rec ? E->tab_ld : 0);
            }
        float sc[SUMSQ_SHARDS + 1];
        DCTR_HIP_CHECK(hipMemcpyAsync(sc, E->eval_scalars, sizeof(sc), hipMemcpyDeviceToHost, st));
        DCTR_HIP_CHECK(hipStreamSynchronize(st));
        double xent = 0.0;
        for (int j = 0; j < SUMSQ_SHARDS; ++j) xent += sc[j];
        *h_loss = (E->eval_examples > 0 ? (float)(xent / (double)E->eval_examples) : 0.f) + E->cfg.l2_reg * 0.5f * sc[SUMSQ_SHARDS];
    }
    if (h_examples) *h_examples = E->eval_examples;
    return DCTR_OK;
}

int dctr_input_slot(dctr_handle E, int slot, int32_t** d_ids, float** d_vals, float** d_labels) {
    DCTR_REQUIRE(E && slot >= 0 && slot < DCTR_INPUT_SLOTS, "slot must be in [0,%d)", DCTR_INPUT_SLOTS);
    if (d_ids) *d_ids = E->slot_ids[slot];
    if (d_vals) *d_vals = E->slot_vals[slot];
    if (d_labels) *d_labels = E->slot_labels[slot];
    return DCTR_OK;
}

// In-step duration of the MLP's forward GEMMs (the `roofline` kernel of bench.py), by hipEvents on the step's own stream, so the figure
// is the kernel as it runs INSIDE the step (beside the grouping / background table pass), comparable with rocprofv3's per-kernel trace.
// enable=1: two records around the first layer's launch (a bracket: the interval holds two barrier packets besides the kernel);
// enable=2: every forward layer's launch carries its own start / stop events (hipExtLaunchKernel): the dispatch alone.  Every 32nd step
// is timed, up to 4096 launches; enable=0 stops and returns the average over all timed launches; dctr_step_timer_layer reads one layer.
static int timer_average(dctr_handle E, int layer, float* h_avg_ms, int* h_count) {
    double tot = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < E->timer_n; i += 2) {
        if (layer >= 0 && (i / 2 >= E->timer_layer.size() || E->timer_layer[i / 2] != layer)) continue;
        DCTR_HIP_CHECK(hipEventSynchronize(E->timer_ev[i + 1]));
        float ms = 0.f;
        DCTR_HIP_CHECK(hipEventElapsedTime(&ms, E->timer_ev[i], E->timer_ev[i + 1]));
        tot += ms; ++n;
    }
    if (h_avg_ms) *h_avg_ms = n ? (float)(tot / n) : 0.f;
    if (h_count) *h_count = n;
    return DCTR_OK;
}
int dctr_step_timer(dctr_handle E, int enable, float* h_avg_ms, int* h_count) {
    DCTR_REQUIRE(E && enable >= 0 && enable <= 2, "dctr_step_timer: enable is 0, 1 or 2");
    if (enable) {
        if (E->timer_ev.empty()) {
            E->timer_ev.resize(8192);
            for (auto& ev : E->timer_ev) DCTR_HIP_CHECK(hipEventCreate(&ev));
        }
        E->timer_n = 0;
        E->timer_tick = 0;
        E->timer_layer.clear();
        E->timer_mode = enable;
        E->timer_on = true;
        return DCTR_OK;
    }
    E->timer_on = false;
    return timer_average(E, -1, h_avg_ms, h_count);
}
int dctr_step_timer_layer(dctr_handle E, int layer, float* h_avg_ms, int* h_count) {
    DCTR_REQUIRE(E && layer >= 0 && !E->timer_on, "dctr_step_timer_layer: after dctr_step_timer(h, 0, ..), layer >= 0");
    return timer_average(E, layer, h_avg_ms, h_count);
}

int dctr_set_dense_input(dctr_handle E, const float* d_dense) {
    DCTR_REQUIRE(E, "null handle");
    DCTR_REQUIRE(E->wnd && E->n_dense > 0, "this model has no dense inputs");
    E->dense = d_dense;
    return DCTR_OK;
}

int dctr_check_ids(dctr_handle E, void* stream) {
    DCTR_REQUIRE(E, "null handle");
    int32_t s[2] = {0, 0}, lag_over = 0;
    DCTR_HIP_CHECK(hipMemcpyAsync(s, E->status, sizeof(s), hipMemcpyDeviceToHost, as_stream(stream)));
    int32_t lag_over_alt = 0;
    if (E->lag_period > 1) {
        DCTR_HIP_CHECK(hipMemcpyAsync(&lag_over, reinterpret_cast<const char*>(E->state) + offsetof(StepState, lag_overflow), sizeof(lag_over),
                                      hipMemcpyDeviceToHost, as_stream(stream)));
        // (the other StepState too: a flag raised late in the last step sits in the state that step ran on, which may be either by now)
        if (E->state_alt != nullptr)
            DCTR_HIP_CHECK(hipMemcpyAsync(&lag_over_alt, reinterpret_cast<const char*>(E->state_alt) + offsetof(StepState, lag_overflow), sizeof(lag_over_alt),
                                          hipMemcpyDeviceToHost, as_stream(stream)));
    }
    DCTR_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    lag_over |= lag_over_alt;
    if (lag_over != 0) {         // (lag.h lag_replay_rows: a row further behind than the replay loop reaches -- never under the engine's own schedule)
        set_error("a table row was found more than %d steps behind global_step: the time-blocked sweep's invariant is broken (stamps wrapped, or the "
                  "owner-side table API was driven without its sweep)", LAG_MAX_PERIOD);
        return DCTR_ERR_INVALID_ARG;
    }
    if (s[0] != 0) {
        DCTR_HIP_CHECK(hipMemsetAsync(E->status, 0, sizeof(s), as_stream(stream)));
        set_error("indices = %d is not in [0, %lld)", s[1], (long long)E->rows);   // GatherOp's message [TF-1.4]
        return DCTR_ERR_INVALID_ARG;
    }
    return DCTR_OK;
}

int dctr_debug_tensor(dctr_handle E, const char* name, float** d_ptr, int64_t* n_elems, int* ld) {
    DCTR_REQUIRE(E && name && d_ptr, "null argument");
    const int B = E->last_B;
    const std::string s(name);
    float* p = nullptr;
    int64_t n = 0;
    int l = 1;
    if (s == "e") { p = E->e; n = (int64_t)B * E->e_ld; l = E->e_ld; }
    else if (s == "y_w") { p = E->yw; n = B; }
    else if (s == "y_v") { p = E->yv; n = B; }
    else if (s == "y_d") { p = E->yd; n = B; }
    else if (s == "y") { p = E->y; n = B; }
    else if (s == "prob") { p = E->prob; n = B; }
    else if (s == "dy") { p = E->dy; n = B; }
    else if (s == "sum") { p = E->S; n = (int64_t)B * E->K; l = E->K; }
    else if (s == "x_in") { p = E->x_in; n = (int64_t)B * E->Din_ld; l = E->Din_ld; }
    else if (s == "dx_in") { p = E->dx_in; n = (int64_t)B * E->Din_ld; l = E->Din_ld; }
    else if (s == "x_cross" && E->xs) { p = E->xs + (size_t)E->cfg.cross_layers * B * E->D; n = (int64_t)B * E->D; l = E->D; }
    else { set_error("no debug tensor named '%s'", name); return DCTR_ERR_NOT_FOUND; }
    *d_ptr = p;
    if (n_elems) *n_elems = n;
    if (ld) *ld = l;
    return DCTR_OK;
}

// ---- row-sharded path (declared in deepctr_hip.h under "row-sharded multi-GPU path") ---------------------------------
// Rows and row gradients travel as packed [K+4]-float records (shard.hip).  The owner keeps TWO grouping states so that the rows
// of step t+1 can be grouped (dctr_table_group_rows, on a side stream) while step t still uses its own.
static Group* owner_group(dctr_engine* E, int which) {
    if (which == 0) return E->group;
    if (E->group_alt == nullptr && group_create(E->rows, E->group->max_entries, E->K, &E->group_alt) != DCTR_OK) return nullptr;
    return E->group_alt;
}

int dctr_table_gather_packed(dctr_handle E, const int32_t* d_rows, int n, float* d_out, void* stream) {
    DCTR_REQUIRE(E && (n == 0 || (d_rows && d_out)), "null argument");
    DCTR_REQUIRE(n >= 0 && (int64_t)n <= E->group->max_entries, "too many rows requested (%d)", n);
    if (owner_lag(E)) {           // rows may lag (lag.h): shipped as of the present
        const LagView L = lag_view(E);
        return pack_table_rows(E->emb, E->lin, E->rows, E->K, d_rows, n, d_out, E->status, as_stream(stream), &L, E->tab_ld, E->lin_ld);
    }
    return pack_table_rows(E->emb, E->lin, E->rows, E->K, d_rows, n, d_out, E->status, as_stream(stream), nullptr, E->tab_ld, E->lin_ld);
}

int dctr_table_group_rows(dctr_handle E, int which, const int32_t* d_rows, int n, void* stream) {
    DCTR_REQUIRE(E && (which == 0 || which == 1) && (n == 0 || d_rows), "bad argument");
    Group* G = owner_group(E, which);
    DCTR_REQUIRE(G != nullptr, "owner group allocation failed");
    DCTR_REQUIRE(n >= 0 && (int64_t)n <= G->max_entries, "too many rows (%d)", n);
    return group_ids(G, n > 0 ? d_rows : E->ids, n, 1, as_stream(stream), !owner_lag(E));
}

int dctr_table_apply_packed(dctr_handle E, int which, int n, const float* d_grads, void* stream) {
    DCTR_REQUIRE(E && (which == 0 || which == 1) && (n == 0 || d_grads), "bad argument");
    Group* G = owner_group(E, which);
    DCTR_REQUIRE(G != nullptr, "owner group allocation failed");
    DCTR_REQUIRE(n >= 0 && (int64_t)n <= G->max_entries, "too many rows (%d)", n);
    hipStream_t st = as_stream(stream);
    const dctr_config& c = E->cfg;
    const int P = E->K + 4;
    const bool lag = owner_lag(E);
    if (lag && !E->want_loss && G->gemb_clean) {
        // time-blocked sweep (lag.h): one block of this shard's untouched rows advances to the step, the rows that received
        // gradients are advanced to t-1, stepped and stamped by the fused scatter + optimizer launch
        E->lag_dirty = true;
        DCTR_TRY(lag_sweep(E->K, E->rows, E->emb, E->emb_s0, E->emb_s1, E->lin, E->lin_s0, E->lin_s1, G->slot, E->row_ts, E->state, c.l2_reg,
                           E->lag_period, st, E->tab_ld, E->lin_ld));
        if (n > 0)
            DCTR_TRY(embed_scatter_apply(G, c.optimizer, &E->state->hyper, E->h_state.hyper, E->emb, E->emb_s0, E->emb_s1, E->lin, E->lin_s0,
                                         E->lin_s1, c.l2_reg, nullptr, nullptr, d_grads, P, nullptr, 0, nullptr, nullptr,
                                         E->lin ? d_grads + E->K : nullptr, E->ones, n, 1, E->K, DCTR_GATHER_RAW, st, P, nullptr, E->row_ts, E->state,
                                         E->tab_ld, E->lin_ld));
        return DCTR_OK;
    }
    // a loss-reporting step needs sum theta^2 of every row (and a group whose compact rows a plain scatter has used cannot take the
    // fused launch): all rows to t-1, the classic sweep below (which sums), all stamped t
    if (lag) {
        DCTR_TRY(lag_flush_tables(E, st, -1, false));
        DCTR_TRY(lag_stamp(E->row_ts, E->rows, E->state, st));
    }
    if (n > 0)      // n "examples" of one field each, value 1: dE = the record's K gradient floats, dy = its linear-weight gradient
        DCTR_TRY(embed_scatter_bwd(G, d_grads, P, nullptr, 0, nullptr, nullptr, E->lin ? d_grads + E->K : nullptr, E->ones, n, 1, E->K,
                                   DCTR_GATHER_RAW, G->gemb, E->lin ? G->glin : nullptr, st, P));
    return opt_table(c.optimizer, &E->state->hyper, E->h_state.hyper, c.table_mode, E->rows, E->K, E->emb, E->emb_s0, E->emb_s1,
                     E->lin, E->lin_s0, E->lin_s1, G->slot, G->uniq, G->counters, G->max_entries, G->gemb, G->glin, c.l2_reg,
                     E->scalars + SUMSQ_SHARDS, E->scalars + 2 * SUMSQ_SHARDS, st, nullptr, OPT_PASS_ALL, E->tab_ld, E->lin_ld);
}

}  // extern "C"

// forward (+ backward when train) of one rank's examples against the packed rows received from their owners.  Training also
// advances the step state (global_step, lr_t, dropout seed; zeroes the loss scalars).  Weight gradients run on the engine's
// side stream beside the dgrad chain; join_wgrad: wait for them on `stream` before returning (the native driver instead
// keeps using the side stream for the dense all-reduce + optimizer and joins at the end of the step).
int sharded_forward_backward(dctr_engine* E, const float* d_rows, int n_rows, const int32_t* d_idx, const float* d_vals,
                             const float* d_labels, int B, int global_batch, bool train, bool join_wgrad, hipStream_t st) {
    DCTR_REQUIRE(E && d_rows && d_idx && d_vals, "null argument");
    DCTR_REQUIRE(!E->wnd, "canned-estimator models are not row-sharded");
    DCTR_REQUIRE(!E->bn || global_batch == B || (E->bn_sync.all_reduce != nullptr && (int64_t)B * E->bn_sync.world == global_batch),
                 "batch_norm over data-parallel ranks needs the cross-rank sum (dctr_set_stat_sync) and equal per-rank batches");
    DCTR_REQUIRE(B > 0 && B <= E->MB && global_batch >= B, "bad batch sizes B=%d global=%d", B, global_batch);
    DCTR_REQUIRE(!train || d_labels, "labels required for training");
    hipStream_t sw = E->s_wgrad;
    const size_t n = (size_t)B * E->F;
    const int P = E->K + 4;
    E->state_ready = false;
    // (on st: see record_train.)  row0: dropout masks are a function of the GLOBAL example row (common.h dropout_row0) -- with equal
    // per-rank batches this rank's examples are rows [rank B, (rank + 1) B) of the step's global batch, and N ranks draw exactly the
    // masks one rank draws on that batch (tests/test_distributed.py "+dropout"); unequal batches: the local row, as before
    if (train) DCTR_TRY(step_state_advance(E->state, E->scalars, 4 * SUMSQ_SHARDS, st,
                                           (int64_t)B * E->cfg.shard_world == (int64_t)global_batch ? (uint64_t)E->cfg.shard_rank * (uint64_t)B : 0ull));
    // inputs that already live in one of the engine's input slots are read in place (no staging copy at the head of the step)
    for (int k = 0; k < DCTR_INPUT_SLOTS; ++k)
        if (d_vals == E->slot_vals[k] && (d_labels == nullptr || d_labels == E->slot_labels[k])) {
            E->cur_slot = k; E->ids = E->slot_ids[k]; E->vals = E->slot_vals[k]; E->labels = E->slot_labels[k];
        }
    if (d_vals != E->vals) DCTR_HIP_CHECK(hipMemcpyAsync(E->vals, d_vals, n * 4, hipMemcpyDeviceToDevice, st));
    if (d_labels && d_labels != E->labels) DCTR_HIP_CHECK(hipMemcpyAsync(E->labels, d_labels, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    const int mode = gather_mode(E);
    float* red = mode == DCTR_GATHER_FM ? E->yv : (mode == DCTR_GATHER_BI ? E->x_in : nullptr);
    DCTR_TRY(embed_gather_strided(d_rows, P, E->lin ? d_rows + E->K : nullptr, P, n_rows, d_idx, E->vals, B, E->F, E->K, mode, E->e,
                                  E->e_ld, E->lin ? E->yw : nullptr, E->S, red, E->status, st));
    DCTR_TRY(forward_rest(E, B, train, st));
    DCTR_TRY(head(E, B, global_batch, d_labels != nullptr, st, nullptr, train));
    if (train) {
        DCTR_TRY(backward_dense(E, B, st, sw, false));
        if (join_wgrad) DCTR_TRY(fork(E, sw, st));
    }
    E->last_B = B;
    return DCTR_OK;
}

extern "C" {

int dctr_set_stat_sync(dctr_handle E, int (*all_reduce_f32)(void* ctx, int channel, float* d_buf, int64_t n, void* stream), void* ctx, int world) {
    DCTR_REQUIRE(E != nullptr && world >= 1 && (world == 1 || all_reduce_f32 != nullptr), "bad argument");
    E->bn_sync.all_reduce = all_reduce_f32;
    E->bn_sync.ctx = ctx;
    E->bn_sync.world = world;
    return DCTR_OK;
}

int dctr_sharded_forward_backward(dctr_handle E, const float* d_rows, int n_rows, const int32_t* d_idx, const float* d_vals,
                                  const float* d_labels, int B, int global_batch, int train, void* stream) {
    return sharded_forward_backward(E, d_rows, n_rows, d_idx, d_vals, d_labels, B, global_batch, train != 0, true, as_stream(stream));
}

// per-distinct-id gradients of this rank's examples (segment sum into g's compact rows), packed in send order:
// d_out[upos[u]] = { gemb[u, :], glin[u], 0, 0, 0 }
int dctr_sharded_pack_row_grads(dctr_handle E, dctr_group_t g, int B, const int32_t* d_upos, float* d_out, void* stream) {
    DCTR_REQUIRE(E && g && d_upos && d_out, "null argument");
    Group* G = reinterpret_cast<Group*>(g);
    hipStream_t st = as_stream(stream);
    const int mode = gather_mode(E);
    const float* dE = mode == DCTR_GATHER_BI ? nullptr : E->dE;
    const float* coef = mode == DCTR_GATHER_FM ? E->dy : (mode == DCTR_GATHER_BI ? E->dx_in : nullptr);
    DCTR_TRY(embed_scatter_bwd(G, dE, E->dE_ld, E->e, E->e_ld, E->S, coef, E->lin ? E->dy : nullptr, E->vals, B, E->F, E->K, mode,
                               G->gemb, E->lin ? G->glin : nullptr, st));
    return pack_unique_grads(G, E->lin ? G->glin : nullptr, d_upos, d_out, st);
}

int dctr_dense_grads(dctr_handle E, float** d_flat, int64_t* n, void* stream) {
    DCTR_REQUIRE(E && d_flat && n, "null argument");
    DCTR_TRY(opt_dense_arena(E->cfg.optimizer, &E->state->hyper, E->h_state.hyper, E->theta, E->as0, E->as1, E->parts, E->meta,
                             E->n_blocks, E->gflat, 0, nullptr, as_stream(stream)));
    *d_flat = E->gflat;
    *n = E->arena_n;
    return DCTR_OK;
}

int dctr_dense_apply(dctr_handle E, void* stream) {
    DCTR_REQUIRE(E, "null handle");
    DCTR_TRY(opt_dense_arena(E->cfg.optimizer, &E->state->hyper, E->h_state.hyper, E->theta, E->as0, E->as1, E->gflat, E->meta_flat,
                             E->n_blocks, nullptr, 1, E->scalars + 3 * SUMSQ_SHARDS, as_stream(stream)));
    return E->gemm_mode == 1 ? wplanes_written(E, 0, (int)E->params.size() - 1, as_stream(stream)) : DCTR_OK;
}

int dctr_read_scalars(dctr_handle E, float h_out[4], void* stream) {
    DCTR_REQUIRE(E && h_out, "null argument");
    return read_scalars(E, h_out, as_stream(stream));
}

// ---- AFM's interaction layer as an op (AFM.py:127-158), on an AFM handle: the attention variables are the handle's (dctr_param_set by
// their TF names), the workspace too.
int dctr_afm_fwd(dctr_handle E, const float* d_e, int e_ld, int B, int train, float* d_y_emb, int y_ld, float* d_att, void* stream) {
    DCTR_REQUIRE(E && d_e && d_y_emb, "null argument");
    DCTR_REQUIRE(E->cfg.model == DCTR_MODEL_AFM, "dctr_afm_fwd: the handle's model is not afm");
    DCTR_REQUIRE(B > 0 && B <= E->MB && e_ld >= E->D && y_ld >= E->K, "dctr_afm_fwd: bad sizes B=%d e_ld=%d y_ld=%d", B, e_ld, y_ld);
    hipStream_t st = as_stream(stream);
    DCTR_HIP_CHECK(hipMemcpy2DAsync(E->e, (size_t)E->e_ld * 4, d_e, (size_t)e_ld * 4, (size_t)E->D * 4, B, hipMemcpyDeviceToDevice, st));
    DCTR_TRY(afm_forward(E, B, train != 0, st));
    DCTR_HIP_CHECK(hipMemcpy2DAsync(d_y_emb, (size_t)y_ld * 4, E->x_in, (size_t)E->Din_ld * 4, (size_t)E->K * 4, B, hipMemcpyDeviceToDevice, st));
    if (d_att != nullptr) DCTR_HIP_CHECK(hipMemcpyAsync(d_att, E->att, (size_t)B * E->P * 4, hipMemcpyDeviceToDevice, st));
    E->last_B = B;
    return DCTR_OK;
}

int dctr_afm_bwd(dctr_handle E, const float* d_dy_emb, int dy_ld, int B, float* d_dE, int de_ld, void* stream) {
    DCTR_REQUIRE(E && d_dy_emb && d_dE, "null argument");
    DCTR_REQUIRE(E->cfg.model == DCTR_MODEL_AFM, "dctr_afm_bwd: the handle's model is not afm");
    DCTR_REQUIRE(B > 0 && B == E->last_B && dy_ld >= E->K && de_ld >= E->D, "dctr_afm_bwd: call dctr_afm_fwd on the same batch first (B=%d, last %d)", B, E->last_B);
    hipStream_t st = as_stream(stream);
    DCTR_HIP_CHECK(hipMemcpy2DAsync(E->dx_in, (size_t)E->Din_ld * 4, d_dy_emb, (size_t)dy_ld * 4, (size_t)E->K * 4, B, hipMemcpyDeviceToDevice, st));
    DCTR_TRY(afm_interaction_backward(E, B, st, E->s_wgrad));
    DCTR_TRY(fork(E, E->s_wgrad, st));
    DCTR_HIP_CHECK(hipMemcpy2DAsync(d_dE, (size_t)de_ld * 4, E->dE_buf, (size_t)E->D * 4, (size_t)E->D * 4, B, hipMemcpyDeviceToDevice, st));
    return DCTR_OK;
}

int dctr_last_outputs(dctr_handle E, float** d_prob, float** d_logit) {
    DCTR_REQUIRE(E, "null handle");
    if (d_prob) *d_prob = E->prob;
    if (d_logit) *d_logit = E->y;
    return DCTR_OK;
}

// Times one stage of the step on the engine's current buffers (last batch): `iters` back-to-back executions are
// captured into a graph (so the host launch rate does not bound the measurement) and bracketed by two hipEvents
// recorded on `stream`.  Stages mutate state exactly as a step does (the optimizer stages do update weights).
int dctr_time_kernel(dctr_handle E, const char* kernel, int iters, float* h_ms_per_launch, void* stream) {
    DCTR_REQUIRE(E && kernel && h_ms_per_launch && iters > 0, "bad argument");
    DCTR_REQUIRE(E->last_B > 0, "run a train step first: stages are timed on the last batch's buffers");
    hipStream_t st = as_stream(stream);
    const int B = E->last_B;
    const std::string s(kernel);
    const dctr_config& c = E->cfg;
    // the stages are captured once and REPLAYED: a grouping in them must not rely on what this enqueue knows about the slot words
    E->group->slots_clean = false;
    DCTR_TRY(join_deferred(E, st));                    // (the stages are captured on a stream of their own: nothing of the last step may be pending)
    DCTR_TRY(lag_flush_tables(E, st, 0, false));       // (lagging rows: the classic stages below assume every row is current)
    E->lag_suspended = s != "train_step";              // (a replayed single stage does not advance global_step: classic kernels)
    auto stage = [&](hipStream_t cs) -> int {
        if (s == "embed_gather") {
            const int mode = gather_mode(E);
            float* red = mode == DCTR_GATHER_FM ? E->yv : (mode == DCTR_GATHER_BI ? E->x_in : nullptr);
            return embed_gather_strided(E->emb, E->tab_ld, E->lin, E->lin_ld, E->rows, E->ids, E->vals, B, E->F, E->K, mode, E->e, E->e_ld,
                                        E->lin ? E->yw : nullptr, E->S, red, E->status, cs);
        }
        if (s == "forward") return forward(E, B, true, cs);
        if (s == "head") return head(E, B, B, true, cs, nullptr, true);
        if (s == "backward_dense") return backward_dense(E, B, cs, cs);
        if (s == "group_ids") return group_ids(E->group, E->ids, B, E->F, cs, !tail_fused(E));
        if (s == "tail") {          // grouping + the step's tail (fused: ONE launch; else scatter + touched-rows table step): repeatable as a pair
            DCTR_TRY(group_ids(E->group, E->ids, B, E->F, cs, !tail_fused(E)));
            return scatter_and_step_tables(E, B, cs, nullptr, split_table_on(E) ? OPT_PASS_TOUCHED : OPT_PASS_ALL);
        }
        if (s == "scatter") {
            const int mode = gather_mode(E);
            const float* dE = mode == DCTR_GATHER_BI ? nullptr : E->dx_in;
            const float* coef = mode == DCTR_GATHER_FM ? E->dy : (mode == DCTR_GATHER_BI ? E->dx_in : nullptr);
            return embed_scatter_bwd(E->group, dE, E->Din_ld, E->e, E->e_ld, E->S, coef, E->lin ? E->dy : nullptr, E->vals, B,
                                     E->F, E->K, mode, E->group->gemb, E->lin ? E->group->glin : nullptr, cs);
        }
        if (s == "opt_table")
            return opt_table(c.optimizer, &E->state->hyper, E->h_state.hyper, c.table_mode, E->rows, E->K, E->emb, E->emb_s0,
                             E->emb_s1, E->lin, E->lin_s0, E->lin_s1, E->group->slot, E->group->uniq, E->group->counters,
                             E->group->max_entries, E->group->gemb, E->group->glin, c.l2_reg, E->scalars + SUMSQ_SHARDS, E->scalars + 2 * SUMSQ_SHARDS, cs,
                             nullptr, OPT_PASS_ALL, E->tab_ld, E->lin_ld);
        if (s == "opt_dense") {
            DCTR_TRY(opt_dense_arena(c.optimizer, &E->state->hyper, E->h_state.hyper, E->theta, E->as0, E->as1, E->parts, E->meta,
                                     E->n_blocks, nullptr, 1, E->scalars + 3 * SUMSQ_SHARDS, cs));
            return E->gemm_mode == 1 ? wplanes_written(E, 0, (int)E->params.size() - 1, cs) : DCTR_OK;
        }
        if (s == "mlp0_fwd" || s == "mlp0_dgrad" || s == "mlp0_wgrad") {
            if (E->opnn_fused) { set_error("dctr_time_kernel: the fused Outer-PNN first layer is not one product (time it with rocprofv3)"); return DCTR_ERR_UNSUPPORTED; }
            const Fc& fc = E->mlp[0];
            const GemmOpt go = gemm_opt(E, fc);          // (the mode the steps run in; the planes are current: the caller synchronised)
            if (s == "mlp0_fwd")
                return fc_fwd(E->x_in, E->Din_ld, E->pp(fc.w), E->pp(fc.b), E->h[0], fc.out, B, fc.in, fc.out, 1, fc.keep,
                              &E->state->seed_t, fc.salt, cs, 1, &go);
            if (s == "mlp0_dgrad")
                return fc_bwd_data(E->dh[0], fc.out, E->pp(fc.w), E->dx_in, E->Din_ld, B, fc.in, fc.out, nullptr, 0, 1.f, cs, 1, &go);
            return fc_bwd_weights_partials(E->x_in, E->Din_ld, E->dh[0], fc.out, E->part(fc.w), E->params[fc.w].padded,
                                           E->part(fc.b), E->params[fc.b].padded, B, fc.in, fc.out, fc.splits, cs, 1, &go);
        }
        if (s == "train_step") return record_train(E, B, cs);
        set_error("unknown stage '%s'", kernel);
        return DCTR_ERR_NOT_FOUND;
    };
    hipGraph_t graph = nullptr;
    hipStream_t cs = nullptr;
    DCTR_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    DCTR_HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    int rc = DCTR_OK;
    E->capturing = true;
    for (int i = 0; i < iters && rc == DCTR_OK; ++i) rc = stage(cs);
    E->capturing = false;
    hipError_t e = hipStreamEndCapture(cs, &graph);
    hipStreamDestroy(cs);
    E->lag_suspended = false;
    if (rc != DCTR_OK) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return DCTR_ERR_HIP; }
    hipGraphExec_t exec = nullptr;
    DCTR_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
    hipEvent_t e0, e1;
    DCTR_HIP_CHECK(hipEventCreate(&e0));
    DCTR_HIP_CHECK(hipEventCreate(&e1));
    DCTR_HIP_CHECK(hipGraphLaunch(exec, st));          // warm-up replay
    DCTR_HIP_CHECK(hipStreamSynchronize(st));
    DCTR_HIP_CHECK(hipEventRecord(e0, st));
    DCTR_HIP_CHECK(hipGraphLaunch(exec, st));
    DCTR_HIP_CHECK(hipEventRecord(e1, st));
    DCTR_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    DCTR_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipGraphExecDestroy(exec);
    *h_ms_per_launch = ms / (float)iters;
    if (E->lag_period > 1) {        // whatever the stages did to the tables: every row counts as current from here on
        DCTR_TRY(lag_stamp(E->row_ts, E->rows, E->state, st));
        DCTR_HIP_CHECK(hipStreamSynchronize(st));
        E->lag_dirty = false;
    }
    if (s == "scatter" && !E->group->gemb_clean) {      // the plain scatter wrote compact rows: back to the fused tail's all-zero invariant
        DCTR_HIP_CHECK(hipMemsetAsync(E->group->gemb, 0, (size_t)E->group->max_entries * E->group->K * 4, st));
        DCTR_HIP_CHECK(hipStreamSynchronize(st));
        E->group->gemb_clean = true;
    }
    return DCTR_OK;
}

}  // extern "C"
