/*
 * deepctr_hip.h -- C ABI of libdeepctr_hip.so: the MI355X (gfx950) replacement for what
 * TensorFlow-1.4's CPU op kernels do underneath the reference's tf.estimator
 * `input_fn` / `model_fn` pair (lambdaji/tf_repos, the deep_ctr Model_pipeline scripts).
 *
 * The reference has NO native ABI for this path -- its boundary is Python
 * (`input_fn(filenames, batch_size, num_epochs, perform_shuffle)` DeepFM.py:63 and
 * `model_fn(features, labels, mode, params)` DeepFM.py:100, driven by tf.estimator.Estimator
 * DeepFM.py:341-366).  This ABI is therefore defined by the build and sits under the
 * TF-1.x-compatible Python surface in tf_repos_amd/ (see INTEGRATION.md for the binding).
 * Every entry point cites the reference lines whose TF ops it replaces.
 *
 * Conventions: plain pointers + explicit sizes, no torch types.  Every function returns an
 * int status (0 = DCTR_OK, <0 = error).  `stream` is a hipStream_t passed as void* (NULL =
 * the null stream).  Pointers named d_* are device (HBM) pointers; h_* are host pointers.
 * Step functions never allocate.  Floating point is IEEE binary32 throughout (dtype "f32").
 */
#ifndef DEEPCTR_HIP_H
#define DEEPCTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (mapped by the Python shim to the TF exception classes) ---------- */
#define DCTR_OK                 0
#define DCTR_ERR_INVALID_ARG   -1   /* tf.errors.InvalidArgumentError: bad shape/flag/OOB id   */
#define DCTR_ERR_HIP           -2   /* a HIP runtime call failed (see dctr_last_error)          */
#define DCTR_ERR_NOT_FOUND     -3   /* unknown parameter name                                   */
#define DCTR_ERR_OUT_OF_RANGE  -4   /* tf.errors.OutOfRangeError: end of input                  */
#define DCTR_ERR_PARSE         -5   /* StringToNumberOp / ragged line (InvalidArgumentError)    */
#define DCTR_ERR_UNSUPPORTED   -6

/* ---- enums ------------------------------------------------------------------------ */
enum { DCTR_MODEL_DEEPFM = 0,  /* DeepFM.py   */
       DCTR_MODEL_FNN    = 1,  /* PNN.py --model_type=FNN   */
       DCTR_MODEL_IPNN   = 2,  /* PNN.py --model_type=Inner */
       DCTR_MODEL_OPNN   = 3,  /* PNN.py --model_type=Outer */
       DCTR_MODEL_NFM    = 4,  /* NFM.py      */
       DCTR_MODEL_AFM    = 5,  /* AFM.py      */
       DCTR_MODEL_DCN    = 6,  /* DCN.py      */
       /* canned estimators of wide_n_deep.py:113-151 over its feature columns (:92-105): field_size categorical
        * (identity) columns share one stacked table, dense_size numeric columns enter as dense inputs */
       DCTR_MODEL_WIDE   = 7,  /* --model_type=wide         LinearClassifier            */
       DCTR_MODEL_DEEP   = 8,  /* --model_type=deep         DNNClassifier               */
       DCTR_MODEL_WND    = 9,  /* --model_type=wide_n_deep  DNNLinearCombinedClassifier */
       DCTR_MODEL_MVM    = 10, /* DeepMVM.py: prod_f (e_f + mvm_b_f) beside the MLP, fc([x_mvm || mlp_out]) */
       /* variable-length (multi-hot) models over CSR batches (dctr_train_step_csr): field_size = number of K-wide SLOTS
        * of the MLP input, each the weighted sum of its entries' rows (one-hot fields = one entry of weight 1) */
       DCTR_MODEL_DIN    = 11, /* DIN.py, field-wise sum pooling (--attention_pooling=False, DIN.py:179-183,199-210) */
       DCTR_MODEL_ESMM   = 12  /* DeepCvrMTL.py: shared embeddings, CTR + CVR towers, pCTCVR = pCTR*pCVR (:153-225) */ };

enum { DCTR_OPT_ADAM = 0, DCTR_OPT_ADAGRAD = 1, DCTR_OPT_MOMENTUM = 2, DCTR_OPT_FTRL = 3 }; /* DeepFM.py:204-211 */

/* how the embedding/linear tables are stepped: the reference's loss adds l2_reg*l2_loss(table)
 * (DeepFM.py:189-190) so TF densifies the table gradient and runs the optimizer over all V rows
 * every step.  DENSE_EXACT reproduces that; TOUCHED_ROWS updates only rows present in the batch
 * (a different, "lazy" semantics -- never used for parity claims). */
enum { DCTR_TABLE_DENSE_EXACT = 0, DCTR_TABLE_TOUCHED_ROWS = 1 };

/* reductions fused into the gather (which model consumes the scaled embeddings) */
enum { DCTR_GATHER_RAW = 0,   /* e only                    (PNN.py:133-136, DCN.py:134-138, AFM.py:127-130) */
       DCTR_GATHER_FM  = 1,   /* + y_v = FM second order   (DeepFM.py:133-135)                               */
       DCTR_GATHER_BI  = 2    /* + bi[B,K] bi-interaction  (NFM.py:126-128)                                  */ };

#define DCTR_MAX_LAYERS 8

/* ---- dropout sites (nn.dropout, DeepFM.py:161-162; NFM.py:136-137; AFM.py:152-153,157-158) --------------------------------
 * The engine draws no random stream: the keep/drop decision of element `idx` (row-major index in the logical shape of the
 * dropped tensor) of site `site` in train step `t` (global_step during that step: 1 for the first) is a pure function
 *     u = hash32((seed ^ t * 0xD1B54A32D192ED03) ^ site ^ idx * 0x9E3779B97F4A7C15) >> 8) / 2^24,   kept iff u >= 1 - keep
 * (nn.dropout's floor(keep + u) [TF-1.4]), kept elements scaled by 1/keep.  dctr_dropout_mask evaluates it on the host, so a
 * parity test can hand the engine's own masks to the oracle. */
#define DCTR_DROPOUT_SITE_MLP(i)   (0x1000ull + (uint64_t)(i))   /* output of deep layer i [B, H_i] (after batch_norm when on)      */
#define DCTR_DROPOUT_SITE_MLP2(i)  (0x2000ull + (uint64_t)(i))   /* second tower: ESMM's cvr_ tower [B, H_i]; DIN's att_fc%d [nnz, A_i] */
#define DCTR_DROPOUT_SITE_NFM_BI   0xB1ull                       /* NFM bi-interaction [B, K]            (NFM.py:136-137)           */
#define DCTR_DROPOUT_SITE_AFM_ATT  0xA0ull                       /* AFM attention weights [B, P]         (AFM.py:152-153)           */
#define DCTR_DROPOUT_SITE_AFM_YEMB 0xA1ull                       /* AFM pooled embedding [B, K]          (AFM.py:157-158)           */
#define DCTR_INPUT_SLOTS 8     /* engine-owned input staging sets, see dctr_input_slot */

typedef struct dctr_config {
    int32_t model;                       /* DCTR_MODEL_*                                         */
    int32_t field_size;                  /* --field_size      DeepFM.py:42                        */
    int32_t embedding_size;              /* --embedding_size  DeepFM.py:43 (multiple of 4)        */
    int64_t feature_size;                /* --feature_size    DeepFM.py:41 (rows of the tables)   */
    int32_t n_deep_layers;               /* --deep_layers     DeepFM.py:51                        */
    int32_t deep_layers[DCTR_MAX_LAYERS];
    float   keep_prob[DCTR_MAX_LAYERS];  /* --dropout = TF keep_prob, DeepFM.py:52,162            */
    int32_t cross_layers;                /* --cross_layers    DCN.py:52                           */
    int32_t n_attention_layers;          /* --attention_layers AFM.py:52                          */
    int32_t attention_layers[DCTR_MAX_LAYERS];
    float   l2_reg;                      /* --l2_reg          DeepFM.py:48                        */
    float   learning_rate;               /* --learning_rate   DeepFM.py:47                        */
    int32_t optimizer;                   /* DCTR_OPT_*        DeepFM.py:50,204-211                */
    int32_t table_mode;                  /* DCTR_TABLE_*                                          */
    int32_t batch_norm;                  /* --batch_norm      DeepFM.py:53 (0/1)                  */
    float   batch_norm_decay;            /* --batch_norm_decay DeepFM.py:54                       */
    int32_t max_batch;                   /* largest batch any step/predict call will pass         */
    uint64_t seed;                       /* dropout RNG seed                                      */
    /* row sharding: this handle owns rows {id : id % world == rank}, local row = id / world.
     * world = 1 -> the whole table.  Only the table allocation + table ops look at these. */
    int32_t shard_rank;
    int32_t shard_world;
    int32_t use_graph;                   /* capture the step into a hipGraph (1) or launch eagerly (0) */
    /* canned-estimator models only (DCTR_MODEL_WIDE/DEEP/WND); zero for the model_fn models */
    int32_t dense_size;                  /* numeric_column count (13, wide_n_deep.py:55,94): dense inputs [B,dense_size]  */
    int32_t lin_optimizer;               /* DCTR_OPT_* of the linear side (TF default Ftrl); `optimizer` drives the DNN side */
    float   lin_learning_rate;
    int32_t loss_sum;                    /* 1: loss (and its gradient) is the SUM over the batch [TF-1.4 canned heads], 0: mean */
    /* CSR models only (DCTR_MODEL_DIN/ESMM); zero elsewhere */
    int32_t max_entries;                 /* largest nnz any CSR call will pass (0: max_batch * field_size * 8)             */
    float   ctr_task_wgt;                /* --ctr_task_wgt    DeepCvrMTL.py:47,225                                         */
    /* DCTR_MODEL_DIN with attention pooling (--attention_pooling=True, DIN.py:45,151-177): n_att_pairs > 0 turns it on.  Pair p
     * = (user multi-hot slot, ad slot whose embedding is the attention query), DIN.py:174-177.  The attention MLP has
     * n_attention_layers layers of widths attention_layers[] (the script sizes them with the DEEP widths layers[i], DIN.py:164 --
     * the caller passes what it wants built) and reuses keep_prob[i]; its variables are shared by all pairs (AUTO_REUSE). */
    int32_t n_att_pairs;
    int32_t att_user_slot[8];
    int32_t att_ad_slot[8];
    /* dense_exact tables under Adam: period N of the TIME-BLOCKED table sweep (csrc/lag.h).  TF's Adam steps every table row every
     * step because l2_loss(table) is in the loss (DeepFM.py:188-190); a row no batch touches follows a recurrence that needs
     * nothing the step computes, so it may lag: each step the background sweep advances 1/N of the table, and a row is advanced
     * through the steps it missed -- the same update calls with the same per-step lr_t, in order -- by whoever reads it next
     * (the gather, the touched-rows step, a flush before predict / eval / parameter reads / a loss-reporting step).  Results are
     * those of the classic sweep, row for row; the table step's HBM traffic drops N-fold.  0 = the library's default
     * (DCTR_SWEEP_PERIOD, else 8), 1 = the classic sweep of every row every step; at most 24. */
    int32_t table_sweep_period;
    /* contrib.layers.batch_norm on the rank-2 layer outputs takes TF-1.4's FUSED path (layers.py: fused defaults to True, rank 2 is
     * fusable; the scripts pass updates_collections=None), whose kernel normalises with the biased batch variance but hands the
     * moving average the Bessel-corrected one, var * B / (B - 1) (fused_batch_norm_op.cc `rest_size_adjust`).  0 = that (default);
     * 1 = the biased variance in the moving average too (the non-fused nn.moments path). */
    int32_t batch_norm_biased_moving_variance;
    /* Arithmetic of the MLP's three matrix products (contrib.layers.fully_connected and its MatMul gradients, DeepFM.py:156-158,
     * 165-166,213).  1 = split precision: every f32 operand element is carried as THREE bf16 planes (x = h + m + l exactly: 3 x 8 =
     * 24 significand bits) and the six leading plane products (hh, hm, mh, hl, lh, mm) are accumulated in f32 on the bf16 matrix
     * pipe, which runs 16x the f32 MFMA's rate; the three dropped products are below 2^-24 of the element product, so the result is
     * an f32 dot product to within the rounding of its f32 accumulation (measured against an fp64 product: at or below the exact
     * kernel's error, tests/test_gemm_split_gpu.py).  2 = exact f32 MFMA (bitwise an fmaf chain).  0 = the library's default: what
     * the environment variable DCTR_GEMM_MODE=split|exact names, else split (round 6: the mode bench.py times is the mode a
     * zero-initialised dctr_config gets, and the mode the GPU test suite runs in).  Shapes the split kernels do not take (small
     * batches, widths not a multiple of 8) run the exact kernels in either mode. */
    int32_t gemm_mode;
} dctr_config;

typedef struct dctr_engine* dctr_handle;

/* ---- library ------------------------------------------------------------------------- */
int         dctr_version(void);
const char* dctr_last_error(void);                       /* thread-local message for the last error */
int         dctr_device_count(int* n);
int         dctr_set_device(int dev);

/* raw device memory helpers for hosts without torch */
int dctr_malloc(void** d_ptr, size_t nbytes);
int dctr_free(void* d_ptr);
int dctr_memcpy_h2d(void* d_dst, const void* h_src, size_t nbytes, void* stream);
int dctr_memcpy_d2h(void* h_dst, const void* d_src, size_t nbytes, void* stream);
int dctr_memset(void* d_dst, int value, size_t nbytes, void* stream);
int dctr_stream_sync(void* stream);

/* ---- K1: libsvm text -> tensors.  Replaces decode_libsvm (DeepFM.py:65-81): string_split(' '),
 * string_to_number(label,f32), string_split(':'), string_to_number(ids,i32 / vals,f32).  Host code
 * (re-entrant).  Parses up to max_rows lines from h_text[0:nbytes); every line must carry exactly
 * field_size id:val tokens (DeepFM.py:92,120-122).  *n_rows = lines parsed, *n_consumed = bytes
 * consumed (whole lines only).  Errors: DCTR_ERR_PARSE (message names line/token). */
int dctr_parse_libsvm(const char* h_text, size_t nbytes, int field_size, int64_t max_rows,
                      int32_t* h_ids, float* h_vals, float* h_labels,
                      int64_t* n_rows, size_t* n_consumed);

/* the same decode of a whole buffer with `threads` workers inside the library (map(decode_libsvm, num_parallel_calls=10),
 * DeepFM.py:84).  h_ids == NULL: count only (*n_rows = rows the buffer holds).  Otherwise the arrays must hold capacity_rows
 * >= that count.  Errors are the serial parser's (same message and line number). */
int dctr_parse_libsvm_mt(const char* h_text, size_t nbytes, int field_size, int threads, int32_t* h_ids, float* h_vals,
                         float* h_labels, int64_t capacity_rows, int64_t* n_rows);

/* CSV text -> column tensors.  Replaces tf.decode_csv(line, record_defaults) (wide_n_deep.py:67-73; defaults :59-64):
 * kinds[c] = 0 float / 1 int32 column; an empty field takes its column's default; float (int) columns land in order of
 * appearance in h_f [rows, n_float] (h_i [rows, n_int]).  Same whole-lines / n_consumed contract as dctr_parse_libsvm. */
int dctr_parse_csv(const char* h_text, size_t nbytes, int n_cols, const int8_t* kinds, const float* f_defaults,
                   const int32_t* i_defaults, int64_t max_rows, float* h_f, int32_t* h_i, int64_t* n_rows,
                   size_t* n_consumed);

/* ---- K2: embedding gather + value scale + fused reductions.
 * Replaces embedding_lookup(FM_W)/multiply/reduce_sum (DeepFM.py:126-127), embedding_lookup(FM_V),
 * multiply (DeepFM.py:130-132), and per `mode` the FM second-order term (DeepFM.py:133-135) or NFM's
 * bi-interaction (NFM.py:126-128).
 *   d_emb [rows,K] f32, d_lin [rows] f32 or NULL (DCN has none), d_ids [B,F] i32, d_vals [B,F] f32
 *   d_e   [B, e_ld] f32 out: e[b, f*K+k] = emb[id,k]*val   (e_ld >= F*K, row stride in floats)
 *   d_yw  [B] out (NULL if d_lin NULL): sum_f lin[id]*val
 *   d_sum [B,K] out or NULL: S[b,k] = sum_f e[b,f,k]       (kept for the backward)
 *   d_red      out: mode FM -> y_v [B]; mode BI -> bi [B,K]; RAW -> ignored
 *   d_status   int32[2] device word: [0] != 0 if any id was outside [0, feature_size) -- the id that
 *              failed is stored in [1]; rows for such ids read as 0.  TF's CPU gather raises
 *              InvalidArgumentError; the host checks the word with dctr_check_ids.
 * ids are GLOBAL ids; with shard_world>1 the caller passes local rows instead (see dctr_table_*). */
int dctr_embed_gather_fwd(const float* d_emb, const float* d_lin, int64_t rows,
                          const int32_t* d_ids, const float* d_vals,
                          int B, int F, int K, int mode,
                          float* d_e, int e_ld, float* d_yw, float* d_sum, float* d_red,
                          int32_t* d_status, void* stream);
/* the same gather from a table held as RECORDS: row r of the embedding table starts at d_emb + r * emb_ld floats, its linear weight
 * sits at d_lin + r * lin_ld (e.g. [row | weight | pad] records of emb_ld = lin_ld = K + 4 or 32 floats with d_lin = d_emb + K: the
 * row-sharded exchange's packed rows, and the layouts compared in profiles/r03_gather_layouts.txt).  emb_ld % 4 == 0, emb_ld >= K. */
int dctr_embed_gather_strided(const float* d_emb, int emb_ld, const float* d_lin, int lin_ld, int64_t rows,
                              const int32_t* d_ids, const float* d_vals, int B, int F, int K, int mode, float* d_e, int e_ld,
                              float* d_yw, float* d_sum, float* d_red, int32_t* d_status, void* stream);

/* dctr_parse_csv over a whole buffer with `threads` workers inside the library (h_f == h_i == NULL: count the records only) */
int dctr_parse_csv_mt(const char* h_text, size_t nbytes, int n_cols, const int8_t* kinds, const float* f_defaults,
                      const int32_t* i_defaults, int threads, float* h_f, int32_t* h_i, int64_t capacity_rows, int64_t* n_rows);

/* ---- TFRecord input of the DIN / ESMM scripts (DIN.py:57-97, DeepCvrMTL.py:61-104): tf.data.TFRecordDataset +
 * tf.parse_single_example over files written by Feature_pipeline/get_tfrecord.py:44-98.  Host code, re-entrant.
 * dctr_tfrecord_scan: record framing.  rec_off[i] / rec_len[i] = payload of record i inside h_buf (either may be NULL to
 * count only); stops at a truncated tail (*n_consumed = bytes of whole records); verify_crc checks both masked crc32c
 * fields (DCTR_ERR_PARSE on mismatch = tf.errors.DataLossError).  dctr_tfrecord_frame writes one framed record
 * (nbytes + 16 bytes) -- used to produce fixtures and caches. */
int dctr_tfrecord_scan(const uint8_t* h_buf, size_t nbytes, int64_t max_records, int verify_crc, int64_t* rec_off,
                       int64_t* rec_len, int64_t* n_records, size_t* n_consumed);
int dctr_tfrecord_frame(const uint8_t* h_payload, size_t nbytes, uint8_t* h_out);
/* crc32c (Castagnoli) of a host buffer; masked = TF's ((crc >> 15 | crc << 17) + 0xa282ead8), the checksum of TFRecord framing
 * and of the tensors / table blocks of a TF checkpoint bundle (the Saver behind Estimator's model_dir, DeepFM.py:288,341). */
int dctr_crc32c(const uint8_t* h_buf, size_t nbytes, int masked, uint32_t* h_crc);
/* one entry of the slot layout = one feature of the script's parse spec and where it lands in the MLP input:
 *   fixed_len  > 0: FixedLenFeature([n], int64)  -> n slots of one entry each, weight 1   ("feat_ids", DIN.py:63)
 *   fixed_len == 0: FixedLenFeature([], int64)   -> one slot, one entry                   ("a_catids",  DIN.py:73)
 *   fixed_len  < 0: VarLenFeature(int64) [+ VarLenFeature(float32) weights]              ("u_catids"/"u_catvals", DIN.py:65-66) */
typedef struct dctr_slot_spec {
    const char* ids_feature;
    const char* vals_feature;            /* NULL: all weights 1 (a_intids, DIN.py:76,148) */
    int32_t     fixed_len;
} dctr_slot_spec;
/* tf.parse_single_example of n_records payloads straight into the slot-ordered CSR that dctr_train_step_csr consumes:
 * h_offsets [n_records*S + 1] (S = total slots of the spec list, in list order), h_ids / h_weights [*n_entries],
 * h_labels [n_labels, n_records] from FixedLenFeature([], float32) features (y, z).  Call once with h_ids == NULL to
 * count *n_entries, then with buffers of cap_entries.  ids outside [0, feature_size) -> DCTR_ERR_INVALID_ARG (the
 * gather's InvalidArgumentError, raised at parse time); a missing / wrong-length FixedLenFeature -> DCTR_ERR_PARSE. */
int dctr_examples_to_slot_csr(const uint8_t* h_buf, const int64_t* rec_off, const int64_t* rec_len, int64_t n_records,
                              const dctr_slot_spec* slots, int n_specs, const char* const* label_names, int n_labels,
                              int64_t feature_size, int64_t cap_entries, int32_t* h_offsets, int32_t* h_ids,
                              float* h_weights, float* h_labels, int64_t* n_entries);

/* ---- K8a: group the batch's ids (the IndexedSlices -> unsorted_segment_sum bookkeeping,
 * SURVEY Appendix B item 2).  n = B*F entries, traversed field-major.  Workspace is owned by a
 * dctr_group object so that step calls never allocate. */
typedef struct dctr_group* dctr_group_t;
int dctr_group_create(int64_t rows, int64_t max_entries, int K, dctr_group_t* g);
int dctr_group_destroy(dctr_group_t g);
/* after this call (all on `stream`): uniq[0:U) = distinct ids, seg_start[0:U], perm[0:n) entry
 * indices (entry = f*B+b) grouped by unique id, slot[id] = u+1 for touched ids (0 elsewhere). */
int dctr_group_ids(dctr_group_t g, const int32_t* d_ids, int B, int F, void* stream);
int dctr_group_num_unique(dctr_group_t g, int32_t* h_U, void* stream);   /* syncs */
/* any out pointer may be NULL.  d_counters: int32[2] = {U, total grouped entries}; d_gemb [cap,K] / d_glin [cap] are
 * the compact gradient rows (row u <-> uniq[u]) zeroed by dctr_group_ids and filled by dctr_embed_scatter_bwd. */
int dctr_group_buffers(dctr_group_t g, const int32_t** d_uniq, const int32_t** d_seg_start, const int32_t** d_cnt,
                       const int32_t** d_perm, const int32_t** d_slot, const int32_t** d_counters,
                       float** d_gemb, float** d_glin);

/* ---- K8b: sparse dW_emb / dW_lin: per-unique-row sums of the batch's row gradients.
 * Replaces the gradient of the two gathers (DeepFM.py:126,130) densified by UnsortedSegmentSum.
 *   d_dE [B, de_ld]: gradient w.r.t. the scaled embeddings e (from the model's backward)
 *   FM term fused (mode FM/BI): dE += coef[b,k]-style terms, see DESIGN.md "scatter"
 *   d_dy [B]: dLoss/dlogit (for the linear table); NULL if no linear table
 *   outputs (compact, row u <-> uniq[u]):  d_gemb [U,K], d_glin [U]  (zeroed by dctr_group_ids) */
int dctr_embed_scatter_bwd(dctr_group_t g, const float* d_dE, int de_ld,
                           const float* d_e, int e_ld, const float* d_sum, const float* d_coef,
                           const float* d_dy, const float* d_vals,
                           int B, int F, int K, int mode,
                           float* d_gemb, float* d_glin, void* stream);

/* ---- K8b + K9 in ONE launch (the step's tail): the segment sum of every distinct id goes straight into that row's optimizer
 * step -- UnsortedSegmentSum (the gradient of DeepFM.py:126,130's gathers) followed by the sparse half of
 * optimizer.minimize (DeepFM.py:204-213) -- without the compact-gradient round trip of dctr_embed_scatter_bwd + dctr_opt_table.
 * Visits ONLY the batch's distinct rows (grad = segment sum + l2*theta; the dense-exact step of the untouched rows is
 * dctr_opt_table's / the engine's background sweep).  `hyper` as for dctr_opt_table; d_sumsq as there (pre-update sums over
 * the visited rows) or NULL.  Call dctr_group_ids(g, ids, B, F) first; leaves g's slot words and compact rows zeroed.
 * Segments of >= 256 entries (Criteo's 13 numeric ids: every example) are reduced by several blocks whose partial sums meet
 * through returned float atomics and a completion ticket -- tests/test_scatter_stress_gpu.py hammers exactly that path. */
int dctr_embed_scatter_apply(dctr_group_t g, int kind, const float* hyper,
                             float* d_emb, float* d_emb_s0, float* d_emb_s1,
                             float* d_lin, float* d_lin_s0, float* d_lin_s1, float l2, float* d_sumsq,
                             const float* d_dE, int de_ld, const float* d_e, int e_ld, const float* d_sum, const float* d_coef,
                             const float* d_dy, const float* d_vals, int B, int F, int K, int mode, void* stream);

/* ---- K2/K8 over CSR batches: tf.nn.embedding_lookup_sparse(params, sp_ids, sp_weights, combiner="sum") of the DIN / ESMM
 * scripts (DIN.py:148,180-183; DeepCvrMTL.py:155-159).  d_offsets [B+1] row pointers into d_ids / d_weights [nnz]
 * (d_weights NULL = all ones, DIN.py:148).
 *   fwd: d_out[b, 0:K] = sum_j weights[j] * emb[ids[j], :]  (row stride out_ld floats; an empty row gives zeros)
 *   bwd: groups the nnz ids in `g` (created with max_entries >= nnz) and leaves the per-distinct-id gradient rows in its
 *        compact buffers (dctr_group_buffers: uniq / gemb), ready for dctr_opt_table; d_entry_row [nnz] is scratch */
int dctr_embed_lookup_sparse_fwd(const float* d_emb, int64_t rows, int K, const int32_t* d_offsets, const int32_t* d_ids,
                                 const float* d_weights, int B, float* d_out, int out_ld, int32_t* d_status, void* stream);
int dctr_embed_lookup_sparse_bwd(dctr_group_t g, const float* d_dout, int dout_ld, const int32_t* d_offsets, const int32_t* d_ids,
                                 const float* d_weights, int B, int nnz, int K, int32_t* d_entry_row, void* stream);

/* ---- K9: optimizers (DeepFM.py:204-213), TF-1.4 update rules.
 * `hyper`: Adam {lr, beta1, beta2, eps, t}; Adagrad {lr}; Momentum {lr, momentum}; Ftrl {lr}.
 * slot0/slot1: Adam m/v, Adagrad accum/-, Momentum accum/-, Ftrl accum/linear.
 * grad = sum_{s<n_partials} d_grad[s*partial_stride + i]  (+ l2*theta if l2 != 0). */
int dctr_opt_dense(int kind, const float* hyper, float* d_theta, float* d_slot0, float* d_slot1,
                   const float* d_grad, int n_partials, int64_t partial_stride, int64_t n, float l2,
                   void* stream);
/* tables: dense-exact streams all `rows` rows: grad = l2*theta + (slot[r] ? compact_grad[slot[r]-1] : 0)
 * and clears slot[r]; touched-rows visits only uniq[0:U).  emb [rows,K] and lin [rows] (lin may be NULL). */
int dctr_opt_table(int kind, const float* hyper, int table_mode, int64_t rows, int K,
                   float* d_emb, float* d_emb_s0, float* d_emb_s1,
                   float* d_lin, float* d_lin_s0, float* d_lin_s1,
                   dctr_group_t g, float l2, float* d_sumsq /* [128] or NULL: 64 shards += sum emb^2, next 64 shards += sum lin^2 (pre-update) */,
                   void* stream);

/* ---- K6: dense layers on the matrix cores (fp32-input MFMA, exact f32).
 * Replaces contrib.layers.fully_connected (DeepFM.py:156-158,165-166): y = act(x W + b), W [in,out].
 * fwd: Y[M,N] = epilogue(X[M,K] W[K,N] + b);  relu!=0 -> max(.,0);  keep<1 -> dropout with a
 *      counter-based RNG keyed by (seed, element index) -- `x*floor(keep+U)/keep` [TF-1.4].
 * bwd_data:  dX[M,K] = dY[M,N] W^T, then if d_act != NULL: dX *= (act>0)/keep_prev  (ReLU+dropout of the
 *            producing layer; act is that layer's stored output)
 * bwd_weights: dW[K,N] = X^T dY, db[N] = colsum(dY) */
/* host logic only (no GPU needed): which kernel a layer product of this shape takes -- op 'f' forward Y[M,N] = X[M,K] W[K,N],
 * 'd' dX[M,K] = dY[M,N] W^T, 'w' dW[K,N] = X^T dY over M rows.  Writes "ws" (weights-stationary, tall operands), "dr TMxTN", "dr TMxTN xS" (S batch splits) or "lds[ xS]". */
int dctr_gemm_plan(char op, int M, int K, int N, char* out, int out_len);
int dctr_fc_fwd(const float* d_x, int ldx, const float* d_w, const float* d_b, float* d_y, int ldy,
                int M, int K, int N, int relu, float keep, uint64_t seed, void* stream);
int dctr_fc_bwd_data(const float* d_dy, int lddy, const float* d_w, float* d_dx, int lddx,
                     int M, int K, int N, const float* d_act, int ldact, float keep_prev, void* stream);
int dctr_fc_bwd_weights(const float* d_x, int ldx, const float* d_dy, int lddy, float* d_dw, float* d_db,
                        int M, int K, int N, float* d_workspace, size_t workspace_bytes, void* stream);
/* The same three products in SPLIT PRECISION (dctr_config.gemm_mode = 1; csrc/gemm_dr3.hip): every f32 operand element as three bf16
 * planes, six plane products, f32 accumulation -- f32-equivalent results on the bf16 matrix pipe.  The weight is pre-split once
 * (dctr_gemm_wsplit, after every change of W) into the two forms the forward and the dgrad product read; their sizes come from
 * dctr_gemm_split_plane_bytes.  Same arguments and epilogues as the exact ops above with the plane buffer in W's place;
 * DCTR_ERR_UNSUPPORTED when no split kernel takes the shape (M >= 1024; K and N multiples of 8 in [64, 16384]; at least 0.8 GFLOP; one
 * round of the chip): the caller then uses the exact op.  dctr_fc_bwd_weights_split needs a workspace (as dctr_fc_bwd_weights with splits).
 * dctr_gemm_split_launches: split-kernel launches so far in this process (tests: the mode is really on). */
int dctr_gemm_split_plane_bytes(int K, int N, int64_t* fwd_bytes, int64_t* dgr_bytes);
int dctr_gemm_wsplit(const float* d_w, int K, int N, void* d_fwd_planes, void* d_dgr_planes, void* stream);
int dctr_fc_fwd_split(const float* d_x, int ldx, const void* d_fwd_planes, const float* d_b, float* d_y, int ldy,
                      int M, int K, int N, int relu, float keep, uint64_t seed, void* stream);
int dctr_fc_bwd_data_split(const float* d_dy, int lddy, const void* d_dgr_planes, float* d_dx, int lddx,
                           int M, int K, int N, const float* d_act, int ldact, float keep_prev, void* stream);
int dctr_fc_bwd_weights_split(const float* d_x, int ldx, const float* d_dy, int lddy, float* d_dw, float* d_db,
                              int M, int K, int N, float* d_workspace, size_t workspace_bytes, void* stream);
int64_t dctr_gemm_split_launches(void);
/* The products of a layer over a TALL operand (M rows in the millions, K and N of 128 or 256: AFM's attention layer over the B * P
 * pair rows, AFM.py:142-147) in split precision (csrc/gemm_ts.h) -- what dctr_create's AFM handle runs under gemm_mode = 1, as ops:
 *   dctr_fc_fwd_dot_split       Y[M,N] = relu(X[M,K] W[K,N] + b) and, from the same accumulators, dot_out[row] = <Y[row,:], dot_w>
 *                               (the score of the (N -> 1) layer that follows, AFM.py:147; d_dot_out may be null);
 *   dctr_fc_bwd_data_gate_split dX[M,K] = (rowscale (x) kscale . 1[H > 0]) W[K,N]^T for the layer's stored output H [M,N]: the input
 *                               gradient when the output gradient is rank one under the ReLU mask (d H = d score (x) w_out . 1[H > 0]).
 *   Given d_sign_bits, d_w [K,N] and d_b [N], dctr_pairs_fc_bwd_weights_gate_split does not read H either (d_h may be null): the gate comes
 *                               from the sign words and dwo from the product itself, dwo[n] = sum_k W[k,n] dWraw[k,n] + b[n] dbraw[n] -- then Y
 *                               of the forward has no reader left and dctr_pairs_fc_fwd_dot_split / dctr_fc_fwd_dot_split accept d_y = null.
 *   d_sign_bits (forward and input gradient, may be null): 32 M bytes the forward fills with the SIGN bits of Y (one 64-bit word per row and quarter of its columns);
 *                               the gradient given them reads 32 bytes per row instead of the row of H (d_h may then be null) -- same result.
 *   dctr_fc_bwd_weights_gate_split  the same layer's weight gradient under that rank-one output gradient, dW[K,N] = X^T (rowscale (x) colscale . 1[H > 0]),
 *                               with db[N] (its bias gradient) and dwo[N] = sum_r rowscale[r] H[r,:] (the (N -> 1) layer's weight gradient);
 *                               workspace: at least (K N + 2 N) floats, 256 x that for full speed (partial slabs over row ranges).
 *   dctr_pairs_fc_fwd_dot_split / dctr_pairs_fc_bwd_weights_gate_split  the first and the third with X never stored: row b P + p of X is the
 *                               element-wise product e[b, pair_i[p], :] . e[b, pair_j[p], :] of two gathered embeddings (AFM.py:130-139; d_e
 *                               [examples, e_ld] floats, field f at f * K; d_pair_i / d_pair_j: P int16 field indices), formed in the registers --
 *                               bit-identical to the same op over the materialised products.
 * d_planes_ws: dctr_ts_plane_bytes(K, N) bytes of device memory the call overwrites (the weight's bf16 planes).  DCTR_ERR_UNSUPPORTED when
 * the shape or the alignment (16-byte pointers, leading dimensions multiples of 4) is not taken: the caller uses the exact ops. */
int dctr_ts_plane_bytes(int R, int N, int64_t* bytes);
int dctr_fc_fwd_dot_split(const float* d_x, int ldx, const float* d_w, const float* d_b, float* d_y, int ldy, int64_t M, int K, int N,
                          const float* d_dot_w, float* d_dot_out, void* d_sign_bits, void* d_planes_ws, void* stream);
int dctr_fc_bwd_data_gate_split(const float* d_h, int ldh, const void* d_sign_bits, const float* d_rowscale, const float* d_kscale, const float* d_w,
                                float* d_dx, int lddx, int64_t M, int K, int N, void* d_planes_ws, void* stream);
int dctr_fc_bwd_weights_gate_split(const float* d_x, int ldx, const float* d_h, int ldh, const float* d_rowscale, const float* d_colscale,
                                   float* d_dw, float* d_db, float* d_dwo, int64_t M, int K, int N, float* d_workspace, size_t workspace_bytes,
                                   void* stream);
int dctr_pairs_fc_fwd_dot_split(const float* d_e, int e_ld, int examples, const int16_t* d_pair_i, const int16_t* d_pair_j, int P, const float* d_w,
                                const float* d_b, float* d_y, int ldy, int64_t M, int K, int N, const float* d_dot_w, float* d_dot_out,
                                void* d_sign_bits, void* d_planes_ws, void* stream);
int dctr_pairs_fc_bwd_weights_gate_split(const float* d_e, int e_ld, int examples, const int16_t* d_pair_i, const int16_t* d_pair_j, int P,
                                         const float* d_h, int ldh, const void* d_sign_bits, const float* d_w, const float* d_b,
                                         const float* d_rowscale, const float* d_colscale, float* d_dw, float* d_db, float* d_dwo, int64_t M, int K,
                                         int N, float* d_workspace, size_t workspace_bytes, void* stream);

/* ---- K3/K5/K4: interaction layers ------------------------------------------------------ */
/* AFM's attention-weighted pairwise interaction (AFM.py:127-158) as an op.  It needs the attention network's variables and ~B P (K + A)
 * floats of workspace, so it runs ON AN AFM HANDLE (dctr_create with model afm: attention_layers, keep_prob[0] = attention dropout,
 * keep_prob[1] = pooled-embedding dropout; the variables are set with dctr_param_set under the handle's names (dctr_param_info):
 * "att_mlp0/weights" [K,A], "att_mlp0/biases" [A] (AFM.py:145, TF scope mlp0), "attention_out/weights" [A,1], "attention_out/biases" [1]
 * (AFM.py:147)).
 *   fwd: d_e [B, e_ld] the value-scaled embeddings e[b,f,:] (AFM.py:129-130) -> d_y_emb [B, y_ld]: sum_p a'[b,p] e_i (.) e_j after both
 *        dropouts when train != 0 (masks of the handle's seed / global_step: include "dropout sites"); d_att [B, P] (optional): the
 *        softmax weights (AFM.py:151, before their dropout), pairs in the reference's (i < j) double-loop order (AFM.py:134-136).
 *   bwd: d_dy_emb [B, dy_ld] = dL/d y_emb -> d_dE [B, de_ld] = dL/de through pair products, attention network and softmax; the
 *        attention variables' gradients are read with dctr_param_grad_get.  Must follow dctr_afm_fwd of the same batch. */
int dctr_afm_fwd(dctr_handle h, const float* d_e, int e_ld, int B, int train, float* d_y_emb, int y_ld, float* d_att, void* stream);
int dctr_afm_bwd(dctr_handle h, const float* d_dy_emb, int dy_ld, int B, float* d_dE, int de_ld, void* stream);
/* PNN inner product (PNN.py:141-152): ip[b,p] = <e[b,i_p,:], e[b,j_p,:]>, pairs lexicographic i<j. */
int dctr_pnn_inner_fwd(const float* d_e, int e_ld, int B, int F, int K, float* d_ip, int ip_ld, void* stream);
/* dE[b,i,:] += sum_j dip[b,pair(i,j)] e[b,j,:]   (accumulates into d_dE) */
int dctr_pnn_inner_bwd(const float* d_e, int e_ld, const float* d_dip, int dip_ld, int B, int F, int K,
                       float* d_dE, int de_ld, void* stream);
/* PNN outer product (PNN.py:154-167): op[b,p,a,c] = e[b,i_p,a] e[b,j_p,c], materialised [B, P*K*K]. */
int dctr_pnn_outer_fwd(const float* d_e, int e_ld, int B, int F, int K, float* d_op, int64_t op_ld, void* stream);
int dctr_pnn_outer_bwd(const float* d_e, int e_ld, const float* d_dop, int64_t dop_ld, int B, int F, int K,
                       float* d_dE, int de_ld, void* stream);
/* Outer-PNN WITHOUT the product tensor: the first fully_connected over [flat embeddings | outer products] (PNN.py:154-167 feeding
 * PNN.py:159-166) with the products e[b,i_p,a] e[b,j_p,c] formed inside the GEMM fragments.  d_w is the layer's weight exactly as
 * the reference declares it, [F K + P K K, H] row-major (rows F K + (p K + a) K + c meet pair p's products).  K a power of two
 * >= 16, H a multiple of 4; bwd_data additionally K <= 64 and H <= 256.
 *   fwd:          y[B,H] = act([e | outer(e)] W + b)   (relu / dropout as dctr_fc_fwd; workspace: dctr_pnn_outer_fc_workspace_bytes)
 *   bwd_weights:  dW[F K + P K K, H], db[H] (overwritten)
 *   bwd_data:     dE[B, F K] = dL/de through both row groups of W (overwritten) */
size_t dctr_pnn_outer_fc_workspace_bytes(int max_batch, int H);
int dctr_pnn_outer_fc_fwd(const float* d_e, int e_ld, int B, int F, int K, const float* d_w, const float* d_b, float* d_y, int ldy,
                          int H, int relu, float keep, uint64_t seed, float* d_workspace, size_t workspace_bytes, void* stream);
int dctr_pnn_outer_fc_bwd_weights(const float* d_e, int e_ld, int B, int F, int K, const float* d_dy, int lddy, int H, float* d_dw,
                                  float* d_db, void* stream);
int dctr_pnn_outer_fc_bwd_data(const float* d_e, int e_ld, int B, int F, int K, const float* d_dy, int lddy, int H, const float* d_w,
                               float* d_dE, int de_ld, void* stream);
/* DCN cross network (DCN.py:140-145): x_{l+1} = x0*(x_l . w_l) + x_l + b_l, w,b [L,D].
 * d_xs [L+1, B, D] keeps every x_l (x_0 copied in) and d_xlw [L,B] every x_l.w_l for the backward. */
int dctr_dcn_cross_fwd(const float* d_x0, int x0_ld, const float* d_w, const float* d_b, int B, int D, int L,
                       float* d_xs, float* d_xlw, void* stream);
/* in: d_dxL [B,D] gradient w.r.t. x_L (row stride dxl_ld).  out: d_dx0 [B,D] (accumulated into, row stride
 * dx0_ld), d_dw/d_db [L,D] (overwritten). */
int dctr_dcn_cross_bwd(const float* d_xs, const float* d_xlw, const float* d_w, const float* d_dxL, int dxl_ld,
                       int B, int D, int L, float* d_dx0, int dx0_ld, float* d_dw, float* d_db,
                       float* d_workspace, size_t workspace_bytes, void* stream);

/* ---- K7: loss head.  y = bias + y_w + y_v + y_d (DeepFM.py:174-175, terms may be NULL);
 * prob = sigmoid(y) (DeepFM.py:176); loss = mean xent (DeepFM.py:188, [TF-1.4] stable form);
 * d_dy = (prob - label)*inv_batch (inv_batch = 1/global batch).  d_loss_sum accumulates sum_b xent (caller zeroes, divides by B). */
int dctr_loss_head(const float* d_bias, const float* d_yw, const float* d_yv, const float* d_yd,
                   const float* d_labels, int B, float inv_batch, float* d_y, float* d_prob, float* d_dy,
                   float* d_loss_sum, void* stream);

/* ---- K10: tf.metrics.auc (DeepFM.py:194), 200 thresholds [TF-1.4].  d_counts int64[4*200]
 * = tp,fn,tn,fp per threshold, accumulated across calls; result computed on the host. */
int dctr_auc_update(const float* d_labels, const float* d_prob, int B, int64_t* d_counts, void* stream);
int dctr_auc_result(const int64_t* d_counts, float* h_auc, void* stream);

/* ---- engine: owns parameters, optimizer slots, activations; one handle per GPU rank ------- */
int dctr_create(const dctr_config* cfg, dctr_handle* h);
int dctr_destroy(dctr_handle h);
/* parameters by engine name ("emb", "linear", "bias", "mlp0/weights", ... -- tf_repos_amd.checkpoint
 * maps them to the TF variable names fm_v/fm_w/fm_bias/... of SURVEY Appendix A) */
int dctr_param_count(dctr_handle h, int* n);
int dctr_param_info(dctr_handle h, int index, const char** name, int* rank, int64_t dims[4]);
int dctr_param_set(dctr_handle h, const char* name, const float* h_src, size_t nbytes);
int dctr_param_get(dctr_handle h, const char* name, float* h_dst, size_t nbytes);
/* optimizer slots: which = 0/1 (see dctr_opt_dense) */
/* gradient of a dense (non-table) variable as of the last backward pass: its partial slabs summed (what the optimizer step consumes),
 * logical shape, host destination.  For gradient-level parity (tests/test_afm_grad_gpu.py) and the op-level interaction entries. */
int dctr_param_grad_get(dctr_handle h, const char* name, float* h_dst, size_t nbytes);
int dctr_slot_get(dctr_handle h, const char* name, int which, float* h_dst, size_t nbytes);
int dctr_slot_set(dctr_handle h, const char* name, int which, const float* h_src, size_t nbytes);
/* raw device pointer of a variable (device-side initialisation of tables too large to stage through the host).  For a TABLE of a
 * handle whose rows may lag (table_sweep_period > 1, csrc/lag.h) the call first brings every row to global_step, and the pointer is a
 * view of the table AS OF THIS CALL: it is valid until the next train step (rows then lag again behind what the pointer shows).  After
 * WRITING through it, call dctr_set_global_step(h, current step) before training on -- that re-stamps every row as current, so the
 * written values are not replayed through steps they were never part of.  (dctr_param_set / dctr_slot_set do all of this themselves.) */
int dctr_param_device_ptr(dctr_handle h, const char* name, float** d_ptr);
/* the same with the distance (floats) between consecutive rows of the variable's first dimension.  With DCTR_TABLE_RECORDS=1 in the
 * environment when it is created, a handle whose table rows may lag keeps each row of `emb` / `linear` together with its Adam slots
 * in one RECORD (csrc/engine.h tab_ld: a lagging row's catch-up and its touched-rows step then read one or two adjacent cache lines
 * instead of six scattered ones; opt-in -- measured, the step as a whole does not gain, profiles/r04_table_records.txt): such a
 * table is a strided view (*row_stride > the row's own width), which dctr_param_device_ptr refuses. */
int dctr_param_device_view(dctr_handle h, const char* name, float** d_ptr, int64_t* row_stride);
int dctr_set_global_step(dctr_handle h, int64_t step);
int dctr_get_global_step(dctr_handle h, int64_t* step);

/* one optimizer step on one batch: forward, loss, backward, optimizer (DeepFM.py:125-213).
 * d_ids/d_vals/d_labels are device pointers ([B,F] i32, [B,F] f32, [B] f32).  h_loss (may be NULL)
 * receives the full loss of DeepFM.py:188-190 evaluated BEFORE the update; passing it forces a sync. */
int dctr_train_step(dctr_handle h, const int32_t* d_ids, const float* d_vals, const float* d_labels,
                    int B, float* h_loss, void* stream);
/* Engine-owned input staging: DCTR_INPUT_SLOTS sets of ([max_batch,F] i32, [max_batch,F] f32, [max_batch] f32) device
 * buffers.  A caller that writes a batch straight into slot k (e.g. the H2D copy of the input pipeline) and passes those
 * same pointers to dctr_train_step / dctr_predict / dctr_eval_batch pays no staging copy; any other pointers are copied
 * device-to-device into slot 0 first. */
int dctr_input_slot(dctr_handle h, int slot, int32_t** d_ids, float** d_vals, float** d_labels);
/* The input pipeline announces the NEXT training batch: its ids -- already written into one of the input slots, and left
 * untouched until that dctr_train_step call -- are grouped (the de-duplication of the IndexedSlices gradient of
 * embedding_lookup, DeepFM.py:126,130 / 213) during the tail of the step in flight instead of beside the next step's first MLP
 * layer.  Purely a scheduling hint: results are identical with or without it; ids outside the input slots, CSR / canned /
 * row-sharded handles are ignored.
 * The hint is tied to the slot's GENERATION: dctr_input_slot_rewrite(slot) -- which an input pipeline calls before it refills a
 * slot -- and the engine's own staging copies advance it, and a step whose slot has moved on since the hint groups the ids
 * itself instead of using a stale grouping.  dctr_prefetch_cancel drops a pending hint (end of a training loop: the announced
 * batch will not be trained).  dctr_input_slot_rewrite is safe to call from the input pipeline's thread. */
int dctr_prefetch_ids(dctr_handle h, const int32_t* d_ids_next, int B);
int dctr_prefetch_cancel(dctr_handle h);
/* time-blocked table sweep (dctr_config.table_sweep_period > 1): advances every lagging table row to global_step, on `stream`.
 * Implied by every entry point that reads the tables as a whole (predict, eval, parameter / slot access, a train step that reports
 * its loss); a no-op when no row lags.  A benchmark calls it at the end of its timed region so that every update of the timed
 * steps has been computed inside it. */
int dctr_tables_sync(dctr_handle h, void* stream);
int dctr_input_slot_rewrite(dctr_handle h, int slot);
/* The host -> device leg of the input pipeline (Dataset.prefetch, DeepFM.py:84, at device granularity) inside the library, so that
 * a Python input thread pays one call per batch and the training thread two trivial ones:
 *   fill          (input thread)    B rows of ids / vals / labels from HOST buffers (pinned for real overlap) into slot `slot`: three
 *                                   async copies on the handle's own copy stream and a "filled" event; advances the slot's generation
 *                                   like dctr_input_slot_rewrite.  h_labels may be NULL.  A whole batch (B = max_batch) whose three
 *                                   host arrays are laid out like the slot -- ids, vals each padded to a multiple of 64 elements,
 *                                   then labels -- goes in one copy.
 *   acquire       (training thread) `stream` waits, on the device, for the slot's last fill
 *   release       (training thread) records "consumed" behind everything enqueued on `stream` so far
 *   wait_released (input thread)    blocks the calling host thread until that record has been reached: slot and host buffers are free
 *   ready                           *ready = 1 when the slot's last fill has landed (never filled: 1)
 * fill and wait_released are safe to call from another thread than the one that runs the steps. */
int dctr_input_slot_fill(dctr_handle h, int slot, const int32_t* h_ids, const float* h_vals, const float* h_labels, int B);
int dctr_input_slot_acquire(dctr_handle h, int slot, void* stream);
int dctr_input_slot_release(dctr_handle h, int slot, void* stream);
int dctr_input_slot_wait_released(dctr_handle h, int slot);
int dctr_input_slot_ready(dctr_handle h, int slot, int* ready);
/* a non-blocking stream owned by the handle (created on first call, destroyed with it) for the caller's step calls: pass it as `stream`
 * instead of the legacy default stream (0), on which every event of the step and of the slot handshake costs more.  The caller orders
 * its own work against it (hipStreamSynchronize / events) as with any stream. */
int dctr_main_stream(dctr_handle h, void** stream);
/* canned-estimator models: the dense (numeric-column) inputs [B, dense_size] f32 of the NEXT train/predict/eval call; the
 * buffer is read in place and must stay valid until that call's work has finished */
int dctr_set_dense_input(dctr_handle h, const float* d_dense);
/* forward only (mode PREDICT/EVAL: dropout off, BN moving stats): d_prob [B] (may be NULL), d_logit [B] (may be NULL) */
int dctr_predict(dctr_handle h, const int32_t* d_ids, const float* d_vals, int B,
                 float* d_prob, float* d_logit, void* stream);
/* ---- CSR (multi-hot) models, DIN.py / DeepCvrMTL.py.  One batch = B examples x S = field_size slots; slot (b, s) owns the
 * entries [d_offsets[b*S+s], d_offsets[b*S+s+1]) of d_ids / d_weights (d_weights NULL = all ones); the MLP input is
 * x[b, s*K:(s+1)*K] = sum_j weights[j] * emb[ids[j], :] -- embedding_lookup (one entry) and embedding_lookup_sparse(sum)
 * (DIN.py:144-148,180-183) in one kernel, slots in the order of the script's tf.concat (DIN.py:199).  nnz <= max_entries.
 * Labels: d_y [B] (clicks); d_z [B] conversions (ESMM only, DeepCvrMTL.py:82-84; NULL for DIN).
 * h_loss (may be NULL; syncs): DIN.py:222 / DeepCvrMTL.py:222-225 evaluated before the update.  Launched eagerly (nnz varies). */
int dctr_train_step_csr(dctr_handle h, const int32_t* d_offsets, const int32_t* d_ids, const float* d_weights, int nnz,
                        const float* d_y, const float* d_z, int B, float* h_loss, void* stream);
/* forward only.  DIN: d_out0 = prob [B], d_out1 = logit [B].  ESMM: d_out0 = pctr, d_out1 = pcvr, d_out2 = pctcvr
 * (predictions dict of DeepCvrMTL.py:212).  Any out pointer may be NULL. */
int dctr_predict_csr(dctr_handle h, const int32_t* d_offsets, const int32_t* d_ids, const float* d_weights, int nnz, int B,
                     float* d_out0, float* d_out1, float* d_out2, void* stream);

/* mode EVAL of the CSR models: accumulates like dctr_eval_batch (same dctr_eval_reset / dctr_eval_result; the AUC there is
 * auc(y, prob) for DIN and CTR_AUC = auc(y, pctr) for ESMM).  ESMM's CVR_AUC = auc(z, pcvr) and CTCVR_AUC = auc(z, pctcvr)
 * (DeepCvrMTL.py:231-235) are read with dctr_eval_auc_extra(which = 1 / 2); syncs. */
int dctr_eval_batch_csr(dctr_handle h, const int32_t* d_offsets, const int32_t* d_ids, const float* d_weights, int nnz,
                        const float* d_y, const float* d_z, int B, void* stream);
int dctr_eval_auc_extra(dctr_handle h, int which, float* h_auc, void* stream);

/* mode EVAL (DeepFM.py:193-201): accumulate the loss and tf.metrics.auc's 200-threshold counters over an eval set */
int dctr_eval_reset(dctr_handle h, void* stream);
int dctr_eval_batch(dctr_handle h, const int32_t* d_ids, const float* d_vals, const float* d_labels, int B, void* stream);
/* h_loss = mean xent over the set + l2_reg * sum l2_loss(regularised variables) (DeepFM.py:188-190); syncs */
int dctr_eval_result(dctr_handle h, float* h_auc, float* h_loss, int64_t* h_examples, void* stream);
/* raises DCTR_ERR_INVALID_ARG if any id seen since the last check was out of range (syncs) */
int dctr_check_ids(dctr_handle h, void* stream);
/* named intermediates of the last forward, for parity tests ("e","y_w","y_v","bi","inner","x_cross","att") */
int dctr_debug_tensor(dctr_handle h, const char* name, float** d_ptr, int64_t* n_elems, int* ld);

/* ---- row-sharded multi-GPU path: owner(id) = id % world, local row = id / world.  There is no reference
 * call site -- this replaces the async parameter-server push/pull of set_dist_env (DeepFM.py:237-282) with a
 * synchronous step whose only exchanges are three all-to-alls (distinct rows out, rows back, row gradients back) and
 * one all-reduce of the dense gradients.  The step is enqueued natively (csrc/dist.hip, dctr_dist_train_step: RCCL resolved from the
 * librccl already in the process, grouped send/recv on three communicators); the split API below is what that driver -- and the
 * torch.distributed orchestration kept as a readable reference -- is built from.
 * Rows and row gradients cross the fabric as PACKED records of K+4 floats: K embedding floats | linear weight | 3 pad
 * (16-byte aligned; one all-to-all per direction covers both tables).
 * Requester side, on a dctr_group_t created over the GLOBAL id space (rows = feature_size):
 *   dctr_group_ids -> dctr_route_unique -> [all-to-all local rows] -> dctr_entry_index            (routing: depends only on
 *                                                                                                  the ids, may run a step ahead)
 *   [all-to-all packed rows] -> dctr_sharded_forward_backward -> dctr_sharded_pack_row_grads -> [all-to-all packed grads]
 * Owner side: dctr_table_group_rows (with the routing) / dctr_table_gather_packed / dctr_table_apply_packed. */
/* d_counts int32[2*world]: [0:world) distinct ids per owner (the all-to-all send split sizes), rest is scratch.
 * d_send_rows [U] local rows grouped by owner; d_upos [U] position of distinct id u in that send order. */
int dctr_route_unique(dctr_group_t g, int world, int32_t* d_send_rows, int32_t* d_upos, int32_t* d_counts, void* stream);
/* d_idx[i] = position (in send order) of entry i's id, -1 for ids outside [0, rows): the `ids` of a gather whose
 * table is the buffer of rows received back from the owners */
int dctr_entry_index(dctr_group_t g, const int32_t* d_ids, int n, const int32_t* d_upos, int32_t* d_idx, void* stream);
/* owner side: packed raw rows of this rank's shard: d_out [n, K+4] */
int dctr_table_gather_packed(dctr_handle h, const int32_t* d_rows, int n, float* d_out, void* stream);
/* owner side: group the n local rows requested by all ranks (a row may be requested by several) into grouping state
 * `which` (0 or 1: two states, so the rows of step t+1 can be grouped while step t still uses its own) */
int dctr_table_group_rows(dctr_handle h, int which, const int32_t* d_rows, int n, void* stream);
/* owner side: segment-sum the n received packed row gradients d_grads [n, K+4] (same order as the rows given to
 * dctr_table_group_rows(which)) and step the shard's tables with the configured optimizer / table_mode.
 * Called directly, these three entry points always run the CLASSIC dense-exact sweep (every row of the shard every step,
 * sum theta^2 of the tables accumulated into dctr_read_scalars' [1], [2]) whatever table_sweep_period says: the time-blocked
 * sweep needs to know per step whether the loss is read, which only a step driver knows -- dctr_dist_train_step opts its handle
 * in and passes that along (h_loss == NULL: rows lag; else flush + classic sweep with the sums). */
int dctr_table_apply_packed(dctr_handle h, int which, int n, const float* d_grads, void* stream);
/* requester side: forward (+ backward through head, MLP, interaction when train != 0) on the packed rows received from
 * the owners: d_rows [n_rows, K+4] is the table the gather reads, d_idx [B,F] its ids (dctr_entry_index).  train != 0
 * also advances global_step / Adam lr_t / the dropout seed and zeroes the loss scalars.  The logit gradient is
 * (prob - label)/global_batch so that summing dense gradients over ranks gives the mean. */
int dctr_sharded_forward_backward(dctr_handle h, const float* d_rows, int n_rows, const int32_t* d_idx,
                                  const float* d_vals, const float* d_labels, int B, int global_batch, int train, void* stream);
/* per-distinct-id gradients of this rank's batch (segment sum into g's compact buffers), packed in send order:
 * d_out[d_upos[u]] = { gemb[u,:], glin[u], 0,0,0 }, d_out [U, K+4] */
int dctr_sharded_pack_row_grads(dctr_handle h, dctr_group_t g, int B, const int32_t* d_upos, float* d_out, void* stream);
/* dense gradients: reduce the partial slabs into one flat array (all-reduce it in place), then apply the optimizer */
int dctr_dense_grads(dctr_handle h, float** d_flat, int64_t* n, void* stream);
int dctr_dense_apply(dctr_handle h, void* stream);
/* [0] sum_b xent of this rank's batch, [1] sum emb^2, [2] sum linear^2 (this shard, pre-update), [3] sum of l2-regularised dense params^2 */
int dctr_read_scalars(dctr_handle h, float h_out[4], void* stream);
/* prob / logit of the last forward (device pointers into the engine's buffers, valid until the next call) */
int dctr_last_outputs(dctr_handle h, float** d_prob, float** d_logit);

/* ---- native driver of the row-sharded step: the sequence above, including its collectives, enqueued from C++ on three
 * HIP streams (rows/gradients on the caller's stream, routing of the NEXT batch and the dense all-reduce on two internal
 * ones).  Stands where the TF-1.x distributed runtime stands for the reference (tf.train.Server / replica_device_setter
 * behind set_dist_env, DeepFM.py:237-282): one process per GPU, rendezvous supplied by the host. */
typedef struct dctr_dist* dctr_dist_t;
/* Collective transport.  `channel` selects an independent communicator: 0 = rows/gradients (caller's stream), 1 = routing,
 * 2 = dense all-reduce; calls on one channel are issued in the same order on every rank.  All pointers are device
 * pointers; counts are per peer, in records; the call only ENQUEUES on `stream`.  Return DCTR_OK or an error code. */
typedef struct {
    void* ctx;
    int (*all_gather_i32)(void* ctx, int channel, const int32_t* d_send, int n, int32_t* d_recv /* [world*n] */, void* stream);
    int (*all_to_all)(void* ctx, int channel, const void* d_send, const int64_t* send_counts, void* d_recv,
                      const int64_t* recv_counts, int64_t record_bytes, void* stream);
    int (*all_reduce_f32)(void* ctx, int channel, float* d_buf, int64_t n, void* stream);   /* sum, in place */
} dctr_transport;
/* batch_norm (DeepFM.py:159-160,231-235) under data-parallel ranks: every BN layer's column sums (forward: sum y, sum y^2;
 * backward: the two gradient sums) pass through this in-place cross-rank SUM on the step's stream, so that N ranks of B examples
 * normalise like one rank of N*B.  dctr_dist_create installs the transport's all_reduce_f32 (channel 0); a host that drives
 * dctr_sharded_forward_backward itself installs its own.  world = 1 removes it. */
int dctr_set_stat_sync(dctr_handle h, int (*all_reduce_f32)(void* ctx, int channel, float* d_buf, int64_t n, void* stream), void* ctx, int world);
#define DCTR_RCCL_ID_BYTES 128
/* rank 0: one ncclUniqueId (128 bytes) per channel -- call 3 times, hand the 384 bytes to every rank (any host-side
 * rendezvous: torch.distributed store, MPI, a file).  rccl_path: librccl to load when the process has none yet (may be NULL). */
int dctr_rccl_unique_id(const char* rccl_path, char* id128);
/* RCCL transport over xGMI: ids = 3 * DCTR_RCCL_ID_BYTES bytes from rank 0.  Collective: every rank calls it. */
int dctr_dist_create_rccl(dctr_handle h, int rank, int world, const char* ids, const char* rccl_path, dctr_dist_t* out);
/* caller-supplied transport (tests: ranks sharing a GPU, staged over gloo) */
int dctr_dist_create(dctr_handle h, int rank, int world, const dctr_transport* t, dctr_dist_t* out);
int dctr_dist_destroy(dctr_dist_t d);
/* one synchronous step on this rank's B examples (global batch = B * world).  d_next_ids/next_B (optional): the ids the NEXT
 * call will be given -- they are routed while this step runs; they must have been written before this call and stay
 * untouched until that call.  h_loss (optional): the global loss (mean xent + l2 terms); asking for it synchronises. */
int dctr_dist_train_step(dctr_dist_t d, const int32_t* d_ids, const float* d_vals, const float* d_labels, int B,
                         const int32_t* d_next_ids, int next_B, float* h_loss, void* stream);
int dctr_dist_predict(dctr_dist_t d, const int32_t* d_ids, const float* d_vals, int B, float* d_prob, void* stream);

/* per-kernel timing hooks used by bench.py for the roofline objects: runs `iters` back-to-back
 * launches of the named kernel on the engine's current buffers between two hipEvents recorded on
 * `stream`, returns the average milliseconds per launch. */
int dctr_time_kernel(dctr_handle h, const char* kernel, int iters, float* h_ms_per_launch, void* stream);
/* in-step duration of the MLP's forward GEMMs, by hipEvents on the step's stream, every 32nd train step.  enable=1: a pair of records
 * around the FIRST layer's launch (a bracket: two barrier packets sit inside the interval); enable=2: every forward layer's launch
 * carries its own start / stop events (the dispatch alone -- what rocprofv3's kernel trace reports); enable=0 stops and returns the
 * average milliseconds over all timed launches and their number.  dctr_step_timer_layer: the same for one product, after the stop:
 * layer = i the forward product of MLP layer i, n_layers + i its dgrad, 2 n_layers + i its weight gradient (mode 2 times all three). */
int dctr_step_timer(dctr_handle h, int enable, float* h_avg_ms, int* h_count);
int dctr_step_timer_layer(dctr_handle h, int layer, float* h_avg_ms, int* h_count);
/* measured HBM roofline: GB/s (read + write) of a streaming float4 copy of `nbytes` (use >> 256 MB so the Infinity Cache does
 * not serve it), averaged over `iters` launches between two hipEvents on `stream` */
int dctr_measure_copy_bw(size_t nbytes, int iters, float* h_gbps, void* stream);
/* host-side evaluation of the engine's dropout decision (see "dropout sites" above): h_mask[idx] = 1 (kept) or 0 for idx in
 * [0, n).  `seed` = dctr_config.seed, `global_step` = the step the mask belongs to (dctr_get_global_step() + 1 before the
 * train call).  No device work; usable without a GPU. */
int dctr_dropout_mask(uint64_t seed, int64_t global_step, uint64_t site, int64_t n, float keep, uint8_t* h_mask);

#ifdef __cplusplus
}
#endif
#endif /* DEEPCTR_HIP_H */
