/* The C ABI without Python or torch: a DeepFM engine created, fed and trained from plain C (gcc), the way a non-Python host
 * (the reference's C++ serving clients, a Go / Java trainer over cgo / JNI) would bind include/deepctr_hip.h.
 *
 *   gcc -O2 -Iinclude examples/c_abi_train.c -o /tmp/c_abi_train -Ltf_repos_amd/_lib -ldeepctr_hip -Wl,-rpath,$PWD/tf_repos_amd/_lib -lm
 *   /tmp/c_abi_train            # prints one loss per step (the same numbers tf_repos_amd.engine.Engine gives on these inputs)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "deepctr_hip.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != DCTR_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dctr_last_error());   \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

/* xorshift: the same stream on every host, so the Python side of the test can reproduce the inputs */
static uint64_t rng_state = 88172645463325252ull;
static uint32_t next_u32(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}
static float next_unit(void) { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }

int main(int argc, char** argv) {
    const int B = 64, F = 39, K = 8, steps = argc > 1 ? atoi(argv[1]) : 3;
    const int64_t V = 2000;
    dctr_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.model = DCTR_MODEL_DEEPFM;
    cfg.field_size = F; cfg.embedding_size = K; cfg.feature_size = V;
    cfg.n_deep_layers = 2; cfg.deep_layers[0] = 32; cfg.deep_layers[1] = 16;
    for (int i = 0; i < DCTR_MAX_LAYERS; ++i) cfg.keep_prob[i] = 1.0f;
    cfg.l2_reg = 1e-3f; cfg.learning_rate = 1e-2f; cfg.optimizer = DCTR_OPT_ADAGRAD; cfg.table_mode = DCTR_TABLE_DENSE_EXACT;
    cfg.max_batch = B; cfg.shard_world = 1;
    CHECK(dctr_set_device(0));
    dctr_handle h = NULL;
    CHECK(dctr_create(&cfg, &h));

    /* weights: every parameter by name, the way a checkpoint loader would inject fm_v / fm_w / mlp weights */
    int n_params = 0;
    CHECK(dctr_param_count(h, &n_params));
    for (int p = 0; p < n_params; ++p) {
        const char* name; int rank; int64_t dims[4];
        CHECK(dctr_param_info(h, p, &name, &rank, dims));
        size_t n = 1;
        for (int d = 0; d < rank; ++d) n *= (size_t)dims[d];
        float* w = (float*)malloc(n * sizeof(float));
        for (size_t i = 0; i < n; ++i) w[i] = 0.1f * (next_unit() - 0.5f);
        CHECK(dctr_param_set(h, name, w, n * sizeof(float)));
        free(w);
    }

    /* one batch in device memory obtained from the library itself (any allocator works: the ABI takes raw pointers) */
    int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * B * F);
    float* vals = (float*)malloc(sizeof(float) * B * F);
    float* labels = (float*)malloc(sizeof(float) * B);
    for (int i = 0; i < B * F; ++i) { ids[i] = (int32_t)(next_u32() % V); vals[i] = next_unit(); }
    for (int i = 0; i < B; ++i) labels[i] = next_unit() < 0.3f ? 1.0f : 0.0f;
    void *d_ids, *d_vals, *d_labels;
    CHECK(dctr_malloc(&d_ids, sizeof(int32_t) * B * F));
    CHECK(dctr_malloc(&d_vals, sizeof(float) * B * F));
    CHECK(dctr_malloc(&d_labels, sizeof(float) * B));
    CHECK(dctr_memcpy_h2d(d_ids, ids, sizeof(int32_t) * B * F, NULL));
    CHECK(dctr_memcpy_h2d(d_vals, vals, sizeof(float) * B * F, NULL));
    CHECK(dctr_memcpy_h2d(d_labels, labels, sizeof(float) * B, NULL));

    for (int s = 0; s < steps; ++s) {
        float loss = 0.f;
        if (s & 1) {
            /* the same batch from HOST memory through an input slot: the library copies (async when the buffers are pinned), the
             * step waits for the copy on the device and reads the slot in place -- what an input thread does per batch */
            const int k = s % DCTR_INPUT_SLOTS;
            int ready = 0;
            int32_t* s_ids; float *s_vals, *s_labels;
            CHECK(dctr_input_slot_wait_released(h, k));                 /* (nothing has used slot k yet: returns at once) */
            CHECK(dctr_input_slot_fill(h, k, ids, vals, labels, B));
            CHECK(dctr_input_slot_ready(h, k, &ready));                 /* 0 while the copies are in flight */
            CHECK(dctr_input_slot_acquire(h, k, NULL));
            CHECK(dctr_input_slot(h, k, &s_ids, &s_vals, &s_labels));
            CHECK(dctr_train_step(h, s_ids, s_vals, s_labels, B, &loss, NULL));
            CHECK(dctr_input_slot_release(h, k, NULL));
        } else {
            CHECK(dctr_train_step(h, (const int32_t*)d_ids, (const float*)d_vals, (const float*)d_labels, B, &loss, NULL));
        }
        printf("step %d loss %.7f\n", s, loss);
    }
    CHECK(dctr_check_ids(h, NULL));
    CHECK(dctr_free(d_ids)); CHECK(dctr_free(d_vals)); CHECK(dctr_free(d_labels));
    CHECK(dctr_destroy(h));
    free(ids); free(vals); free(labels);
    return 0;
}
