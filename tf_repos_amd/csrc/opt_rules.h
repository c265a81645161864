// K9 optimizer update rules (DeepFM.py:204-211), TF-1.4 semantics; shared by dense_ops.hip and wnd.hip.
#pragma once
#include "ops.h"

namespace dctr {

// EXACT = true: Adam's update term with the correctly rounded sqrtf and division -- the dense arena (the MLP / cross / attention
// weights: ~1e6 elements, not ALU-bound) always takes it.  The table kernels (dense-exact sweep, replay of lagging rows, fused
// tail) take the default: the fast forms, or the exact ones in a library built with -DDCTR_IEEE_ADAM.  That second library IS built
// (tf_repos_amd/build.py: libdeepctr_hip_ieee.so) and DCTR_IEEE_ADAM=1 in the environment loads it (capi.py); tests/test_ieee_adam_gpu.py
// runs the Adam parity suites on it.  (A run-time flag in Hyper was tried first: a branch per element inside the replay loops, which the
// compiler can no longer interleave across rows -- the flag belongs at compile time.)
#ifdef DCTR_IEEE_ADAM
constexpr bool ADAM_TABLES_EXACT = true;
#else
constexpr bool ADAM_TABLES_EXACT = false;
#endif
template <bool EXACT = ADAM_TABLES_EXACT>
__device__ __forceinline__ void opt_update(int kind, const Hyper& h, float& th, float& s0, float& s1, float g) {
    switch (kind) {
        case DCTR_OPT_ADAM: {           // m,v ; theta -= lr_t m/(sqrt(v)+eps), lr_t = lr sqrt(1-b2^t)/(1-b1^t)
            s0 = h.beta1 * s0 + (1.0f - h.beta1) * g;
            s1 = h.beta2 * s1 + (1.0f - h.beta2) * g * g;
            if constexpr (EXACT) {
                th = th - h.lr_t * s0 / (sqrtf(s1) + h.eps);
            } else {
                // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sqrtf and division (~10 instructions each):
                // the update term lr_t m / (sqrt(v) + eps) is ~1e-3 of theta, so its 2-3 ulp change theta by < 1e-9 relative --
                // three orders below the parity tolerances -- while the dense Adam over every table row (and the replay of lagging
                // rows, lag.h) is ALU work the step pays for: ~40 -> ~15 instructions per element
                th = th - (h.lr_t * s0) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(s1) + h.eps);
            }
        } break;
        case DCTR_OPT_ADAGRAD: {        // accum += g^2 ; theta -= lr g / sqrt(accum)
            s0 = s0 + g * g;
            th = th - h.lr * g / sqrtf(s0);
        } break;
        case DCTR_OPT_MOMENTUM: {       // accum = mom*accum + g ; theta -= lr accum
            s0 = h.momentum * s0 + g;
            th = th - h.lr * s0;
        } break;
        case DCTR_OPT_FTRL: {           // lr_power = -0.5, l1 = l2 = 0: s0 = accum, s1 = linear
            const float na = s0 + g * g;
            const float sigma = (sqrtf(na) - sqrtf(s0)) / h.lr;
            s1 = s1 + g - sigma * th;
            th = -s1 / (sqrtf(na) / h.lr);
            s0 = na;
        } break;
    }
}

__device__ __forceinline__ Hyper load_hyper(const Hyper* dev, const Hyper& val) { return dev ? *dev : val; }

}  // namespace dctr
