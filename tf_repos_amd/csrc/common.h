// Shared helpers for libdeepctr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/deepctr_hip.h"

namespace dctr {

void set_error(const char* fmt, ...);
// A record that would follow a kernel launch on the same stream (hipEventRecord = a barrier packet of its own: the stream's pipeline
// drains, ~5 us in the step's timeline) can ride on the launch instead: hipExtLaunchKernel's stopEvent is bound to the dispatch packet's
// own completion signal.  The engine ARMS an event (arm_stop_event) right before calling the op whose kernel is the last thing the fork
// depends on; the launch site that supports it TAKES it (take_stop_event, at most once); the engine then checks stop_event_taken().
void arm_stop_event(hipEvent_t ev);
hipEvent_t take_stop_event();          // the armed event (and disarms), or nullptr
bool stop_event_pending();             // still armed: nobody took it (the caller records it the ordinary way and disarms)
void disarm_stop_event();
// ... and a launch's own START and STOP events for timing it (dctr_step_timer mode 2): the pair brackets exactly the dispatch, with no
// barrier packet inside the interval.  Armed by the engine right before the op; a launch site that supports it takes both.
void arm_timer_events(hipEvent_t start, hipEvent_t stop);
bool take_timer_events(hipEvent_t* start, hipEvent_t* stop);      // true: taken (and disarmed)
bool timer_events_pending();
void disarm_timer_events();
const char* get_error();
// a launch that carries the armed event, if there is one (kernel templates with commas in their argument list go in parentheses)
#define DCTR_LAUNCH_RIDE(kern, grid, block, lds, st, ...)                                                                     \
    do {                                                                                                                      \
        if (hipEvent_t stop_ = ::dctr::take_stop_event())                                                                     \
            hipExtLaunchKernelGGL(kern, grid, block, lds, st, nullptr, stop_, 0, __VA_ARGS__);                                \
        else                                                                                                                  \
            hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                      \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define DCTR_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::dctr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                      \
            return DCTR_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define DCTR_LAUNCH_CHECK()                                                                   \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess) {                                                               \
            ::dctr::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),      \
                              __FILE__, __LINE__);                                            \
            return DCTR_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define DCTR_REQUIRE(cond, ...)                                                               \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            ::dctr::set_error(__VA_ARGS__);                                                   \
            return DCTR_ERR_INVALID_ARG;                                                      \
        }                                                                                     \
    } while (0)

#define DCTR_TRY(expr)                                                                        \
    do {                                                                                      \
        int _rc = (expr);                                                                     \
        if (_rc != DCTR_OK) return _rc;                                                       \
    } while (0)

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// counter-based RNG for dropout: one 32-bit hash per element, keyed by (seed, index); seed = StepState::seed_t ^ site
// (include/deepctr_hip.h "dropout sites"; dctr_dropout_mask in core.hip is the host-side evaluation of the same function).
// u in [0,1): mask = floor(keep + u) = (u >= 1-keep)   [nn.dropout, TF-1.4]
constexpr uint64_t STEP_SEED_MULT = 0xD1B54A32D192ED03ULL;      // seed_t = seed ^ global_step * STEP_SEED_MULT (step_state_kernel)
__host__ __device__ __forceinline__ uint32_t hash32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}
// Every kernel that draws a dropout mask gets `seed_ptr` = &StepState::seed_t (ops.h) or null (op-level calls with an explicit
// seed); the word behind it is StepState::row0, the global row of this rank's first example: element (row, col) of a [rows, width]
// site is drawn at index (row0 + row) * width + col.
__device__ __forceinline__ uint64_t dropout_row0(const uint64_t* seed_ptr) { return seed_ptr != nullptr ? seed_ptr[1] : 0ull; }
__host__ __device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float keep) {
    const float u = (float)(hash32(seed ^ (idx * 0x9E3779B97F4A7C15ULL)) >> 8) * (1.0f / 16777216.0f);
    return (u >= 1.0f - keep) ? 1.0f / keep : 0.0f;
}

}  // namespace dctr
