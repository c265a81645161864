// Library-level entry points: error string, device selection, raw memory helpers.
#include "common.h"

namespace dctr {
static thread_local char g_err[1024] = "";
static thread_local hipEvent_t g_stop_event = nullptr;
void arm_stop_event(hipEvent_t ev) { g_stop_event = ev; }
hipEvent_t take_stop_event() { hipEvent_t e = g_stop_event; g_stop_event = nullptr; return e; }
bool stop_event_pending() { return g_stop_event != nullptr; }
void disarm_stop_event() { g_stop_event = nullptr; }
static thread_local hipEvent_t g_timer_start = nullptr, g_timer_stop = nullptr;
void arm_timer_events(hipEvent_t start, hipEvent_t stop) { g_timer_start = start; g_timer_stop = stop; }
bool take_timer_events(hipEvent_t* start, hipEvent_t* stop) {
    if (g_timer_start == nullptr) return false;
    *start = g_timer_start; *stop = g_timer_stop;
    g_timer_start = g_timer_stop = nullptr;
    return true;
}
bool timer_events_pending() { return g_timer_start != nullptr; }
void disarm_timer_events() { g_timer_start = g_timer_stop = nullptr; }
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace dctr

using namespace dctr;

// streaming float4 copy: the "measured HBM roofline" the gather / optimizer fractions are quoted against
// (one float4 per lane and a grid that covers the buffer: measured 6.2 TB/s; grid-stride variants of the same copy reach 4.4)
__global__ __launch_bounds__(256) void copy_f4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

extern "C" {

int dctr_measure_copy_bw(size_t nbytes, int iters, float* h_gbps, void* stream) {
    DCTR_REQUIRE(h_gbps != nullptr && nbytes >= (1u << 20) && iters > 0, "bad argument");
    hipStream_t st = as_stream(stream);
    const size_t n4 = nbytes / 16;
    float4 *a = nullptr, *b = nullptr;
    DCTR_HIP_CHECK(hipMalloc(&a, n4 * 16));
    if (hipMalloc(&b, n4 * 16) != hipSuccess) { hipFree(a); set_error("copy probe: out of memory"); return DCTR_ERR_HIP; }
    hipMemsetAsync(a, 0, n4 * 16, st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)((n4 + 255) / 256);
    copy_f4_kernel<<<grid, 256, 0, st>>>(a, b, n4);       // warm-up
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) copy_f4_kernel<<<grid, 256, 0, st>>>(a, b, n4);
    hipEventRecord(e1, st);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    if (e != hipSuccess) { set_error("copy probe failed: %s", hipGetErrorString(e)); return DCTR_ERR_HIP; }
    *h_gbps = (float)(2.0 * (double)(n4 * 16) * iters / ((double)ms * 1e6));     // read + write bytes per second / 1e9
    return DCTR_OK;
}

int dctr_dropout_mask(uint64_t seed, int64_t global_step, uint64_t site, int64_t n, float keep, uint8_t* h_mask) {
    DCTR_REQUIRE(h_mask != nullptr && n >= 0, "bad argument");
    DCTR_REQUIRE(keep > 0.f && keep <= 1.f, "keep_prob must be in (0,1], got %f", keep);
    const uint64_t s = (seed ^ ((uint64_t)global_step * STEP_SEED_MULT)) ^ site;      // StepState::seed_t ^ site
    for (int64_t i = 0; i < n; ++i) h_mask[i] = (keep >= 1.f || dropout_scale(s, (uint64_t)i, keep) != 0.f) ? 1 : 0;
    return DCTR_OK;
}

int dctr_version(void) { return 106; }          // (106: dctr_config.gemm_mode 0 = the library default = split, 2 = exact)
const char* dctr_last_error(void) { return get_error(); }

int dctr_device_count(int* n) {
    DCTR_REQUIRE(n != nullptr, "null out pointer");
    DCTR_HIP_CHECK(hipGetDeviceCount(n));
    return DCTR_OK;
}
int dctr_set_device(int dev) {
    DCTR_HIP_CHECK(hipSetDevice(dev));
    return DCTR_OK;
}
int dctr_malloc(void** d_ptr, size_t nbytes) {
    DCTR_REQUIRE(d_ptr != nullptr, "null out pointer");
    DCTR_HIP_CHECK(hipMalloc(d_ptr, nbytes ? nbytes : 4));
    return DCTR_OK;
}
int dctr_free(void* d_ptr) {
    if (d_ptr) DCTR_HIP_CHECK(hipFree(d_ptr));
    return DCTR_OK;
}
int dctr_memcpy_h2d(void* d_dst, const void* h_src, size_t nbytes, void* stream) {
    DCTR_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, as_stream(stream)));
    DCTR_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return DCTR_OK;
}
int dctr_memcpy_d2h(void* h_dst, const void* d_src, size_t nbytes, void* stream) {
    DCTR_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, as_stream(stream)));
    DCTR_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return DCTR_OK;
}
int dctr_memset(void* d_dst, int value, size_t nbytes, void* stream) {
    DCTR_HIP_CHECK(hipMemsetAsync(d_dst, value, nbytes, as_stream(stream)));
    return DCTR_OK;
}
int dctr_stream_sync(void* stream) {
    DCTR_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return DCTR_OK;
}

}  // extern "C"
