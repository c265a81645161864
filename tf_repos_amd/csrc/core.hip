// Library-level entry points: error string, device selection, raw memory helpers.
#include "common.h"

namespace dctr {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace dctr

using namespace dctr;

extern "C" {

int dctr_version(void) { return 100; }
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
