// api.hip -- context, memory and error entry points of the C ABI (include/uqhip.h).
// Replaces the reference's cffi -> libc malloc/free ownership (uq.py:111-126, 712-713).
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[1024] = "";

void uq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* uq_last_error(void) { return g_err; }
extern "C" int uq_abi_version(void) { return UQ_ABI_VERSION; }

extern "C" int uq_device_count(int* h_count) {
    UQ_REQUIRE(h_count, "uq_device_count: null out pointer");
    UQ_CHECK_HIP(hipGetDeviceCount(h_count));
    return 0;
}

extern "C" int uq_ctx_create(int device, void* stream, uq_ctx** out) {
    UQ_REQUIRE(out, "uq_ctx_create: null out pointer");
    int n = 0;
    UQ_CHECK_HIP(hipGetDeviceCount(&n));
    UQ_REQUIRE(device >= 0 && device < n, "uq_ctx_create: device %d out of range (%d visible)", device, n);
    UQ_CHECK_HIP(hipSetDevice(device));
    uq_ctx* c = new uq_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { UQ_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    UQ_CHECK_HIP(hipEventCreate(&c->ev0));
    UQ_CHECK_HIP(hipEventCreate(&c->ev1));
    UQ_CHECK_HIP(hipHostMalloc((void**)&c->h_pinned, 65536, hipHostMallocDefault));
    UQ_CHECK_HIP(hipHostGetDevicePointer((void**)&c->d_pinned, c->h_pinned, 0));
    UQ_CHECK_HIP(hipMalloc((void**)&c->d_async, 64));
    *out = c;
    return 0;
}

extern "C" int uq_ctx_destroy(uq_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->idx_partials) (void)hipFree(c->idx_partials);
    if (c->idx_bitmap) (void)hipFree(c->idx_bitmap);
    if (c->scan_ws) (void)hipFree(c->scan_ws);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->d_async) (void)hipFree(c->d_async);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

namespace {
__global__ void read_back_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t nwords) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace

int uq_read_back(uq_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    const size_t off = (size_t)((const uint8_t*)h_dst - (const uint8_t*)c->h_pinned);
    UQ_REQUIRE(bytes % 4 == 0 && off % 4 == 0 && off + bytes <= 65536, "uq_read_back: destination outside the pinned staging buffer");
    if (bytes == 0) return 0;
    const uint32_t nwords = (uint32_t)(bytes / 4);
    read_back_kernel<<<nwords > 1024 ? 8 : 1, 256, 0, c->stream>>>((const uint32_t*)d_src, (uint32_t*)((uint8_t*)c->d_pinned + off), nwords);
    UQ_LAUNCH_CHECK();
    return 0;
}

int uq_scratch(uq_ctx* c, size_t bytes, void** out) {
    if (bytes > c->scratch_bytes) {
        UQ_CHECK_HIP(hipSetDevice(c->device));
        UQ_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (c->scratch) { UQ_CHECK_HIP(hipFree(c->scratch)); c->scratch = nullptr; c->scratch_bytes = 0; }
        size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
        UQ_CHECK_HIP(hipMalloc(&c->scratch, want));
        c->scratch_bytes = want;
    }
    *out = c->scratch;
    return 0;
}

extern "C" int uq_ctx_reserve(uq_ctx* c, size_t bytes) {
    UQ_REQUIRE(c, "null context");
    void* p;
    return uq_scratch(c, bytes, &p);
}

extern "C" int uq_ctx_sync(uq_ctx* c) {
    UQ_REQUIRE(c, "null context");
    UQ_CHECK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int uq_dev_alloc(uq_ctx* c, size_t bytes, void** d_out) {
    UQ_REQUIRE(c && d_out, "uq_dev_alloc: null argument");
    UQ_CHECK_HIP(hipSetDevice(c->device));
    UQ_CHECK_HIP(hipMalloc(d_out, bytes ? bytes : 1));
    return 0;
}

extern "C" int uq_dev_free(uq_ctx* c, void* p) {
    UQ_REQUIRE(c, "null context");
    if (!p) return 0;
    UQ_CHECK_HIP(hipStreamSynchronize(c->stream));
    UQ_CHECK_HIP(hipFree(p));
    return 0;
}

extern "C" int uq_h2d(uq_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    UQ_REQUIRE(c, "null context");
    if (!bytes) return 0;
    UQ_CHECK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    UQ_CHECK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int uq_d2h(uq_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    UQ_REQUIRE(c, "null context");
    if (!bytes) return 0;
    UQ_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    UQ_CHECK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// Test aid: LDS keeps its contents between launches, so a kernel that reads LDS it never wrote usually sees the (correct)
// bytes its previous launch left there.  This fills every CU's LDS with `pattern` so that such a read shows in the result.
__global__ __launch_bounds__(256) void scribble_lds_kernel(uint32_t pattern, uint32_t words, uint32_t* sink) {
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < words; i += 256) lds[i] = pattern ^ i;
    __syncthreads();
    if (lds[(threadIdx.x * 61u) % words] == 0x12345678u && sink) *sink = 1;     // keeps the stores alive
}

extern "C" int uq_debug_scribble_lds(uq_ctx* c, uint32_t pattern) {
    UQ_REQUIRE(c, "null context");
    const uint32_t bytes = 40u * 1024u;                 // four resident workgroups cover a CU's 160 KiB
    scribble_lds_kernel<<<UQ_NUM_CU * 16, 256, bytes, c->stream>>>(pattern, bytes / 4, nullptr);
    UQ_LAUNCH_CHECK();
    return 0;
}

extern "C" int uq_memset(uq_ctx* c, void* d_dst, int value, size_t bytes) {
    UQ_REQUIRE(c, "null context");
    if (!bytes) return 0;
    UQ_CHECK_HIP(hipMemsetAsync(d_dst, value, bytes, c->stream));
    return 0;
}

extern "C" int uq_timer_start(uq_ctx* c) {
    UQ_REQUIRE(c, "null context");
    UQ_CHECK_HIP(hipEventRecord(c->ev0, c->stream));
    return 0;
}

extern "C" int uq_timer_stop(uq_ctx* c, float* h_ms) {
    UQ_REQUIRE(c && h_ms, "uq_timer_stop: null argument");
    UQ_CHECK_HIP(hipEventRecord(c->ev1, c->stream));
    UQ_CHECK_HIP(hipEventSynchronize(c->ev1));
    UQ_CHECK_HIP(hipEventElapsedTime(h_ms, c->ev0, c->ev1));
    return 0;
}
