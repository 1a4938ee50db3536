// storebench.hip -- what do misaligned global stores cost on gfx950?  A wave writes 64 x BYTES contiguous bytes per
// instruction at  base + OFF  (OFF = 0: aligned; 1, 3, 4: misaligned), a workgroup writes contiguous 256 x BYTES x U,
// the grid streams over 3.4 GB.  (The LDS answer is in ldsbench.hip: 40 - 128 cycles a wave.  The decoder could skip its
// LDS text image if the global side were cheap.)
//   hipcc --offload-arch=gfx950 -O3 -o build/storebench tools/storebench.hip && build/storebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BYTES, int U>
__global__ __launch_bounds__(256) void store_kernel(uint8_t* __restrict__ out, uint64_t ntiles, uint32_t off) {
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint8_t* dst = out + 64 + off + t * (256ull * BYTES * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint8_t* p = dst + (u * 256 + threadIdx.x) * BYTES;
            if constexpr (BYTES == 16) { uint4 v = make_uint4(t, u, threadIdx.x, off); __builtin_memcpy(p, &v, 16); }
            else if constexpr (BYTES == 8) { uint64_t v = t * 31 + u + threadIdx.x; __builtin_memcpy(p, &v, 8); }
            else if constexpr (BYTES == 4) { uint32_t v = (uint32_t)t * 31 + u + threadIdx.x; __builtin_memcpy(p, &v, 4); }
            else { *p = (uint8_t)(t + u + threadIdx.x); }
        }
    }
}

template <int BYTES, int U>
int run(uint8_t* out, uint64_t bytes, uint32_t off) {
    const uint64_t ntiles = (bytes - 128) / (256ull * BYTES * U);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        store_kernel<BYTES, U><<<256 * 8, 256>>>(out, ntiles, off);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double wr = (double)ntiles * 256 * BYTES * U;
    printf("store %2d B per lane, offset %u: %.2f GB in %.3f ms = %.2f TB/s\n", BYTES, off, wr / 1e9, best, wr / 1e9 / best);
    return 0;
}

int main() {
    const uint64_t bytes = 3400ull << 20;
    uint8_t* out;
    CK(hipMalloc(&out, bytes)); CK(hipMemset(out, 0, bytes));
    for (uint32_t off : {0u, 1u, 4u, 8u}) if (run<16, 4>(out, bytes, off)) return 1;
    for (uint32_t off : {0u, 1u, 3u, 4u}) if (run<8, 8>(out, bytes, off)) return 1;
    for (uint32_t off : {0u, 1u, 2u}) if (run<4, 8>(out, bytes, off)) return 1;
    if (run<1, 8>(out, bytes, 0)) return 1;
    return 0;
}
