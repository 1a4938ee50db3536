// mallbench.hip -- does the 256 MiB Infinity Cache serve the 2nd and 3rd read of a chunk of the FASTQ stream?
// (the encode step reads the stream three times -- census, statistics, pack.  If the three kernels walk the
// stream chunk by chunk, the later reads of a chunk can come from the on-die cache instead of HBM.)
//   (1) re-read: one read-only kernel launched repeatedly over the same S bytes, S = 16 MiB .. 1 GiB
//   (2) chunked triple: for each chunk of a 3.4 GiB buffer: read, read, read + write 0.4x -- against the same three
//       kernels each over the whole buffer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mallbench tools/mallbench.hip && tools/mallbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int RD, int WR>
__global__ __launch_bounds__(256) void mix_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t ntiles) {
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint4* src = in + t * (256 * RD);
        uint4* dst = out + t * (256 * WR);
        uint4 v[RD];
#pragma unroll
        for (int u = 0; u < RD; ++u) v[u] = src[u * 256 + threadIdx.x];
        uint4 acc = v[0];
#pragma unroll
        for (int u = WR; u < RD; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
        if (WR > 0) {
            v[0] = acc;
#pragma unroll
            for (int u = 0; u < WR; ++u) dst[u * 256 + threadIdx.x] = v[u];
        } else if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;
    }
}

int main() {
    const uint64_t bytes = 3400ull << 20;
    uint4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# (1) read-only kernel, same S bytes 20 times back to back (first launch discarded)\n");
    for (uint64_t mib : {16, 32, 64, 96, 128, 192, 256, 512, 1024}) {
        const uint64_t nt = (mib << 20) / (256 * 4 * 16);
        const uint32_t grid = nt < 2048 ? (uint32_t)nt : 2048;
        mix_kernel<4, 0><<<grid, 256>>>(in, out, nt);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) mix_kernel<4, 0><<<grid, 256>>>(in, out, nt);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("S = %4llu MiB  %.2f us per launch  %.2f TB/s\n", (unsigned long long)mib, ms * 1000 / 20, (double)(mib << 20) * 20 / 1e9 / ms);
    }
    printf("# (2) three passes over 3.4 GiB (read; read; read + write 2/5): whole buffer per kernel vs chunk by chunk\n");
    const uint64_t nt_all = bytes / (256 * 5 * 16);
    for (uint64_t mib : {0, 32, 64, 96, 128, 256}) {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            if (mib == 0) {
                mix_kernel<5, 0><<<2048, 256>>>(in, out, nt_all);
                mix_kernel<5, 0><<<2048, 256>>>(in, out, nt_all);
                mix_kernel<5, 2><<<2048, 256>>>(in, out, nt_all);
            } else {
                const uint64_t ct = (mib << 20) / (256 * 5 * 16);
                for (uint64_t t0 = 0; t0 < nt_all; t0 += ct) {
                    const uint64_t n = nt_all - t0 < ct ? nt_all - t0 : ct;
                    const uint32_t grid = n < 2048 ? (uint32_t)n : 2048;
                    mix_kernel<5, 0><<<grid, 256>>>(in + t0 * 256 * 5, out, n);
                    mix_kernel<5, 0><<<grid, 256>>>(in + t0 * 256 * 5, out, n);
                    mix_kernel<5, 2><<<grid, 256>>>(in + t0 * 256 * 5, out + t0 * 256 * 2, n);
                }
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("chunk %4llu MiB%s  %.3f ms for the three passes\n", (unsigned long long)mib, mib ? "" : " (whole buffer)", best);
    }
    return 0;
}
