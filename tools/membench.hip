// membench.hip -- what does HBM give a streaming kernel with a given read : write mix?  (Context for the roofline
// fractions in DESIGN.md: the pack kernel reads 3.8 GB and writes 1.5 GB per launch.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench tools/membench.hip && tools/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every thread reads RD 16-byte vectors and writes WR of them (WR <= RD), grid-stride over tiles of 256 * RD vectors
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

template <int RD, int WR>
int run(const uint4* in, uint4* out, uint64_t nvec, const char* name) {
    const uint64_t ntiles = nvec / (256 * RD);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0));
        mix_kernel<RD, WR><<<256 * 8, 256>>>(in, out, ntiles);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double rd = (double)ntiles * 256 * RD * 16, wr = (double)ntiles * 256 * WR * 16;
    printf("%-34s read %.2f GB  write %.2f GB  %.3f ms  total %.2f TB/s\n", name, rd / 1e9, wr / 1e9, best, (rd + wr) / 1e9 / best);
    return 0;
}

int main() {
    const uint64_t bytes = 3400ull << 20;
    uint4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    const uint64_t nvec = bytes / 16;
    if (run<5, 0>(in, out, nvec, "read only")) return 1;
    if (run<8, 1>(in, out, nvec, "8 : 1 (census: stream + bitmap)")) return 1;
    if (run<5, 2>(in, out, nvec, "5 : 2 (pack: 3.8 GB in, 1.5 GB out)")) return 1;
    if (run<4, 2>(in, out, nvec, "2 : 1")) return 1;
    if (run<4, 4>(in, out, nvec, "1 : 1 (copy)")) return 1;
    return 0;
}
