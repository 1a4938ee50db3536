// ldsbench.hip -- what do LDS accesses of the decoder's shapes cost on gfx950?  One workgroup of 256 lanes per CU slot
// (4 per CU), each lane in a loop of dependent-free LDS operations at  base + lane * STRIDE + OFF:
//   write b64 / read b64 at offsets 0 (aligned), 1, 4; write b32 aligned / unaligned; write b8; read u16 odd
// Prints cycles per wave-instruction (wall clock x 2.4 GHz / instructions per SIMD... reported as ns per instruction per CU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ldsbench tools/ldsbench.hip && tools/ldsbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096, UNROLL = 8;

template <int BYTES, bool WRITE>
__global__ __launch_bounds__(256) void lds_kernel(uint32_t stride, uint32_t off, uint64_t* out) {
    extern __shared__ __align__(16) uint8_t lds[];
    for (uint32_t i = threadIdx.x; i < 9216; i += 256) ((uint32_t*)lds)[i] = i * 2654435761u;
    __syncthreads();
    uint8_t* p = lds + threadIdx.x * stride + off;
    uint64_t acc = threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            uint8_t* q = p + u * 16 * ((BYTES + 15) / 16) * 0 + ((it + u) & 7) * 2048;      // 8 disjoint 2 KiB windows + lane offsets
            if constexpr (WRITE) {
                if constexpr (BYTES == 8) { uint64_t v = acc + u; __builtin_memcpy(q, &v, 8); }
                else if constexpr (BYTES == 4) { uint32_t v = (uint32_t)acc + u; __builtin_memcpy(q, &v, 4); }
                else if constexpr (BYTES == 2) { uint16_t v = (uint16_t)(acc + u); __builtin_memcpy(q, &v, 2); }
                else { *q = (uint8_t)(acc + u); }
            } else {
                if constexpr (BYTES == 8) { uint64_t v; __builtin_memcpy(&v, q, 8); acc += v; }
                else if constexpr (BYTES == 4) { uint32_t v; __builtin_memcpy(&v, q, 4); acc += v; }
                else if constexpr (BYTES == 2) { uint16_t v; __builtin_memcpy(&v, q, 2); acc += v; }
                else { acc += *q; }
            }
        }
        if constexpr (WRITE) acc = acc * 3 + 1;
    }
    __syncthreads();
    if constexpr (WRITE) acc += lds[threadIdx.x];
    if (acc == 0x123456789ull) out[0] = acc;
}

template <int BYTES, bool WRITE>
int run(const char* what, uint32_t stride, uint32_t off, uint64_t* d_out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 4;
    const size_t lds = 36 * 1024;
    lds_kernel<BYTES, WRITE><<<blocks, 256, lds>>>(stride, off, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    lds_kernel<BYTES, WRITE><<<blocks, 256, lds>>>(stride, off, d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // per CU: 16 waves x ITER x UNROLL wave-instructions share one LDS
    const double instr = 16.0 * ITER * UNROLL;
    printf("%-34s stride %2u off %u: %7.3f ms  %6.2f ns per wave-instruction per CU (%.1f cycles at 2.4 GHz), %6.1f B/clk/CU\n", what, stride, off, ms,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4, 64.0 * BYTES / (ms * 1e6 / instr * 2.4));
    return 0;
}

int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 64));
    for (uint32_t off : {0u, 1u, 2u, 4u}) { run<8, true>("ds_write_b64", 8, off, d_out); }
    for (uint32_t off : {0u, 1u, 2u, 4u}) { run<8, false>("ds_read_b64", 8, off, d_out); }
    for (uint32_t off : {0u, 1u, 2u}) { run<8, false>("ds_read_b64 (6-byte stride)", 6, off, d_out); }
    for (uint32_t off : {0u, 1u, 2u}) { run<4, true>("ds_write_b32", 4, off, d_out); }
    for (uint32_t off : {0u, 1u, 2u}) { run<4, false>("ds_read_b32", 4, off, d_out); }
    for (uint32_t off : {0u, 1u}) { run<2, false>("ds_read_u16", 2, off, d_out); }
    for (uint32_t off : {0u, 1u}) { run<2, true>("ds_write_b16", 2, off, d_out); }
    run<1, true>("ds_write_b8", 1, 0, d_out);
    run<1, true>("ds_write_b8 (stride 4)", 4, 0, d_out);
    run<1, false>("ds_read_u8", 1, 0, d_out);
    return 0;
}
