// ldsatomic.hip -- what does a no-return LDS atomic add (ds_add_u32) cost on gfx950, by address pattern?  The (base, quality) pair
// counters of the pack + statistics kernel are eight such atomics per group of eight symbols; this bench prices the layouts a
// replicated count table can take.  256 lanes per workgroup, 4 workgroups per CU slot, each lane in a loop of independent atomics.
//   mode 0  lane * 4                         (one address per bank: conflict-free)
//   mode 1  random bin < NB, 4 copies u32    (bin * 4 + (lane & 3)) * 4          -- round 2's table
//   mode 2  random bin, 8 copies u32         (bin * 8 + (lane & 7)) * 4
//   mode 3  random bin, 16 copies of 16-bit counters packed two to a word: ((bin >> 1) * 16 + (lane & 15)) * 4, data 1 << 16 (bin & 1)
//   mode 4  random bin, 32 copies packed     ((bin >> 1) * 32 + (lane & 31)) * 4
//   mode 5  random bin, 16 copies u32        (bin * 16 + (lane & 15)) * 4
//   mode 6  every lane the same address
//   mode 7  two lanes per bank, different addresses (deterministic 2-way)
//   mode 8  four lanes per bank (deterministic 4-way)
//   mode 9  random bin, 32 copies u32        (bin * 32 + (lane & 31)) * 4
// Prints ns and cycles (at 2.4 GHz) per wave-instruction per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ldsatomic tools/ldsatomic.hip && tools/ldsatomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 2048, UNROLL = 8;
constexpr uint32_t NB = 164;      // 4 bases x 41 qualities

template <int MODE>
__global__ __launch_bounds__(256) void atomic_kernel(uint64_t* out) {
    extern __shared__ __align__(16) uint32_t tab[];
    for (uint32_t i = threadIdx.x; i < 9216; i += 256) tab[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            s = s * 1664525u + 1013904223u;
            const uint32_t bin = ((s >> 16) * NB) >> 16;
            uint32_t w, d = 1;
            if (MODE == 0) w = lane;
            else if (MODE == 1) w = bin * 4 + (lane & 3);
            else if (MODE == 2) w = bin * 8 + (lane & 7);
            else if (MODE == 3) { w = (bin >> 1) * 16 + (lane & 15); d = 1u << ((bin & 1) << 4); }
            else if (MODE == 4) { w = (bin >> 1) * 32 + (lane & 31); d = 1u << ((bin & 1) << 4); }
            else if (MODE == 5) w = bin * 16 + (lane & 15);
            else if (MODE == 6) w = 5;
            else if (MODE == 7) w = (lane & 15) + 32 * ((lane >> 4) & 1) + 64 * (lane >> 5);
            else if (MODE == 8) w = (lane & 7) + 32 * ((lane >> 3) & 3) + 128 * (lane >> 5);
            else w = bin * 32 + (lane & 31);
            __hip_atomic_fetch_add(&tab[w], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    uint64_t acc = tab[threadIdx.x];
    if (acc == 0x123456789ull) out[0] = acc;
}

template <int MODE>
int run(const char* what, uint64_t* d_out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 4;
    const size_t lds = 36 * 1024;
    atomic_kernel<MODE><<<blocks, 256, lds>>>(d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    atomic_kernel<MODE><<<blocks, 256, lds>>>(d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = 16.0 * ITER * UNROLL;      // per CU: 16 waves share one LDS
    printf("mode %d %-58s %7.3f ms  %6.2f ns per wave-instruction per CU (%.1f cycles at 2.4 GHz)\n", MODE, what, ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
    return 0;
}

int main() {
    uint64_t* d_out; CK(hipMalloc(&d_out, 64));
    run<0>("lane * 4 (conflict-free)", d_out);
    run<7>("two lanes per bank (2-way)", d_out);
    run<8>("four lanes per bank (4-way)", d_out);
    run<6>("all lanes one address", d_out);
    run<1>("random bin of 164, 4 copies u32 (round 2)", d_out);
    run<2>("random bin, 8 copies u32", d_out);
    run<5>("random bin, 16 copies u32", d_out);
    run<9>("random bin, 32 copies u32", d_out);
    run<3>("random bin, 16 copies of packed 16-bit counters", d_out);
    run<4>("random bin, 32 copies of packed 16-bit counters", d_out);
    return 0;
}
