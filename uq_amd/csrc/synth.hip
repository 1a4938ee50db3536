// synth.hip -- "synth-v1" synthetic FASTQ generated directly in HBM.
// Byte-identical to uq_amd/synth.py: every character is a pure function of (seed, read index, slot).
// Workload generation for tests and bench.py; not part of the encode path.
#include "common.h"

namespace {
constexpr uint64_t TEMPLATE_BASE = uint64_t(1) << 29;
constexpr uint32_t SLOT_QUAL = 512, SLOT_X = 1020, SLOT_Y = 1021, SLOT_LEN = 1022, SLOT_CTL = 1023;
constexpr uint32_t PREFIX_LEN = 17;
__device__ __constant__ char kPrefix[PREFIX_LEN + 1] = "@SIM001:42:FCX01:";

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t value(uint64_t seed, uint64_t i, uint32_t s) {
    return splitmix64((seed << 40) + i * 1024 + s);
}
__device__ __forceinline__ uint32_t ndigits(uint32_t v) {
    return 1 + (v >= 10) + (v >= 100) + (v >= 1000) + (v >= 10000);
}

struct Rec {
    uint32_t L, x, y, xd, yd, hdr, size;
    uint64_t src_d, src_q;
};

__device__ __forceinline__ void rec_info(const uq_synth_spec& sp, uint64_t i, Rec& r) {
    if (sp.len_lo == sp.len_hi) r.L = sp.len_lo;
    else {
        uint32_t span = (uint32_t)(sp.len_hi - sp.len_lo + 1);
        r.L = sp.len_lo + (uint32_t)(value(sp.seed, i, SLOT_LEN) % span);
        if (sp.skip_len_mod4 && (r.L & 3) == 0) r.L = (r.L + 1 <= (uint32_t)sp.len_hi) ? r.L + 1 : r.L - 1;
    }
    r.x = 1000 + (uint32_t)(value(sp.seed, i, SLOT_X) % 29000);
    r.y = 1000 + (uint32_t)(value(sp.seed, i, SLOT_Y) % 29000);
    r.xd = ndigits(r.x); r.yd = ndigits(r.y);
    r.hdr = PREFIX_LEN + 1 + 1 + 4 + 1 + r.xd + 1 + r.yd + 1;
    r.size = r.hdr + r.L + 1 + 2 + r.L + 1;
    r.src_d = i; r.src_q = i;
    if (sp.dup) {
        uint64_t ctl = value(sp.seed, i, SLOT_CTL);
        if (ctl % 10 == 0) {
            uint64_t t = TEMPLATE_BASE + (ctl / 10) % (uint64_t)sp.dup_templates;
            if (sp.dup & 1) r.src_d = t;
            if (sp.dup & 2) r.src_q = t;
        }
    }
}

__device__ __forceinline__ uint8_t digit_of(uint32_t v, uint32_t nd, uint32_t k) {   // k-th digit from the left
    uint32_t p = 1;
    for (uint32_t i = 0; i + k + 1 < nd; ++i) p *= 10;
    return (uint8_t)('0' + (v / p) % 10);
}

__device__ __forceinline__ uint8_t rec_byte(const uq_synth_spec& sp, uint64_t i, const Rec& r, uint32_t p) {
    if (p < r.hdr) {
        if (p < PREFIX_LEN) return (uint8_t)kPrefix[p];
        p -= PREFIX_LEN;
        if (p == 0) return (uint8_t)('1' + (uint32_t)(i % 4));
        if (p == 1) return ':';
        if (p < 6) return digit_of(1101 + (uint32_t)(i % 64), 4, p - 2);
        if (p == 6) return ':';
        p -= 7;
        if (p < r.xd) return digit_of(r.x, r.xd, p);
        if (p == r.xd) return ':';
        p -= r.xd + 1;
        if (p < r.yd) return digit_of(r.y, r.yd, p);
        return '\n';
    }
    p -= r.hdr;
    if (p < r.L) {
        uint64_t vb = value(sp.seed, r.src_d, p);
        if (sp.n_rate > 0 && ((vb >> 8) % 100) < (uint64_t)sp.n_rate) return 'N';
        return (uint8_t)"ACGT"[vb & 3];
    }
    if (p == r.L) return '\n';
    if (p == r.L + 1) return '+';
    if (p == r.L + 2) return '\n';
    p -= r.L + 3;
    if (p < r.L) {
        uint64_t vq = value(sp.seed, r.src_q, SLOT_QUAL + p);
        if (sp.n_rate > 0) {
            uint64_t vb = value(sp.seed, r.src_d, p);
            bool isn = ((vb >> 8) % 100) < (uint64_t)sp.n_rate;
            if (sp.n_qual_exclusive) return isn ? (uint8_t)33 : (uint8_t)(34 + vq % 40);
            return isn ? (uint8_t)35 : (uint8_t)(33 + vq % 41);
        }
        return (uint8_t)(33 + vq % 41);
    }
    return '\n';
}

__global__ void synth_sizes_kernel(uq_synth_spec sp, uint64_t first, uint64_t n, uint64_t* __restrict__ sizes) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Rec r;
    rec_info(sp, first + k, r);
    sizes[k] = r.size;
}

__global__ __launch_bounds__(256) void synth_write_kernel(uq_synth_spec sp, uint64_t first, uint64_t n,
                                                          const uint64_t* __restrict__ offsets, uint8_t* __restrict__ out,
                                                          uint64_t capacity) {
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t GW = (uint64_t)gridDim.x * 4;
    const uint32_t lane = lane_id();
    for (uint64_t k = gw; k < n; k += GW) {
        Rec r;
        rec_info(sp, first + k, r);
        const uint64_t off = offsets[k];
        if (off + r.size > capacity) continue;
        for (uint32_t p = lane; p < r.size; p += 64) out[off + p] = rec_byte(sp, first + k, r, p);
    }
}

int check_spec(const uq_synth_spec* s) {
    UQ_REQUIRE(s, "null synth spec");
    UQ_REQUIRE(s->len_lo >= 1 && s->len_lo <= s->len_hi && s->len_hi <= 508, "synth: read length out of range");
    UQ_REQUIRE(s->dup >= 0 && s->dup <= 3 && (s->dup == 0 || s->dup_templates > 0), "synth: bad dup settings");
    return 0;
}
}  // namespace

extern "C" int uq_synth_size(uq_ctx* ctx, const uq_synth_spec* h_spec, uint64_t first, uint64_t n, uint64_t* h_bytes) {
    UQ_REQUIRE(ctx && h_bytes, "uq_synth_size: null argument");
    UQ_TRY(check_spec(h_spec));
    if (n == 0) { *h_bytes = 0; return 0; }
    void* scr;
    UQ_TRY(uq_scratch(ctx, (n + 1) * 8 + 256, &scr));
    uint64_t* sizes = (uint64_t*)scr;
    synth_sizes_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, ctx->stream>>>(*h_spec, first, n, sizes);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_scan_exclusive_u64(ctx, sizes, sizes, n, sizes + n));
    UQ_CHECK_HIP(hipMemcpyAsync(ctx->h_pinned, sizes + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    UQ_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *h_bytes = ctx->h_pinned[0];
    return 0;
}

extern "C" int uq_synth_fastq(uq_ctx* ctx, const uq_synth_spec* h_spec, uint64_t first, uint64_t n, uint8_t* d_out, uint64_t capacity) {
    UQ_REQUIRE(ctx && d_out, "uq_synth_fastq: null argument");
    UQ_TRY(check_spec(h_spec));
    if (n == 0) return 0;
    void* scr;
    UQ_TRY(uq_scratch(ctx, (n + 1) * 8 + 256, &scr));
    uint64_t* sizes = (uint64_t*)scr;
    synth_sizes_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, ctx->stream>>>(*h_spec, first, n, sizes);
    UQ_LAUNCH_CHECK();
    UQ_TRY(uq_scan_exclusive_u64(ctx, sizes, sizes, n, sizes + n));
    uint64_t blocks = (n + 3) / 4;
    if (blocks > UQ_NUM_CU * 16) blocks = UQ_NUM_CU * 16;
    synth_write_kernel<<<(uint32_t)blocks, 256, 0, ctx->stream>>>(*h_spec, first, n, sizes, d_out, capacity);
    UQ_LAUNCH_CHECK();
    return 0;
}
