// Batch-1 decode op, "stream" form (round 6): the packed weights go HBM -> LDS by LDS-DMA (buffer_load ... lds) into a per-wave
// ring that is R KiB deep, the dequantisation is done by the MATRIX pipe instead of the vector ALU, and the x row (with its glue) is
// staged ONCE per block.
//
// Replaces the reference hot loop TorchLinear._forward_eager (gptqmodel/nn_modules/qlinear/torch.py:326-347; dequant torch.py:700-717)
// at M = 1, like skinny_kernel (gptqhip_skinny.hip), for 4-bit codes with one group constant per 128-row chunk.  Why a second form
// (VERDICT r5 item 1): skinny_kernel spends 1.8-4 us of every launch on things that are not streaming --
//   * 52 of its ~61 VALU instructions per 1 KiB chunk rebuild the reference's per-weight rounding fp16(s * (q - z));
//   * its bytes in flight live in VGPRs (4 KiB per wave), so it needs 4096 waves per launch (~1.5 us to start them) and a 16-wave
//     LDS reduction at the end;
//   * every ring stage re-loads (and re-glues) its piece of the x row and its group constants with separate VMEM instructions.
// Here:
//   * ALGEBRAIC dequant.  (w & 0x000F000F) | 0x64006400 is the fp16 pair (1024 + q) and (w & 0x00F000F0) | 0x54005400 the pair
//     (64 + q) -- five VALU per packed word (one shift, four and_or), no subtract, no multiply.  The MFMA contracts those raw pairs
//     with x, and the offsets come out again per chunk:  sum_k x_k s (q_k - z) = s * ( sum_k x_k (o_k + q_k) - (c1 + z * c2) )  with
//     c1 = sum_k o_k x_k and c2 = sum_k x_k  computed ONCE per block while the row is staged.  The products are exact (11 x 11 bits
//     into fp32) and the accumulation is fp32, so the result is the exact-arithmetic value up to fp32 rounding; it differs from the
//     reference's chain only by the reference's own per-weight rounding (2^-12 relative, random) -- inside north_star's 1e-3 bar,
//     checked against every golden (tests/test_gpu_stream_decode.py).  bf16 activations: (nibble | 0x4300) = 128 + q (7 VALU per word).
//     The bit-faithful skinny_kernel stays in the library behind GPTQHIP_DECODE_BITFAITHFUL=1 / gptqhip_set_decode_form(0).
//   * 8 waves per block, each with a private ring of R (8) slots of 1 KiB + the chunk's 64-byte constant row, filled by
//     buffer_load_dwordx4 ... lds (nt): 2048 waves per launch, 64 KiB per CU in flight, no VGPRs spent on bytes in flight.
//   * a block works on tiles b, b + grid, ... (the fused gate_up has 7 per CU): the ring runs ahead across tile boundaries, so a
//     tile's reduction / epilogue overlaps the next tile's stream; prologue (RMSNorm statistics, glue, c1 / c2) once per block.
//   * every memory instruction of the main loop is hand-counted (the compiler sees no VMEM result and no LDS read it could order
//     behind the DMAs): s_waitcnt vmcnt(2 * items still in flight), raw s_barrier.
#include <stdlib.h>

#include "gptqhip_device.h"
#include "gptqhip_host.h"

namespace gptqhip {

constexpr int kStreamSlot = 1024;         // one (tile, chunk) block
constexpr int kStreamMaxR = 8;            // ring slots per wave (the planner picks 4 when two blocks share a CU)
constexpr int kStreamMaxWaves = 16;

struct StreamParams {
    const void* x;
    const void* norm_w;
    const uint32_t* qw;
    const uint32_t* meta;
    const void* bias;
    const void* residual;
    const float* stats_in;
    float* stats_out;
    void* out;
    float eps;
    int K, N, G, chunks, tiles, cpg_shift, stats_n;
    int out_glue, out_f32;
    uint32_t qw_bytes, meta_bytes;
    int x_rounds;       // 1 KiB pieces of the x row per wave
    int tiles_per_block;
    int ring_slots;     // R
    // LDS map (byte offsets from the dynamic LDS base)
    int off_nw, off_csum, off_stats, off_ring, off_red, off_epi, off_scr, off_meta;
    int meta_pieces;    // 1 KiB pieces of a tile's [G][16] constant block
#ifdef GPTQHIP_STREAM_STAMPS
    unsigned long long* stamps;
#endif
};

typedef __attribute__((address_space(3))) void* stream_lptr_t;

// dev timing ablations (tests/dev/stream_ablate.py builds one library per value; results wrong by construction; the product build has 0):
// 1 no dequant / MFMA, 2 no x-fragment reads, 4 no ring refill, 8 no weight / constant reads, 16 no reduction / epilogue, 32 no tile barrier
#ifndef GPTQHIP_STREAM_ABLATE
#define GPTQHIP_STREAM_ABLATE 0
#endif
#define ST_ABL(bit) ((GPTQHIP_STREAM_ABLATE & (bit)) != 0)

// dev builds (-DGPTQHIP_STREAM_STAMPS, tests/dev/stream_stamps.py): per-wave phase clocks (s_memtime) into a global buffer whose address
// comes from the environment; the product build compiles none of it
#ifdef GPTQHIP_STREAM_STAMPS
#define ST_STAMP(i) do { if (p.stamps) { st_t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) p.stamps[((size_t)blockIdx.x * 16 + wave) * 16 + (i)] = st_t1; } } while (0)
#define ST_ACCUM(i) do { if (GPTQHIP_STREAM_STAMPS >= 2 && p.stamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t1; st_t1 = t_; } } while (0)
#define ST_FLUSH() do { if (p.stamps && lane == 0) { for (int i_ = 0; i_ < 6; ++i_) p.stamps[((size_t)blockIdx.x * 16 + wave) * 16 + 8 + i_] = st_acc[i_]; } } while (0)
#else
#define ST_STAMP(i) do { } while (0)
#define ST_ACCUM(i) do { } while (0)
#define ST_FLUSH() do { } while (0)
#endif

// ---- hand-counted memory instructions ---------------------------------------------------------------------------------------
__device__ __forceinline__ void st_dma16(__amdgpu_buffer_rsrc_t r, char* lds_dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (stream_lptr_t)lds_dst, 16, voff, soff, 0, 2 /* nt */);
}
__device__ __forceinline__ void st_dma4(__amdgpu_buffer_rsrc_t r, char* lds_dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (stream_lptr_t)lds_dst, 4, voff, soff, 0, 0);
}
__device__ __forceinline__ void st_read128(u4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
template <int OFF>
__device__ __forceinline__ void st_read128o(u4_t& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void st_read64(u2_t& d, uint32_t addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void st_read32(uint32_t& d, uint32_t addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void st_read16(uint32_t& d, uint32_t addr) { asm volatile("ds_read_u16 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void st_write128(uint32_t addr, const u4_t& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void st_write64(uint32_t addr, const u2_t& v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void st_write32(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void st_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void st_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, MAX] (the immediate must be a constant)
template <int MAX>
__device__ __forceinline__ void st_vmwait(int n) {
    if constexpr (MAX > 0) {
        if (n < MAX) {
            st_vmwait<MAX - 1>(n);
            return;
        }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAX) : "memory");
}

// sum over the 16 lanes of a DPP row (every lane of the row gets the total; fixed order)
__device__ __forceinline__ float st_row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ float st_wave_sum(float v) {   // all 64 lanes, fixed order
    v = st_row_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// hipcc (ROCm 7.2) mis-selects element reads of a VECTOR written by inline asm: v[1], v.y, ... all come back as element 0 (reduced case:
// `asm("ds_read_b128 %0, %1" : "=v"(w)); use(w[2])` reads the first register).  Whole-vector uses (MFMA operands) are fine, and so is a
// bit_cast to a struct -- every per-word access to an asm-loaded vector below goes through these.
struct StWords4 { uint32_t w[4]; };
struct StWords2 { uint32_t w[2]; };
__device__ __forceinline__ StWords4 st_words(const u4_t& v) { return __builtin_bit_cast(StWords4, v); }
__device__ __forceinline__ StWords2 st_words(const u2_t& v) { return __builtin_bit_cast(StWords2, v); }

template <int ACT>
__device__ __forceinline__ void st_unpack8(const u4_t& hv, float (&f)[8]) {
    const StWords4 h = st_words(hv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = bits16_to_f32<ACT>((uint16_t)(h.w[j] & 0xffffu));
        f[2 * j + 1] = bits16_to_f32<ACT>((uint16_t)(h.w[j] >> 16));
    }
}

// GLUE: 0 none, 1 RMSNorm (HF LlamaRMSNorm: w * act(h32 * rsqrt(mean(h32^2) + eps)))
template <int ACT, int SCL, int GLUE>
__global__ __launch_bounds__(64 * kStreamMaxWaves) void decode_stream_kernel(StreamParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int R = p.ring_slots;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW = blockDim.x >> 6;
    const int c = lane & 15, rq = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
#ifdef GPTQHIP_STREAM_STAMPS
    unsigned long long st_t1 = 0, st_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
    ST_STAMP(0);

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.qw), 0, (int)p.qw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.meta), 0, (int)p.meta_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.K * 2, 0x00020000);

    // ---- this wave's items: tiles b, b + grid, ... x chunks wave, wave + NW, ... ----------------------------------------------
    const int my_tiles = (p.tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int cpw = wave < p.chunks ? (p.chunks - wave + NW - 1) / NW : 0;   // chunks of a tile this wave owns
    const int nitems = my_tiles * cpw;
    char* const ring = lds + p.off_ring + wave * (R * kStreamSlot);
    const uint32_t ring_a = lds0 + (uint32_t)p.off_ring + (uint32_t)wave * (R * kStreamSlot);
    const uint32_t lane16 = (uint32_t)lane * 16u, lane4 = (uint32_t)lane * 4u, c4 = (uint32_t)c * 4u;

    // issue cursor
    int iq = 0, i_ti = 0, i_ci = 0, i_slot = 0;
    auto issue_item = [&]() __attribute__((always_inline)) {
        const int tile = (int)blockIdx.x + i_ti * (int)gridDim.x;
        const int chunk = wave + i_ci * NW;
        char* dst = ring + i_slot * kStreamSlot;
        st_dma16(rs_w, dst, lane16, (uint32_t)(tile * p.chunks + chunk) * 1024u);
        ++iq;
        i_slot = i_slot + 1 == R ? 0 : i_slot + 1;
        if (++i_ci == cpw) {
            i_ci = 0;
            ++i_ti;
        }
    };

    // A tile's group constants ([G][16] words, contiguous) are staged ONCE per block, double-buffered by tile parity: waves 0 .. pieces-1
    // fetch one 1 KiB piece each while the previous tile is being multiplied and wait for it before that tile's closing barrier.
    // (Per-chunk 64-byte DMAs of the constant row were measured first: an LDS-DMA instruction costs the CU ~40 ns whatever its size, so
    // they halved the stream rate -- profiles/r06_stream_decode.txt.)
    int iq_meta = 0;
    auto issue_meta = [&](int ti) __attribute__((always_inline)) {
        const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
        for (int pc = wave; pc < p.meta_pieces; pc += NW)
            st_dma16(rs_m, lds + p.off_meta + (ti & 1) * (p.meta_pieces * 1024) + pc * 1024, lane16, (uint32_t)(tile * p.G) * 64u + (uint32_t)pc * 1024u);
        iq_meta = iq;
    };

    // ---- prologue: every DMA of the block's small operands first (they are waited for first), then the ring ----------------------
    int pro_ops = 0;   // VMEM ops issued before the ring
    {
        // residual / bias of this block's tiles (16 columns = 32 bytes each; 64 bytes fetched, clamped by the descriptor): wave 0
        if (wave == 0 && (p.residual != nullptr || p.bias != nullptr)) {
            const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, p.residual ? p.N * 2 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bias), 0, p.bias ? p.N * 2 : 0, 0x00020000);
            for (int ti = 0; ti < my_tiles; ++ti) {
                const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
                st_dma4(rs_r, lds + p.off_epi + ti * 512, lane4, (uint32_t)tile * 32u);
                st_dma4(rs_b, lds + p.off_epi + ti * 512 + 256, lane4, (uint32_t)tile * 32u);
            }
        }
        if constexpr (GLUE == 1) {
            if (p.stats_in != nullptr) {   // the producer's per-tile sums of h^2: a private copy per wave (no barrier before the glue)
                const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.stats_in), 0, p.stats_n * 4, 0x00020000);
                // (ONE copy: every wave writes the same bytes to the same place and waits for its own DMA, so no barrier is needed)
                st_dma16(rs_s, lds + p.off_stats, lane16, 0u);
                st_dma16(rs_s, lds + p.off_stats + 1024, lane16, 1024u);
                pro_ops += 2;
            }
            const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.norm_w), 0, p.K * 2, 0x00020000);
            for (int r = 0; r < p.x_rounds; ++r) {
                const uint32_t piece = (uint32_t)(wave * p.x_rounds + r) * 1024u;
                st_dma16(rs_n, lds + p.off_nw + piece, lane16, piece);
            }
            pro_ops += p.x_rounds;
        }
        for (int r = 0; r < p.x_rounds; ++r) {
            const uint32_t piece = (uint32_t)(wave * p.x_rounds + r) * 1024u;
            st_dma16(rs_x, lds + piece, lane16, piece);   // raw row piece, glued in place below (rows past K: zeros)
        }
        pro_ops += p.x_rounds;
    }
    issue_meta(0);
    const int first = nitems < R ? nitems : R;
    for (int i = 0; i < first; ++i) issue_item();
    ST_STAMP(1);
    // everything before the ring has landed once at most 2 * first ops are outstanding (in-order return)
    st_vmwait<kStreamMaxR>(first);
    (void)pro_ops;
    ST_STAMP(2);

    // ---- glue + per-chunk offset sums, once per block -----------------------------------------------------------------------------
    float inv = 1.f;
    if constexpr (GLUE == 1) {
        if (p.stats_in != nullptr) {
            u4_t s0, s1;
            st_read128(s0, lds0 + (uint32_t)p.off_stats + lane16);
            st_read128(s1, lds0 + (uint32_t)p.off_stats + 1024u + lane16);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1));
            float ssum = 0.f;   // (entries past stats_n are outside the descriptor: zeros)
            const StWords4 w0 = st_words(s0), w1 = st_words(s1);
#pragma unroll
            for (int j = 0; j < 4; ++j) ssum += __builtin_bit_cast(float, w0.w[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) ssum += __builtin_bit_cast(float, w1.w[j]);
            ssum = st_wave_sum(ssum);
            inv = rsqrtf(ssum / (float)p.K + p.eps);
        } else {
            // no producer statistics (the first op of a step): reduce the row in the block, fixed order
            float ss = 0.f;
            for (int r = 0; r < p.x_rounds; ++r) {
                u4_t h;
                st_read128(h, lds0 + (uint32_t)(wave * p.x_rounds + r) * 1024u + lane16);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h));
                float f[8];
                st_unpack8<ACT>(h, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(f[j], f[j], ss);
            }
            ss = st_wave_sum(ss);
            if (lane == 0) st_write32(lds0 + (uint32_t)p.off_scr + (uint32_t)wave * 4u, __builtin_bit_cast(uint32_t, ss));
            st_barrier();
            float tot = 0.f;
            for (int w = 0; w < NW; ++w) {
                uint32_t t;
                st_read32(t, lds0 + (uint32_t)p.off_scr + (uint32_t)w * 4u);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t));
                tot += __builtin_bit_cast(float, t);
            }
            inv = rsqrtf(tot / (float)p.K + p.eps);
        }
    }
    for (int r = 0; r < p.x_rounds; ++r) {
        const uint32_t piece = (uint32_t)(wave * p.x_rounds + r) * 1024u;
        u4_t h, g = {0u, 0u, 0u, 0u};
        st_read128(h, lds0 + piece + lane16);
        if constexpr (GLUE == 1) st_read128(g, lds0 + (uint32_t)p.off_nw + piece + lane16);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h), "+v"(g));
        float f[8];
        if constexpr (GLUE == 1) {
            float w8[8];
            st_unpack8<ACT>(h, f);
            st_unpack8<ACT>(g, w8);
            u4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // HF: weight * hidden.to(dtype): the normalised value is rounded to the activation dtype, the product once more
                const float a0 = round_through<ACT>(w8[2 * j] * round_through<ACT>(f[2 * j] * inv));
                const float a1 = round_through<ACT>(w8[2 * j + 1] * round_through<ACT>(f[2 * j + 1] * inv));
                f[2 * j] = a0;
                f[2 * j + 1] = a1;
                o[j] = (uint32_t)f32_to_16<ACT>(a0) | ((uint32_t)f32_to_16<ACT>(a1) << 16);
            }
            st_write128(lds0 + piece + lane16, o);
        } else {
            st_unpack8<ACT>(h, f);
        }
        // element e of this lane's 16 bytes is row 8 * (lane & 3 ...) -- precisely: k = piece / 2 + 8 * lane + e, i.e. K-step position
        // e of an rq slice; codes e in {0, 1, 4, 5} carry offset o_lo, codes {2, 3, 6, 7} offset o_hi (tiled_shift4)
        constexpr float o_lo = ACT == kFP16 ? 1024.f : 128.f, o_hi = ACT == kFP16 ? 64.f : 128.f;
        float c2 = ((f[0] + f[1]) + (f[4] + f[5])), c2h = ((f[2] + f[3]) + (f[6] + f[7]));
        float c1 = o_lo * c2 + o_hi * c2h;
        c2 += c2h;
        c1 = st_row_sum(c1);     // 16 lanes x 8 elements = one 128-row chunk
        c2 = st_row_sum(c2);
        if (c == 0) {
            u2_t cs = {__builtin_bit_cast(uint32_t, c1), __builtin_bit_cast(uint32_t, c2)};
            st_write64(lds0 + (uint32_t)p.off_csum + (piece / 256u + (uint32_t)rq) * 8u, cs);
        }
    }
    st_barrier();
    ST_STAMP(3);

    // ---- main loop ------------------------------------------------------------------------------------------------------------------
    const uint32_t LO = 0x000F000Fu, HI = 0x00F000F0u;
    uint32_t magic_lo = ACT == kFP16 ? 0x64006400u : 0x43004300u, magic_hi = 0x54005400u;
    asm volatile("" : "+v"(magic_lo));
    asm volatile("" : "+v"(magic_hi));
    float acc = 0.f;
    int c_ti = 0, c_ci = 0, c_slot = 0;
    // (every kernel argument the loop touches is consumed here once, so hipcc retires its scalar loads BEFORE the loop instead of
    // placing an s_waitcnt lgkmcnt(0) between the hand-issued LDS reads inside it)
    asm volatile("" ::"s"(p.out), "s"(p.stats_out), "s"(p.N), "s"(p.out_glue), "s"(p.out_f32), "s"(p.bias), "s"(p.residual), "s"(p.off_red), "s"(p.off_epi),
                 "s"(p.off_csum));
    uint32_t meta_a = lds0 + (uint32_t)p.off_meta;      // the current tile's constants
    struct Item {
        u4_t wv, a0, a1, a2, a3;
        uint32_t mw;
        u2_t cs;
    };
    auto read_item = [&](Item& it, int slot, int chunk) __attribute__((always_inline)) {
        const uint32_t sa = ring_a + (uint32_t)slot * kStreamSlot;
        const uint32_t xa = lds0 + (uint32_t)chunk * 256u + (uint32_t)rq * 16u;
        const uint32_t ca = lds0 + (uint32_t)p.off_csum + (uint32_t)chunk * 8u;
        asm volatile("" ::"v"(xa), "v"(ca));
        if (!ST_ABL(8)) {
        st_read128(it.wv, sa + lane16);
        st_read32(it.mw, meta_a + (uint32_t)(chunk >> p.cpg_shift) * 64u + c4);
        }
        if (!ST_ABL(2)) {
        st_read128o<0>(it.a0, xa);
        st_read128o<64>(it.a1, xa);
        st_read128o<128>(it.a2, xa);
        st_read128o<192>(it.a3, xa);
        }
        st_read64(it.cs, ca);
    };
    auto own_item = [&](Item& it) __attribute__((always_inline)) {   // makes every consumer of the item depend on the wait before it
        asm volatile("" : "+v"(it.wv), "+v"(it.mw), "+v"(it.a0), "+v"(it.a1), "+v"(it.a2), "+v"(it.a3), "+v"(it.cs));
    };
    auto bfrag = [&](uint32_t w) __attribute__((always_inline)) {
        u4_t b;
        if constexpr (ACT == kFP16) {
            const uint32_t w8 = w >> 8;
            b.x = (w & LO) | magic_lo;
            b.y = (w & HI) | magic_hi;
            b.z = (w8 & LO) | magic_lo;
            b.w = (w8 & HI) | magic_hi;
        } else {
            b.x = (w & LO) | magic_lo;
            b.y = ((w >> 4) & LO) | magic_lo;
            b.z = ((w >> 8) & LO) | magic_lo;
            b.w = ((w >> 12) & LO) | magic_lo;
        }
        return b;
    };
    // raw code pairs straight into the matrix pipe; offsets, zero-point and scale come out per chunk in fp32
    auto compute_item = [&](const Item& it, float a_in) __attribute__((always_inline)) {
        f4_t ga = {0.f, 0.f, 0.f, 0.f}, gb = {0.f, 0.f, 0.f, 0.f};
        const StWords4 ww = st_words(it.wv);
        ga = mfma16<ACT>(it.a0, bfrag(ww.w[0]), ga);
        gb = mfma16<ACT>(it.a1, bfrag(ww.w[1]), gb);
        ga = mfma16<ACT>(it.a2, bfrag(ww.w[2]), ga);
        gb = mfma16<ACT>(it.a3, bfrag(ww.w[3]), gb);
        const float sc = bits16_to_f32<SCL>((uint16_t)(it.mw & 0xffffu));
        const float z = (float)((it.mw >> 16) & 0xFu);
        const StWords2 cw = st_words(it.cs);
        const float c1 = __builtin_bit_cast(float, cw.w[0]), c2 = __builtin_bit_cast(float, cw.w[1]);
        return __builtin_fmaf(sc, (ga[0] + gb[0]) - __builtin_fmaf(z, c2, c1), a_in);
    };
    if (my_tiles > 1) issue_meta(1);
    for (int q = 0; q < nitems;) {
        // two chunks of the tile per iteration (their LDS reads share one wait, their dequant / MFMA chains interleave); a lone last chunk
        // of an odd count runs alone
        const int u = cpw - c_ci >= 2 ? 2 : 1;
        const int chunk = wave + c_ci * NW;
        st_vmwait<kStreamMaxR - 1>(iq - (q + u - 1) - 1);
        ST_ACCUM(0);
        Item A, B;
        const int slot1 = c_slot + 1 == R ? 0 : c_slot + 1;
        if (u == 2) {
            read_item(A, c_slot, chunk);
            read_item(B, slot1, chunk + NW);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            own_item(A);
            own_item(B);
            ST_ACCUM(1);
            if (iq < nitems && !ST_ABL(4)) issue_item();     // the slots are free: their words are in registers
            if (iq < nitems && !ST_ABL(4)) issue_item();
            ST_ACCUM(2);
            if (ST_ABL(4)) iq = q + R + 2 < nitems ? q + R + 2 : nitems;
            if (!ST_ABL(1)) {
            // (hand-interleaved: hipcc keeps two compute_item calls back to back, each with its own dependent MFMA pair and the
            // wait states behind it; here the four chains alternate, so a dependent MFMA issues three MFMAs after its producer)
            const StWords4 wa = st_words(A.wv), wb = st_words(B.wv);
            const u4_t ba0 = bfrag(wa.w[0]), bb0 = bfrag(wb.w[0]), ba1 = bfrag(wa.w[1]), bb1 = bfrag(wb.w[1]);
            const u4_t ba2 = bfrag(wa.w[2]), bb2 = bfrag(wb.w[2]), ba3 = bfrag(wa.w[3]), bb3 = bfrag(wb.w[3]);
            const f4_t zero4 = {0.f, 0.f, 0.f, 0.f};
            f4_t gaA = mfma16<ACT>(A.a0, ba0, zero4);
            f4_t gaB = mfma16<ACT>(B.a0, bb0, zero4);
            f4_t gbA = mfma16<ACT>(A.a1, ba1, zero4);
            f4_t gbB = mfma16<ACT>(B.a1, bb1, zero4);
            gaA = mfma16<ACT>(A.a2, ba2, gaA);
            gaB = mfma16<ACT>(B.a2, bb2, gaB);
            gbA = mfma16<ACT>(A.a3, ba3, gbA);
            gbB = mfma16<ACT>(B.a3, bb3, gbB);
            const StWords2 ca_ = st_words(A.cs), cb_ = st_words(B.cs);
            const float sA = bits16_to_f32<SCL>((uint16_t)(A.mw & 0xffffu)), sB = bits16_to_f32<SCL>((uint16_t)(B.mw & 0xffffu));
            const float tA = __builtin_fmaf((float)((A.mw >> 16) & 0xFu), __builtin_bit_cast(float, ca_.w[1]), __builtin_bit_cast(float, ca_.w[0]));
            const float tB = __builtin_fmaf((float)((B.mw >> 16) & 0xFu), __builtin_bit_cast(float, cb_.w[1]), __builtin_bit_cast(float, cb_.w[0]));
            acc = __builtin_fmaf(sA, (gaA[0] + gbA[0]) - tA, acc);
            acc = __builtin_fmaf(sB, (gaB[0] + gbB[0]) - tB, acc);
            } else {
                acc += __builtin_bit_cast(float, st_words(A.wv).w[0] ^ st_words(B.wv).w[1] ^ A.mw ^ st_words(A.a0).w[0] ^ st_words(B.a3).w[0] ^ st_words(B.cs).w[0]);
            }
#ifdef GPTQHIP_STREAM_STAMPS
            asm volatile("" : "+v"(acc));
#endif
            ST_ACCUM(3);
        } else {
            read_item(A, c_slot, chunk);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            own_item(A);
            if (iq < nitems) issue_item();
            acc = compute_item(A, acc);
        }
        q += u;
        c_slot = c_slot + u >= R ? c_slot + u - R : c_slot + u;
        c_ci += u;
        if (c_ci == cpw) {
            // ---- tile done: in-block split-K reduction (double-buffered by tile parity), epilogue by wave (tile index % NW) ----
            const int tile = (int)blockIdx.x + c_ti * (int)gridDim.x;
            const uint32_t red = lds0 + (uint32_t)p.off_red + (uint32_t)(c_ti & 1) * (kStreamMaxWaves * 64u);
            st_write32(red + (uint32_t)wave * 64u + c4, __builtin_bit_cast(uint32_t, acc));   // (the four lane quads hold the same 16 sums)
            acc = 0.f;
            if (c_ti + 1 < my_tiles && !ST_ABL(4)) st_vmwait<kStreamMaxR>(iq - iq_meta);   // this wave's piece of the next tile's constants has landed
            if (!ST_ABL(32)) st_barrier();
            ST_ACCUM(4);
            if (wave == c_ti % NW && !ST_ABL(16)) {
                float v = 0.f;
                for (int w0 = 0; w0 < NW; w0 += 4) {   // fixed order w = 0, 1, ...; four reads per wait
                    uint32_t t[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) st_read32(t[i], red + (uint32_t)(w0 + i < NW ? w0 + i : NW - 1) * 64u + c4);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
#pragma unroll
                    for (int i = 0; i < 4; ++i) v += (w0 + i < NW) ? __builtin_bit_cast(float, t[i]) : 0.f;
                }
                const int n = tile * kTileN + c;
                const bool live = lane < 16 && n < p.N;
                const uint32_t epi = lds0 + (uint32_t)p.off_epi + (uint32_t)c_ti * 512u + (uint32_t)c * 2u;
                if (p.out_f32) {
                    if (live) reinterpret_cast<float*>(p.out)[n] = v;
                } else {
                    float y = round_through<ACT>(v);
                    if (p.bias != nullptr) {
                        uint32_t b;
                        st_read16(b, epi + 256u);
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b));
                        y = round_through<ACT>(y + bits16_to_f32<ACT>((uint16_t)b));
                    }
                    if (p.out_glue == 1) {
                        // interleaved gate|up tile (fuse_gate_up_interleaved): lanes 0..7 gate columns, lanes 8..15 the matching up columns
                        const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x108 /* row_shl:8 */, 0xf, 0xf, false));
                        const float a = round_through<ACT>(y / (1.0f + expf(-y))) * up;
                        const int j = tile * 8 + c;
                        if (live && c < 8 && j < p.N / 2) reinterpret_cast<uint16_t*>(p.out)[j] = f32_to_16<ACT>(a);
                    } else {
                        if (p.residual != nullptr) {
                            uint32_t rr;
                            st_read16(rr, epi);
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr));
                            y = bits16_to_f32<ACT>((uint16_t)rr) + y;
                        }
                        const float h = round_through<ACT>(y);
                        if (live) reinterpret_cast<uint16_t*>(p.out)[n] = f32_to_16<ACT>(h);
                        if (p.stats_out != nullptr) {
                            const float sq = st_row_sum(live ? h * h : 0.f);
                            if (lane == 0) p.stats_out[tile] = sq;
                        }
                    }
                }
            }
            c_ci = 0;
            ++c_ti;
            meta_a = lds0 + (uint32_t)p.off_meta + (uint32_t)(c_ti & 1) * (uint32_t)(p.meta_pieces * 1024);
            if (c_ti + 1 < my_tiles) issue_meta(c_ti + 1);   // (its buffer held tile c_ti - 1, whose reads all finished before the barrier above)
            ST_ACCUM(5);
        }
    }
    ST_FLUSH();
    ST_STAMP(4);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------------
StreamPlan plan_stream(int K, int N, int group_size, int bits, int force_waves, bool with_norm, bool with_epi) {
    StreamPlan pl;
    pl.ok = 0;
    if (bits != 4 || K % kChunkK != 0 || group_size % kChunkK != 0 || K % group_size != 0) return pl;
    const int cpg = group_size >= K ? K / kChunkK : group_size / kChunkK;
    if (group_size < K && (cpg & (cpg - 1)) != 0) return pl;
    pl.chunks = K / kChunkK;
    pl.tiles = ceil_div(N, kTileN);
    if ((size_t)pl.tiles * pl.chunks * 1024 >= ((size_t)1 << 31) || (size_t)pl.tiles * (K / group_size) * 64 >= ((size_t)1 << 31)) return pl;
    static const int env_waves = [] { const char* v = getenv("GPTQHIP_STREAM_WAVES"); return (v && *v) ? atoi(v) : 0; }();
    static const int env_grid = [] { const char* v = getenv("GPTQHIP_STREAM_GRID"); return (v && *v) ? atoi(v) : 0; }();
    int waves = force_waves > 0 ? force_waves : (env_waves > 0 ? env_waves : 8);
    if (waves > kStreamMaxWaves) waves = kStreamMaxWaves;
    if (waves > pl.chunks) waves = pl.chunks;
    pl.waves = waves;
    // one block per CU working on tiles b, b + grid, ... once the layer has more tiles than two rounds of CUs; otherwise one tile per block
    const int cus = 256;
    int grid = pl.tiles;
    if (pl.tiles >= 2 * cus) grid = pl.tiles % cus == 0 ? cus : (pl.tiles % (2 * cus) == 0 ? 2 * cus : cus);
    if (env_grid > 0 && env_grid < pl.tiles) grid = env_grid;
    pl.grid = grid;
    pl.tiles_per_block = ceil_div(pl.tiles, grid);
    pl.x_rounds = ceil_div(K * 2, 1024 * waves);
    // ring depth: 8 KiB in flight per wave with one block per CU; 4 KiB when the grid needs two co-resident blocks per CU (then the map
    // must stay under 80 KiB)
    static const int env_r = [] { const char* v = getenv("GPTQHIP_STREAM_RING"); return (v && *v) ? atoi(v) : 0; }();
    pl.ring_slots = grid > cus ? 4 : kStreamMaxR;
    while (pl.ring_slots > 2 && waves * pl.ring_slots > 64) pl.ring_slots >>= 1;
    pl.meta_pieces = ceil_div((K / group_size) * 64, 1024);
    if (env_r >= 2 && env_r <= kStreamMaxR) pl.ring_slots = env_r;
    const int x_bytes = pl.x_rounds * waves * 1024;
    int off = x_bytes;
    pl.off_nw = off;        off += with_norm ? x_bytes : 0;
    pl.off_csum = off;      off += ((x_bytes / 256) * 8 + 15) & ~15;
    pl.off_stats = off;     off += with_norm ? 2048 : 0;
    pl.off_ring = off;      off += waves * pl.ring_slots * kStreamSlot;
    pl.off_red = off;       off += 2 * kStreamMaxWaves * 64;
    pl.off_epi = off;       off += with_epi ? pl.tiles_per_block * 512 : 0;
    pl.off_scr = off;       off += 256;
    pl.off_meta = off;      off += 2 * pl.meta_pieces * 1024;
    pl.lds_bytes = off;
    if (off > 160 * 1024) return pl;
    pl.ok = 1;
    return pl;
}

template <int ACT, int SCL>
static int launch_stream_glue(const StreamParams& p, const StreamPlan& pl, int in_glue, hipStream_t stream) {
    const dim3 grid(pl.grid), block(64 * pl.waves);
    auto go = [&](auto kern) {
        static bool attr_done = false;   // (per instantiation: the lambda's static is per template argument)
        if (!attr_done) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return check_hip(e, "decode_stream_kernel: hipFuncSetAttribute");
            attr_done = true;
        }
        hipLaunchKernelGGL(kern, grid, block, (size_t)pl.lds_bytes, stream, p);
        return check_hip(hipGetLastError(), "decode_stream_kernel launch");
    };
    if (in_glue == 1) return go(decode_stream_kernel<ACT, SCL, 1>);
    return go(decode_stream_kernel<ACT, SCL, 0>);
}

int launch_stream(const GemmArgs& a, const StreamPlan& pl, hipStream_t stream) {
    StreamParams p;
    p.x = a.x;
    p.norm_w = a.glue_b;
    p.qw = a.qweight;
    p.meta = a.meta;
    p.bias = a.bias;
    p.residual = a.residual;
    p.stats_in = a.stats_in;
    p.stats_out = a.stats_out;
    p.out = a.out;
    p.eps = a.eps;
    p.K = a.K;
    p.N = a.N;
    p.G = a.K / a.group_size;
    p.chunks = pl.chunks;
    p.tiles = pl.tiles;
    {
        const int cpg = a.group_size >= a.K ? 1 << 20 : a.group_size / kChunkK;
        int sh = 0;
        while ((1 << sh) < cpg) ++sh;
        p.cpg_shift = sh;
    }
    p.stats_n = a.stats_n;
    p.out_glue = a.out_glue;
    p.out_f32 = a.out_f32;
    p.qw_bytes = (uint32_t)((size_t)pl.tiles * pl.chunks * 1024);
    p.meta_bytes = (uint32_t)((size_t)pl.tiles * p.G * 64);
    p.x_rounds = pl.x_rounds;
    p.tiles_per_block = pl.tiles_per_block;
    p.ring_slots = pl.ring_slots;
    p.off_nw = pl.off_nw;
    p.off_csum = pl.off_csum;
    p.off_stats = pl.off_stats;
    p.off_ring = pl.off_ring;
    p.off_red = pl.off_red;
    p.off_epi = pl.off_epi;
    p.off_scr = pl.off_scr;
    p.off_meta = pl.off_meta;
    p.meta_pieces = pl.meta_pieces;
#ifdef GPTQHIP_STREAM_STAMPS
    {
        const char* v = getenv("GPTQHIP_STREAM_STAMPS_PTR");
        p.stamps = (v && *v) ? reinterpret_cast<unsigned long long*>(strtoull(v, nullptr, 0)) : nullptr;
    }
#endif
    if (a.act_dtype == kFP16 && a.scale_dtype == kFP16) return launch_stream_glue<kFP16, kFP16>(p, pl, a.in_glue, stream);
    if (a.act_dtype == kBF16 && a.scale_dtype == kFP16) return launch_stream_glue<kBF16, kFP16>(p, pl, a.in_glue, stream);
    if (a.act_dtype == kFP16 && a.scale_dtype == kBF16) return launch_stream_glue<kFP16, kBF16>(p, pl, a.in_glue, stream);
    return launch_stream_glue<kBF16, kBF16>(p, pl, a.in_glue, stream);
}

}  // namespace gptqhip
