// Device-side building blocks shared by the gfx950 kernels: vector types, the tile-major weight layout,
// the int4/int8 -> fp16/bf16 dequantisation that reproduces the reference's rounding bit-for-bit, and
// MFMA wrappers.
//
// Rounding contract being reproduced (reference gptqmodel/nn_modules/qlinear/torch.py:716-717,
// torch_awq.py:149-155, packing_utils.py:117-119):
//     W  = round_to(scales.dtype)( float(scale) * (code - zero) )         (one rounding)
//     W' = round_to(x.dtype)(W)                                           (only if dtypes differ)
// (code - zero) is a small exact integer.  We get it without any int->float convert: OR the nibble
// into the mantissa of 1024.0h (0x6400 | q  ==  1024 + q exactly) and add the precomputed constant
// -(1024 + zero); for the nibbles sitting 4 bits higher (0x6400 | q<<4 == 1024 + 16q) one fma by 1/16 with
// -(64 + zero) gives the same exact integer.  The fp16 multiply by the scale is then the single
// correctly-rounded product torch computes.
//
// ------------------------------------------------------------------------------------------------
// TILE-MAJOR LAYOUT ("tiled"), produced once in post_init by gptqhip_repack_tiled:
//   Kp = ceil(K/128)*128, Np = ceil(N/16)*16, tiles = Np/16, chunks = Kp/128.
//   A (tile, chunk) block is the B operand of four mfma_f32_16x16x32 K-steps: 16 columns x 128 rows.
//   4-bit:  word[((tile*chunks + chunk)*64 + lane)*4 + j]          lane = (rq<<4)|c, rq=0..3, c=0..15, j=0..3
//           holds column n = 16*tile + c, rows k = 128*chunk + 32*j + 8*rq + e (e=0..7), code e at bit
//           offset 4*(e>>1) + 16*(e&1):   [k0|k2|k4|k6 | k1|k3|k5|k7]  (low nibble first)
//           => (w & 0x000F000F) = (k0,k1), (w & 0x00F000F0) = (k2,k3)<<4, same on w>>8 for k4..k7:
//           the 8 halves come out in NATURAL k order, so activations need no permutation.
//   8-bit:  word[(((tile*chunks + chunk)*2 + h)*64 + lane)*4 + jj]  K-step j = 2h + (jj>>1), half = jj&1
//           rows k = 128*chunk + 32*j + 8*rq + 4*half + e (e=0..3), code e at bit offset 8*(e>>1) + 16*(e&1).
//   One wave-instruction (dwordx4 per lane) therefore reads 1 KiB CONTIGUOUS bytes = a whole (tile, chunk)
//   block, and consecutive chunks of a tile are contiguous: every wave streams a linear address range.
//   meta[(tile*G + g)*16 + c] = scale16 | (0xE400|zero)<<16   -- the per-(group, column) constants, pre-baked
//   (scale bits in the scales dtype; -(1024+zero) as fp16).  Padded rows (k >= K) carry code == zero-point of the
//   last group (they dequantise to exactly 0); padded columns carry code 0 / scale 0 and are never stored.
// ------------------------------------------------------------------------------------------------
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gptqhip {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));

constexpr int kFP16 = 0;
constexpr int kBF16 = 1;
constexpr int kChunkK = 128;  // rows per (tile, chunk) block
constexpr int kTileN = 16;    // columns per tile

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// bit offset of code e (0..pf-1) inside a tiled word
__host__ __device__ inline int tiled_shift4(int e) { return 4 * (e >> 1) + 16 * (e & 1); }
__host__ __device__ inline int tiled_shift8(int e) { return 8 * (e >> 1) + 16 * (e & 1); }

__device__ __forceinline__ h2_t as_h2(uint32_t u) { return __builtin_bit_cast(h2_t, u); }
__device__ __forceinline__ uint32_t as_u32(h2_t h) { return __builtin_bit_cast(uint32_t, h); }

// fp32 -> bf16 bits, round-to-nearest-even (the compiler selects v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    bf2_t r;
    r.x = (__bf16)lo;
    r.y = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
    h2_t r;
    r.x = (_Float16)lo;
    r.y = (_Float16)hi;
    return as_u32(r);
}
__device__ __forceinline__ float bf16lo_to_f32(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// 16-bit scalar helpers ------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ float bits16_to_f32(uint16_t v) {
    if constexpr (DT == kFP16) {
        return (float)__builtin_bit_cast(_Float16, v);
    } else {
        return __builtin_bit_cast(float, (uint32_t)v << 16);
    }
}
template <int DT>
__device__ __forceinline__ float load16_as_f32(const void* p, size_t idx) {
    return bits16_to_f32<DT>(reinterpret_cast<const uint16_t*>(p)[idx]);
}
template <int DT>
__device__ __forceinline__ uint16_t f32_to_16(float f) {
    if constexpr (DT == kFP16) {
        return __builtin_bit_cast(uint16_t, (_Float16)f);
    } else {
        return __builtin_bit_cast(uint16_t, (__bf16)f);
    }
}
template <int DT>
__device__ __forceinline__ float round_through(float f) {  // round fp32 to DT and back
    if constexpr (DT == kFP16) {
        return (float)(_Float16)f;
    } else {
        return (float)(__bf16)f;
    }
}

// Per-(group, column) dequant constants expanded from one meta word.
struct ColConst {
    uint32_t s;    // SCL fp16: half2(scale,scale);  SCL bf16: fp32 scale bits
    uint32_t zlo;  // half2(-(1024+z), -(1024+z))
    uint32_t zhi;  // half2(-(64+z),   -(64+z))      (4-bit high nibbles only)
    float nzs;     // SCL bf16 only: -zero * scale (exact in fp32: 8 x 8 significant bits)
    float s16;     // SCL bf16, 4-bit: scale / 16
    float clo;     //                  -(1024 + zero) * scale        (exact: 11 x 8 significant bits)
    float chi;     //                  -(64 + zero) * scale  ==  -(1024 + 16 zero) * (scale / 16)
};

template <int BITS, int SCL>
__device__ __forceinline__ ColConst expand_meta(uint32_t meta) {
    ColConst c;
    const uint32_t sb = meta & 0xffffu;
    if constexpr (SCL == kFP16) {
        c.s = sb | (sb << 16);
    } else {
        c.s = sb << 16;
    }
    c.nzs = SCL == kBF16 ? -(float)((meta >> 16) & 0x3FFu) * __builtin_bit_cast(float, sb << 16) : 0.f;
    if constexpr (SCL == kBF16 && BITS == 4) {
        const float sf = __builtin_bit_cast(float, sb << 16);
        const float zf = (float)((meta >> 16) & 0xFu);
        c.s16 = sf * 0.0625f;
        c.clo = -(1024.f + zf) * sf;
        c.chi = -(64.f + zf) * sf;
    } else {
        c.s16 = c.clo = c.chi = 0.f;
    }
    const uint32_t zc = meta >> 16;  // 0xE400 | zero
    c.zlo = zc | (zc << 16);
    if constexpr (BITS == 4) {
        const uint32_t zh = 0xD400u | ((zc & 0xFu) << 4);  // -(64 + zero): ulp(64) = 1/16
        c.zhi = zh | (zh << 16);
    } else {
        c.zhi = 0;
    }
    return c;
}

// d = half2 of exact integers (code - zero)  ->  two dequantised weights packed in ACT dtype.
template <int ACT, int SCL>
__device__ __forceinline__ uint32_t scale_pair(h2_t d, const ColConst& c) {
    if constexpr (SCL == kFP16) {
        const h2_t w = d * as_h2(c.s);  // single fp16 rounding == torch fp16 mul
        if constexpr (ACT == kFP16) {
            return as_u32(w);
        } else {
            return pack_bf16((float)w.x, (float)w.y);  // weights.to(bf16): second rounding
        }
    } else {
        const float s = __builtin_bit_cast(float, c.s);
        const uint32_t wb = pack_bf16((float)d.x * s, (float)d.y * s);  // exact product, one bf16 rounding
        if constexpr (ACT == kBF16) {
            return wb;
        } else {
            return pack_f16(bf16lo_to_f32(wb), bf16hi_to_f32(wb));  // weights.to(fp16)
        }
    }
}

// (w & mask) | magic must be ONE VALU op (v_and_or_b32).  With two literal constants hipcc has to split it into
// v_and + v_or (a VOP3 encoding reads at most one constant-bus value on gfx9): we hand it the masks in SGPRs and
// the magic in a VGPR, made opaque once per kernel so they stay in registers, and the pattern then selects itself.
struct DequantConsts {
    uint32_t magic;      // 0x64006400 in a VGPR
    uint32_t magic_bf;   // 0x43004300 in a VGPR: (nibble | magic_bf) = bf16 pair (128 + q) exactly
    uint32_t lo, hi;     // nibble (or byte) masks in SGPRs
    uint32_t sixteenth;  // half2(1/16, 1/16)
};
template <int BITS>
__device__ __forceinline__ DequantConsts make_dequant_consts() {
    DequantConsts k;
    k.magic = 0x64006400u;
    k.magic_bf = 0x43004300u;
    k.lo = BITS == 4 ? 0x000F000Fu : 0x00FF00FFu;
    k.hi = 0x00F000F0u;
    k.sixteenth = 0x2C002C00u;
    asm volatile("" : "+v"(k.magic));
    asm volatile("" : "+v"(k.magic_bf));
    asm volatile("" : "+s"(k.lo));
    asm volatile("" : "+s"(k.hi));
    return k;
}
__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask_sgpr, uint32_t magic_vgpr) {
    return (w & mask_sgpr) | magic_vgpr;
}

// One tiled int32 word of 4-bit codes -> MFMA B fragment (8 x 16-bit, natural k order).
template <int ACT, int SCL>
__device__ __forceinline__ u4_t dequant_word4(uint32_t w, const ColConst& c, const DequantConsts& k) {
    if constexpr (ACT == kBF16 && SCL == kBF16) {
        // bf16 scales and activations: W = bf16(s * (q - z)), ONE rounding of the exact product (torch bf16 mul).  gfx950 has no
        // packed bf16 VALU, but v_fma_mix_f32 reads an f16 HALF as an fma operand: OR the nibbles into the mantissa of 1024.0h
        // like the fp16 route (exact 1024 + q, or 1024 + 16 q for the nibbles sitting 4 bits up), then ONE mixed fma per
        // weight, (1024 + q) * s - (1024 + z) * s, lands on the exact product s * (q - z) in fp32 (19-bit intermediate, 13-bit
        // result), and v_cvt_pk_bf16_f32 rounds it once.  16 VALU per word (4 and_or + 8 fma_mix + 4 cvt_pk) instead of the 23
        // of the byte-convert route (2 and + shift, 8 cvt_f32_ubyte, 8 fma, 4 cvt_pk).
        const float s = __builtin_bit_cast(float, c.s);
        const uint32_t w8 = w >> 8;
        const uint32_t t0 = and_or(w, k.lo, k.magic), t1 = and_or(w, k.hi, k.magic);
        const uint32_t t2 = and_or(w8, k.lo, k.magic), t3 = and_or(w8, k.hi, k.magic);
        auto mix = [](uint32_t pair, float scale, float addend, float& lo, float& hi) {
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(pair), "v"(scale), "v"(addend));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(pair), "v"(scale), "v"(addend));
        };
        float f0, f1, f2, f3, f4, f5, f6, f7;
        mix(t0, s, c.clo, f0, f1);      // k0, k1
        mix(t1, c.s16, c.chi, f2, f3);  // k2, k3
        mix(t2, s, c.clo, f4, f5);      // k4, k5
        mix(t3, c.s16, c.chi, f6, f7);  // k6, k7
        u4_t r;
        r.x = pack_bf16(f0, f1);
        r.y = pack_bf16(f2, f3);
        r.z = pack_bf16(f4, f5);
        r.w = pack_bf16(f6, f7);
        return r;
    }
    const uint32_t LO = k.lo, HI = k.hi;
    const h2_t sixteenth = as_h2(k.sixteenth);
    const uint32_t w8 = w >> 8;
    u4_t r;
    r.x = scale_pair<ACT, SCL>(as_h2(and_or(w, LO, k.magic)) + as_h2(c.zlo), c);                                   // k0,k1
    r.y = scale_pair<ACT, SCL>(__builtin_elementwise_fma(as_h2(and_or(w, HI, k.magic)), sixteenth, as_h2(c.zhi)), c);   // k2,k3
    r.z = scale_pair<ACT, SCL>(as_h2(and_or(w8, LO, k.magic)) + as_h2(c.zlo), c);                                  // k4,k5
    r.w = scale_pair<ACT, SCL>(__builtin_elementwise_fma(as_h2(and_or(w8, HI, k.magic)), sixteenth, as_h2(c.zhi)), c);  // k6,k7
    return r;
}

// Two tiled int32 words of 8-bit codes (rows e=0..3 and 4..7) -> B fragment, natural k order.
template <int ACT, int SCL>
__device__ __forceinline__ u4_t dequant_word8(uint32_t w0, uint32_t w1, const ColConst& c, const DequantConsts& k) {
    const uint32_t MASK = k.lo;
    u4_t r;
    r.x = scale_pair<ACT, SCL>(as_h2(and_or(w0, MASK, k.magic)) + as_h2(c.zlo), c);       // k0,k1
    r.y = scale_pair<ACT, SCL>(as_h2(and_or(w0 >> 8, MASK, k.magic)) + as_h2(c.zlo), c);  // k2,k3
    r.z = scale_pair<ACT, SCL>(as_h2(and_or(w1, MASK, k.magic)) + as_h2(c.zlo), c);       // k4,k5
    r.w = scale_pair<ACT, SCL>(as_h2(and_or(w1 >> 8, MASK, k.magic)) + as_h2(c.zlo), c);  // k6,k7
    return r;
}

// D(16x16) += A(16x32) * B(32x16); lane l: A[m=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][n=l&15],
// D[m=4*(l>>4)+i][n=l&15]  (cdna_hip_programming.md §3).
template <int ACT>
__device__ __forceinline__ f4_t mfma16(u4_t a, u4_t b, f4_t c) {
    if constexpr (ACT == kFP16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
    }
}

}  // namespace gptqhip
