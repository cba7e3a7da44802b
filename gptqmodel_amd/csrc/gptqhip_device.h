// Device-side building blocks shared by the gfx950 kernels: vector types, the int4/int8 -> fp16/bf16
// dequantisation that reproduces the reference's rounding bit-for-bit, and MFMA wrappers.
//
// Rounding contract being reproduced (reference gptqmodel/nn_modules/qlinear/torch.py:716-717,
// torch_awq.py:149-155, packing_utils.py:117-119):
//     W  = round_to(scales.dtype)( float(scale) * (code - zero) )         (one rounding)
//     W' = round_to(x.dtype)(W)                                           (only if dtypes differ)
// (code - zero) is a small exact integer.  We get it without any int->float convert: OR the nibble
// into the mantissa of 1024.0h (0x6400 | q  ==  1024 + q exactly) and add the precomputed constant
// -(1024 + zero); the fp16 multiply by the scale is then the single correctly-rounded product.
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

// 16-bit scalar load helpers ------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ float load16_as_f32(const void* p, size_t idx) {
    const uint16_t v = reinterpret_cast<const uint16_t*>(p)[idx];
    if constexpr (DT == kFP16) {
        return (float)__builtin_bit_cast(_Float16, v);
    } else {
        return __builtin_bit_cast(float, (uint32_t)v << 16);
    }
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

// Per-(group, column) dequant constants.
//   SCL == fp16: s = half2(scale, scale) bits
//   SCL == bf16: s = fp32 scale bits
//   zc  = half2(-(1024+zero), -(1024+zero)) bits   (bits 4) -- for 8-bit codes the same form works
//         because 1024 + q is exact in fp16 for q <= 1023.
struct ColConst {
    uint32_t s;
    uint32_t zc;
};

template <int SCL>
__device__ __forceinline__ ColConst make_col_const(uint16_t scale_bits, uint32_t zero) {
    ColConst c;
    if constexpr (SCL == kFP16) {
        c.s = (uint32_t)scale_bits | ((uint32_t)scale_bits << 16);
    } else {
        c.s = (uint32_t)scale_bits << 16;
    }
    const uint32_t z = 0xE400u | zero;  // -(1024 + zero) in fp16
    c.zc = z | (z << 16);
    return c;
}

// q2 = half2 bits (1024+qa, 1024+qb)  ->  two dequantised weights packed in ACT dtype.
template <int ACT, int SCL>
__device__ __forceinline__ uint32_t dequant_pair(uint32_t q2, const ColConst& c) {
    const h2_t d = as_h2(q2) + as_h2(c.zc);  // exact small integers (code - zero)
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

// One int32 word of 4-bit codes (k = 8r .. 8r+7 of one column) -> MFMA B fragment (8 x 16-bit) in the
// k-order [0,4,1,5,2,6,3,7] (the order the masks produce for free); the A fragment is permuted to match.
template <int ACT, int SCL>
__device__ __forceinline__ u4_t dequant_word4(uint32_t w, const ColConst& c) {
    constexpr uint32_t MAGIC = 0x64006400u, MASK = 0x000F000Fu;
    u4_t r;
    r.x = dequant_pair<ACT, SCL>((w & MASK) | MAGIC, c);          // k0, k4
    r.y = dequant_pair<ACT, SCL>(((w >> 4) & MASK) | MAGIC, c);   // k1, k5
    r.z = dequant_pair<ACT, SCL>(((w >> 8) & MASK) | MAGIC, c);   // k2, k6
    r.w = dequant_pair<ACT, SCL>(((w >> 12) & MASK) | MAGIC, c);  // k3, k7
    return r;
}

// Two int32 words of 8-bit codes (k = 8R..8R+3 and 8R+4..8R+7) -> B fragment in k-order [0,2,1,3,4,6,5,7].
template <int ACT, int SCL>
__device__ __forceinline__ u4_t dequant_word8(uint32_t w0, uint32_t w1, const ColConst& c) {
    constexpr uint32_t MAGIC = 0x64006400u, MASK = 0x00FF00FFu;
    u4_t r;
    r.x = dequant_pair<ACT, SCL>((w0 & MASK) | MAGIC, c);         // k0, k2
    r.y = dequant_pair<ACT, SCL>(((w0 >> 8) & MASK) | MAGIC, c);  // k1, k3
    r.z = dequant_pair<ACT, SCL>((w1 & MASK) | MAGIC, c);         // k4, k6
    r.w = dequant_pair<ACT, SCL>(((w1 >> 8) & MASK) | MAGIC, c);  // k5, k7
    return r;
}

// Permute 8 consecutive activations (x0..x7 as 4 dwords) into the B-fragment k-order.
template <int BITS>
__device__ __forceinline__ u4_t permute_a(u4_t a) {
    u4_t r;
    if constexpr (BITS == 4) {
        // [x0,x4,x1,x5,x2,x6,x3,x7]
        r.x = __builtin_amdgcn_perm(a.z, a.x, 0x05040100u);
        r.y = __builtin_amdgcn_perm(a.z, a.x, 0x07060302u);
        r.z = __builtin_amdgcn_perm(a.w, a.y, 0x05040100u);
        r.w = __builtin_amdgcn_perm(a.w, a.y, 0x07060302u);
    } else {
        // [x0,x2,x1,x3,x4,x6,x5,x7]
        r.x = __builtin_amdgcn_perm(a.y, a.x, 0x05040100u);
        r.y = __builtin_amdgcn_perm(a.y, a.x, 0x07060302u);
        r.z = __builtin_amdgcn_perm(a.w, a.z, 0x05040100u);
        r.w = __builtin_amdgcn_perm(a.w, a.z, 0x07060302u);
    }
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

// D(32x32) += A(32x16) * B(16x32); lane l: A[m=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][n=l&31],
// D[m=(i&3)+8*(i>>2)+4*(l>>5)][n=l&31].
template <int ACT>
__device__ __forceinline__ f16_t mfma32(u4_t a, u4_t b, f16_t c) {
    if constexpr (ACT == kFP16) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
    }
}

}  // namespace gptqhip
