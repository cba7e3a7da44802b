// Skinny fused dequant-GEMM for decode / small batch (M <= 64): HBM-bound streaming of the packed
// int32 qweight with the dequantisation fused into the MFMA contraction.
//
// Replaces the reference hot loop TorchLinear._forward_eager (gptqmodel/nn_modules/qlinear/torch.py:326-347)
// which materialises the whole fp16 [K,N] weight (torch.py:700-717) and then calls aten matmul.
//
// Mapping (wave64, mfma_f32_16x16x32):
//   * One int32 word of the GPTQ layout holds 8 consecutive k of ONE column -- exactly one lane's B
//     fragment of a 16x16x32 MFMA (lane l: B[k = 8*(l>>4)+j][n = l&15]).  No LDS, no shuffles: the
//     packed word goes HBM -> VGPR -> (mask|magic, add, mul) -> MFMA operand.
//   * A wave owns a strip of 64 columns.  Per K-step (32 k = 4 packed rows) lane l loads ONE dwordx4:
//     packed row 4*step + (l>>4), columns n0 + 4*(l&15) .. +3  -> 4 x 256 B contiguous segments per
//     wave-instruction (1 KiB), every fetched byte used once.  Word t of the dwordx4 feeds MFMA t, whose
//     16 output columns are {n0 + 4c + t}; a lane therefore ends up with 4 CONSECUTIVE output columns.
//   * Activations: lane l loads 16 B  x[m = l&15][k0 + 8*(l>>4) .. +7] straight from L2 (M*K*2 bytes,
//     cache resident) and permutes the 8 halves to the k-order the nibble masks produce.
//   * The 4 waves of a block split the block's K range (in-block split-K) and reduce through LDS; blocks
//     split K further (grid.y) when N/64 strips cannot fill 256 CUs.  Cross-block partials are fp32 slabs
//     published with write-through (sc1) stores + one relaxed agent-scope ticket; the last arriver reduces
//     them in a fixed order (deterministic, no float atomics) -- cdna_hip_programming.md §5 item 2.
#include "gptqhip_device.h"
#include "gptqhip_host.h"

namespace gptqhip {

struct SkinnyParams {
    const void* x;
    const int32_t* qw;
    const int32_t* qz;
    const void* scales;
    const void* bias;
    void* out;
    float* slabs;
    int* counters;
    int M, K, N, group_size;
    int chunks_total;      // K / (32*SPG)
    int chunks_per_split;  // ceil(chunks_total / splits)
    int splits;
};

template <int BITS, int SPG>
struct Stage {
    u4_t w[SPG][BITS == 4 ? 1 : 2];
    u2_t sc;      // 4 raw 16-bit scales of the lane's 4 columns
    uint32_t zw;  // packed zero word covering the lane's columns
};

template <int BITS, int SPG>
__device__ __forceinline__ void load_stage(Stage<BITS, SPG>& st, const SkinnyParams& p, int chunk, int col, int rq,
                                           bool ok) {
    if (!ok) {
#pragma unroll
        for (int s = 0; s < SPG; ++s) {
            st.w[s][0] = u4_t{0, 0, 0, 0};
            if constexpr (BITS == 8) st.w[s][1] = u4_t{0, 0, 0, 0};
        }
        st.sc = u2_t{0, 0};
        st.zw = 0;
        return;
    }
    const int step0 = chunk * SPG;
    const int g = (step0 * 32) / p.group_size;
#pragma unroll
    for (int s = 0; s < SPG; ++s) {
        if constexpr (BITS == 4) {
            const size_t prow = (size_t)(step0 + s) * 4 + rq;
            st.w[s][0] = *reinterpret_cast<const u4_t*>(p.qw + prow * p.N + col);
        } else {
            const size_t prow = (size_t)(step0 + s) * 8 + 2 * rq;
            st.w[s][0] = *reinterpret_cast<const u4_t*>(p.qw + prow * p.N + col);
            st.w[s][1] = *reinterpret_cast<const u4_t*>(p.qw + (prow + 1) * p.N + col);
        }
    }
    st.sc = *reinterpret_cast<const u2_t*>(reinterpret_cast<const uint16_t*>(p.scales) + (size_t)g * p.N + col);
    if constexpr (BITS == 4) {
        st.zw = (uint32_t)p.qz[(size_t)g * (p.N >> 3) + (col >> 3)] >> (4 * (col & 7));
    } else {
        st.zw = (uint32_t)p.qz[(size_t)g * (p.N >> 2) + (col >> 2)];
    }
}

template <int BITS, int ACT, int SCL, int MT, int SPG>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyParams p) {
    __shared__ float red[4][MT * 16][64];
    __shared__ int s_last;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int cq = lane & 15;
    const int rq = lane >> 4;
    const int strip = blockIdx.x;
    const int split = blockIdx.y;
    const int col = strip * 64 + cq * 4;
    const bool col_ok = col < p.N;

    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.chunks_total, c_begin + p.chunks_per_split);
    const int per_wave = (max(c_end - c_begin, 0) + 3) >> 2;
    const int wc0 = c_begin + wave * per_wave;
    const int wc1 = min(c_end, wc0 + per_wave);

    f4_t acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = f4_t{0.f, 0.f, 0.f, 0.f};

    const uint16_t* xs = reinterpret_cast<const uint16_t*>(p.x);

    Stage<BITS, SPG> cur, nxt;
    if (wc0 < wc1) load_stage<BITS, SPG>(cur, p, wc0, col, rq, col_ok);

    for (int c = wc0; c < wc1; ++c) {
        if (c + 1 < wc1) load_stage<BITS, SPG>(nxt, p, c + 1, col, rq, col_ok);

        // activations of this chunk (L2 resident)
        u4_t a[SPG][MT];
#pragma unroll
        for (int s = 0; s < SPG; ++s) {
            const int k0 = (c * SPG + s) * 32 + 8 * rq;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + cq;
                u4_t v = u4_t{0, 0, 0, 0};
                if (m < p.M) v = *reinterpret_cast<const u4_t*>(xs + (size_t)m * p.K + k0);
                a[s][mt] = v;
            }
        }

        // per-(group, column) constants
        ColConst cc[4];
        {
            const uint16_t s0 = (uint16_t)(cur.sc.x & 0xffffu), s1 = (uint16_t)(cur.sc.x >> 16);
            const uint16_t s2 = (uint16_t)(cur.sc.y & 0xffffu), s3 = (uint16_t)(cur.sc.y >> 16);
            constexpr uint32_t ZM = (1u << BITS) - 1u;
            cc[0] = make_col_const<SCL>(s0, cur.zw & ZM);
            cc[1] = make_col_const<SCL>(s1, (cur.zw >> BITS) & ZM);
            cc[2] = make_col_const<SCL>(s2, (cur.zw >> (2 * BITS)) & ZM);
            cc[3] = make_col_const<SCL>(s3, (cur.zw >> (3 * BITS)) & ZM);
        }

#pragma unroll
        for (int s = 0; s < SPG; ++s) {
            u4_t ap[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ap[mt] = permute_a<BITS>(a[s][mt]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                u4_t b;
                if constexpr (BITS == 4) {
                    b = dequant_word4<ACT, SCL>(cur.w[s][0][t], cc[t]);
                } else {
                    b = dequant_word8<ACT, SCL>(cur.w[s][0][t], cur.w[s][1][t], cc[t]);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][t] = mfma16<ACT>(ap[mt], b, acc[mt][t]);
            }
        }
        cur = nxt;
    }

    // ---- in-block split-K reduction through LDS -------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][(mt * 4 + t) * 4 + i][lane] = acc[mt][t][i];
    __syncthreads();

    // thread (wave w, lane) now owns row m = mt*16 + 4*rq + w, columns col .. col+3
    float v[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = (mt * 4 + t) * 4 + wave;
            v[mt][t] = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
        }

    if (p.splits > 1) {
        // publish fp32 partials write-through (sc1) -- no release fence needed
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 16 + 4 * rq + wave;
            if (m < p.M && col_ok) {
                unsigned long long* dst =
                    reinterpret_cast<unsigned long long*>(p.slabs + ((size_t)split * p.M + m) * p.N + col);
                const unsigned long long lo = (unsigned long long)__builtin_bit_cast(uint32_t, v[mt][0]) |
                                              ((unsigned long long)__builtin_bit_cast(uint32_t, v[mt][1]) << 32);
                const unsigned long long hi = (unsigned long long)__builtin_bit_cast(uint32_t, v[mt][2]) |
                                              ((unsigned long long)__builtin_bit_cast(uint32_t, v[mt][3]) << 32);
                __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(p.counters + strip, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == p.splits - 1);
        }
        __syncthreads();
        if (!s_last) return;
        // last arriver: deterministic reduction over the splits, slabs read with sc1 (L1-bypassing) loads
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 16 + 4 * rq + wave;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if (m < p.M && col_ok) {
                for (int sp = 0; sp < p.splits; ++sp) {
                    unsigned long long* src =
                        reinterpret_cast<unsigned long long*>(p.slabs + ((size_t)sp * p.M + m) * p.N + col);
                    const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s0 += __builtin_bit_cast(float, (uint32_t)lo);
                    s1 += __builtin_bit_cast(float, (uint32_t)(lo >> 32));
                    s2 += __builtin_bit_cast(float, (uint32_t)hi);
                    s3 += __builtin_bit_cast(float, (uint32_t)(hi >> 32));
                }
            }
            v[mt][0] = s0;
            v[mt][1] = s1;
            v[mt][2] = s2;
            v[mt][3] = s3;
        }
        if (threadIdx.x == 0) __hip_atomic_store(p.counters + strip, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- epilogue: round like the reference (matmul result, then += bias in the activation dtype) ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + 4 * rq + wave;
        if (m < p.M && col_ok) {
            float y[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                y[t] = round_through<ACT>(v[mt][t]);
                if (p.bias != nullptr) y[t] = y[t] + load16_as_f32<ACT>(p.bias, (size_t)col + t);
            }
            u2_t o;
            o.x = (uint32_t)f32_to_16<ACT>(y[0]) | ((uint32_t)f32_to_16<ACT>(y[1]) << 16);
            o.y = (uint32_t)f32_to_16<ACT>(y[2]) | ((uint32_t)f32_to_16<ACT>(y[3]) << 16);
            *reinterpret_cast<u2_t*>(reinterpret_cast<uint16_t*>(p.out) + (size_t)m * p.N + col) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int BITS, int ACT, int SCL, int MT>
static int launch_skinny_spg(const SkinnyParams& p0, int spg, hipStream_t stream) {
    SkinnyParams p = p0;
    const dim3 grid((p.N + 63) / 64, p.splits);
    if (spg == 4) {
        hipLaunchKernelGGL((skinny_kernel<BITS, ACT, SCL, MT, 4>), grid, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((skinny_kernel<BITS, ACT, SCL, MT, 1>), grid, dim3(256), 0, stream, p);
    }
    return check_hip(hipGetLastError(), "skinny_kernel launch");
}

template <int BITS, int ACT, int SCL>
static int launch_skinny_mt(const SkinnyParams& p, int spg, hipStream_t stream) {
    const int mt = (p.M + 15) / 16;
    if (mt <= 1) return launch_skinny_spg<BITS, ACT, SCL, 1>(p, spg, stream);
    if (mt <= 2) return launch_skinny_spg<BITS, ACT, SCL, 2>(p, spg, stream);
    return launch_skinny_spg<BITS, ACT, SCL, 4>(p, spg, stream);
}

SkinnyPlan plan_skinny(int M, int K, int N, int group_size, int force_split) {
    SkinnyPlan pl;
    pl.spg = (group_size % 128 == 0) ? 4 : 1;
    pl.chunks_total = K / (32 * pl.spg);
    const int strips = (N + 63) / 64;
    // fill ~2 blocks (8 waves) per CU; every wave of a split should own at least one chunk
    const int target_blocks = 2 * 256;
    int s = (target_blocks + strips / 2) / strips;
    const int max_s = pl.chunks_total / 4 > 0 ? pl.chunks_total / 4 : 1;
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (force_split > 0) s = force_split < pl.chunks_total ? force_split : pl.chunks_total;
    pl.chunks_per_split = (pl.chunks_total + s - 1) / s;
    pl.splits = (pl.chunks_total + pl.chunks_per_split - 1) / pl.chunks_per_split;
    pl.slab_floats = pl.splits > 1 ? (size_t)pl.splits * M * N : 0;
    pl.counters = strips;
    (void)M;
    return pl;
}

int launch_skinny(const GemmArgs& a, const SkinnyPlan& pl, float* slabs, int* counters, hipStream_t stream) {
    SkinnyParams p;
    p.x = a.x;
    p.qw = a.qweight;
    p.qz = a.qzeros;
    p.scales = a.scales;
    p.bias = a.bias;
    p.out = a.out;
    p.slabs = slabs;
    p.counters = counters;
    p.M = a.M;
    p.K = a.K;
    p.N = a.N;
    p.group_size = a.group_size;
    p.chunks_total = pl.chunks_total;
    p.chunks_per_split = pl.chunks_per_split;
    p.splits = pl.splits;
#define GPTQHIP_DISPATCH(B, A_, S_) return launch_skinny_mt<B, A_, S_>(p, pl.spg, stream)
    if (a.bits == 4) {
        if (a.act_dtype == kFP16 && a.scale_dtype == kFP16) GPTQHIP_DISPATCH(4, kFP16, kFP16);
        if (a.act_dtype == kBF16 && a.scale_dtype == kFP16) GPTQHIP_DISPATCH(4, kBF16, kFP16);
        if (a.act_dtype == kFP16 && a.scale_dtype == kBF16) GPTQHIP_DISPATCH(4, kFP16, kBF16);
        GPTQHIP_DISPATCH(4, kBF16, kBF16);
    } else {
        if (a.act_dtype == kFP16 && a.scale_dtype == kFP16) GPTQHIP_DISPATCH(8, kFP16, kFP16);
        if (a.act_dtype == kBF16 && a.scale_dtype == kFP16) GPTQHIP_DISPATCH(8, kBF16, kFP16);
        if (a.act_dtype == kFP16 && a.scale_dtype == kBF16) GPTQHIP_DISPATCH(8, kFP16, kBF16);
        GPTQHIP_DISPATCH(8, kBF16, kBF16);
    }
#undef GPTQHIP_DISPATCH
}

}  // namespace gptqhip
