// Auxiliary kernels around the hot path: standalone dequantisation (parity/debug + dequantize_weight()),
// the one-time post_init relayouts (AWQ -> canonical, act-order row sort) and the activation gather.
// All are pure HBM-bound integer/byte kernels: coalesced along N, one packed word per thread.
#include "gptqhip_codes.h"
#include "gptqhip_device.h"
#include "gptqhip_host.h"

namespace gptqhip {

// ---------------------------------------------------------------------------------------------
// dequant: thread (r, n) unpacks one packed word and writes pf rows of column n (coalesced over n).
// Reference: torch.py:700-717 (_dequantize_weight_cached_248) == qlinear/__init__.py:1001-1003.
// ---------------------------------------------------------------------------------------------
template <int BITS, int SCL, int OUT>
__global__ __launch_bounds__(256) void dequant_kernel(const int32_t* __restrict__ qw, const int32_t* __restrict__ qz,
                                                      const uint16_t* __restrict__ scales,
                                                      const int32_t* __restrict__ g_idx, uint16_t* __restrict__ out,
                                                      int K, int N, int group_size, int G) {
    constexpr int PF = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= N) return;
    const uint32_t w = (uint32_t)qw[(size_t)r * N + n];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int k = r * PF + j;
        int g = g_idx ? g_idx[k] : k / group_size;
        if (g < 0) g += G;  // python-style negative index wrap (scales[g_idx], torch.py:717)
        const uint32_t zw = (uint32_t)qz[(size_t)g * (N / PF) + n / PF];
        const int zero = (int)((zw >> (BITS * (n % PF))) & MASK);
        const int code = (int)((w >> (BITS * j)) & MASK);
        const float s = load16_as_f32<SCL>(scales, (size_t)g * N + n);
        float v = round_through<SCL>(s * (float)(code - zero));  // exact product, one rounding
        out[(size_t)k * N + n] = f32_to_16<OUT>(v);
    }
}

int launch_dequant(const int32_t* qweight, const int32_t* qzeros, const void* scales, const int32_t* g_idx, void* out,
                   int K, int N, int group_size, int bits, int scale_dtype, int out_dtype, hipStream_t stream) {
    const int pf = 32 / bits;
    const int G = K / group_size;
    const dim3 grid((N + 255) / 256, K / pf);
    const uint16_t* sc = reinterpret_cast<const uint16_t*>(scales);
    uint16_t* o = reinterpret_cast<uint16_t*>(out);
#define GPTQHIP_DQ(B, S_, O_)                                                                                       \
    hipLaunchKernelGGL((dequant_kernel<B, S_, O_>), grid, dim3(256), 0, stream, qweight, qzeros, sc, g_idx, o, K, N, \
                       group_size, G)
    if (bits == 4) {
        if (scale_dtype == kFP16 && out_dtype == kFP16) GPTQHIP_DQ(4, kFP16, kFP16);
        else if (scale_dtype == kFP16) GPTQHIP_DQ(4, kFP16, kBF16);
        else if (out_dtype == kFP16) GPTQHIP_DQ(4, kBF16, kFP16);
        else GPTQHIP_DQ(4, kBF16, kBF16);
    } else {
        if (scale_dtype == kFP16 && out_dtype == kFP16) GPTQHIP_DQ(8, kFP16, kFP16);
        else if (scale_dtype == kFP16) GPTQHIP_DQ(8, kFP16, kBF16);
        else if (out_dtype == kFP16) GPTQHIP_DQ(8, kBF16, kFP16);
        else GPTQHIP_DQ(8, kBF16, kBF16);
    }
#undef GPTQHIP_DQ
    return check_hip(hipGetLastError(), "dequant_kernel launch");
}

// ---------------------------------------------------------------------------------------------
// AWQ -> canonical.  AWQ word (k, c) nibble i holds logical column 8c + ORDER[i], ORDER = [0,2,4,6,1,3,5,7]
// (torch_awq.py:134-139); logical column 8c + j therefore sits in nibble REV[j], REV = [0,4,1,5,2,6,3,7]
// (packing_utils.py:10).  Output word (r, n) collects rows 8r..8r+7 of logical column n.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int awq_rev(int j) { return ((j & 1) << 2) | (j >> 1); }

__global__ __launch_bounds__(256) void repack_awq_qweight_kernel(const int32_t* __restrict__ src,
                                                                 int32_t* __restrict__ dst, int K, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= N) return;
    const int shift = 4 * awq_rev(n & 7);
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t s = (uint32_t)src[(size_t)(8 * r + j) * (N >> 3) + (n >> 3)];
        w |= ((s >> shift) & 0xFu) << (4 * j);
    }
    dst[(size_t)r * N + n] = (int32_t)w;
}

__global__ __launch_bounds__(256) void repack_awq_qzeros_kernel(const int32_t* __restrict__ src,
                                                                int32_t* __restrict__ dst, size_t words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    const uint32_t s = (uint32_t)src[i];
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) w |= ((s >> (4 * awq_rev(j))) & 0xFu) << (4 * j);
    dst[i] = (int32_t)w;
}

int launch_repack_awq(const int32_t* qw_awq, const int32_t* qz_awq, int32_t* qw_out, int32_t* qz_out, int K, int N,
                      int G, hipStream_t stream) {
    const dim3 grid((N + 255) / 256, K / 8);
    hipLaunchKernelGGL(repack_awq_qweight_kernel, grid, dim3(256), 0, stream, qw_awq, qw_out, K, N);
    const size_t words = (size_t)G * (N / 8);
    hipLaunchKernelGGL(repack_awq_qzeros_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, qz_awq,
                       qz_out, words);
    return check_hip(hipGetLastError(), "repack_awq launch");
}

// ---------------------------------------------------------------------------------------------
// Other bit widths (the reference's generic dequantize_weight, gptqmodel/nn_modules/qlinear/__init__.py:947-999):
// a checkpoint with 2-, 3-, 5-, 6- or 7-bit codes is brought to the continuous 4-bit (bits <= 4) or 8-bit layout the kernels read.
// The codes and zero-points keep their VALUES, only the field width grows, so W = scale * (code - zero) -- hence every result -- is
// unchanged.  Source layouts: continuous fields of `bits` bits in the little-endian bit stream of each group of 32 codes (2 / 4 / 8
// bits never straddle a word; 3 bits do, at codes 10 and 21: the reference special-cases exactly those, :982-991), or PLANAR
// (utils/planar_packing.py:7-24; always for 5 / 6 / 7 bits): per 32 codes `bits` words, low plane first, a plane of width w holding
// codes [i*32/w, (i+1)*32/w) of the group in its word i at shifts w*j.  One thread per OUTPUT word; reads and writes are coalesced
// along N for qweight (N-major words) and along the packed columns for qzeros.
// ---------------------------------------------------------------------------------------------
struct WidenSrc {
    int bits, planar;
};
// code `i` (0..31) of a group whose `bits` words are word(0) .. word(bits-1)
template <class F>
__device__ __forceinline__ uint32_t widen_extract(const WidenSrc& s, int i, F&& word) {
    if (!s.planar) {
        const int pos = s.bits * i, w = pos >> 5, sh = pos & 31;
        uint32_t v = word(w) >> sh;
        if (sh + s.bits > 32) v |= word(w + 1) << (32 - sh);
        return v & ((1u << s.bits) - 1u);
    }
    // planes: (width, offset) low to high -- 2: (2,0); 3: (2,0)(1,2); 4: (4,0); 5: (4,0)(1,4); 6: (4,0)(2,4); 7: (4,0)(2,4)(1,6); 8: (8,0)
    uint32_t v = 0;
    int row = 0, off = 0, left = s.bits;
    while (left > 0) {
        const int width = left >= 8 ? 8 : (left >= 4 ? 4 : (left >= 2 ? 2 : 1));
        const int pf = 32 / width;
        v |= ((word(row + i / pf) >> (width * (i % pf))) & ((1u << width) - 1u)) << off;
        row += width;
        off += width;
        left -= width;
    }
    return v;
}

__global__ __launch_bounds__(256) void widen_qweight_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int K, int N,
                                                            WidenSrc ws, int wide) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;                 // output word row: rows pf_out * r .. + pf_out - 1
    if (n >= N) return;
    const int pf_out = 32 / wide;
    uint32_t w = 0;
    for (int j = 0; j < pf_out; ++j) {
        const int k = r * pf_out + j;
        const int grp = k >> 5, i = k & 31;
        const uint32_t code = widen_extract(ws, i, [&](int t) { return (uint32_t)src[(size_t)(grp * ws.bits + t) * N + n]; });
        w |= code << (wide * j);
    }
    dst[(size_t)r * N + n] = (int32_t)w;
}

__global__ __launch_bounds__(256) void widen_qzeros_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int G, int N,
                                                           WidenSrc ws, int wide) {
    const int pf_out = 32 / wide;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;   // output word column: columns pf_out * c ..
    const int g = blockIdx.y;
    if (c >= N / pf_out) return;
    const size_t src_cols = (size_t)N * ws.bits / 32;
    uint32_t w = 0;
    for (int j = 0; j < pf_out; ++j) {
        const int n = c * pf_out + j;
        const int grp = n >> 5, i = n & 31;
        const uint32_t z = widen_extract(ws, i, [&](int t) { return (uint32_t)src[(size_t)g * src_cols + (size_t)grp * ws.bits + t]; });
        w |= z << (wide * j);
    }
    dst[(size_t)g * (N / pf_out) + c] = (int32_t)w;
}

int launch_widen_codes(const int32_t* qweight, const int32_t* qzeros, int32_t* qweight_out, int32_t* qzeros_out, int K, int N, int G,
                       int bits, int planar, hipStream_t stream) {
    const int wide = bits <= 4 ? 4 : 8;
    const WidenSrc ws = {bits, planar};
    hipLaunchKernelGGL(widen_qweight_kernel, dim3((N + 255) / 256, K * wide / 32), dim3(256), 0, stream, qweight, qweight_out, K, N, ws, wide);
    const int zc = N * wide / 32;
    hipLaunchKernelGGL(widen_qzeros_kernel, dim3((zc + 255) / 256, G), dim3(256), 0, stream, qzeros, qzeros_out, G, N, ws, wide);
    return check_hip(hipGetLastError(), "widen_codes launch");
}

// ---------------------------------------------------------------------------------------------
// canonical GPTQ layout (+ optional act-order row permutation) -> tile-major layout + meta constants.
// One thread per output word.  With perm, sorted row k' is checkpoint row perm[k'] (stable argsort of
// g_idx), so rows of one group become contiguous -- same role as ExllamaV2 make_sequential
// (gptqmodel_ext/exllamav2/cuda/q_matrix.cu:502-604) and marlin_sort_g_idx (gptqmodel/utils/marlin.py:368-372).
// ---------------------------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(256) void repack_tiled_kernel(const int32_t* __restrict__ src,
                                                           const int32_t* __restrict__ perm,
                                                           const int32_t* __restrict__ qz,
                                                           uint32_t* __restrict__ dst, int K, int N, int G, int chunks,
                                                           size_t total_words) {
    constexpr int PF = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_words) return;
    const int jj = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    size_t blk;
    int kbase;
    if constexpr (BITS == 4) {
        blk = idx >> 8;
        const int chunk = (int)(blk % chunks);
        kbase = kChunkK * chunk + 32 * jj + 8 * (lane >> 4);
    } else {
        const int h = (int)((idx >> 8) & 1);
        blk = idx >> 9;
        const int chunk = (int)(blk % chunks);
        kbase = kChunkK * chunk + 32 * (2 * h + (jj >> 1)) + 8 * (lane >> 4) + 4 * (jj & 1);
    }
    const int tile = (int)(blk / chunks);
    const int n = tile * kTileN + (lane & 15);
    uint32_t w = 0;
    if (n < N && kbase >= K) {
        // zero-padded tail of a K that is not a multiple of 128: store code == zero-point of the last group, so
        // the padded rows dequantise to EXACTLY 0 and the kernel needs no masking of the activations
        const uint32_t z = ((uint32_t)qz[(size_t)(G - 1) * (N / PF) + n / PF] >> (BITS * (n % PF))) & MASK;
#pragma unroll
        for (int e = 0; e < PF; ++e) w |= z << (BITS == 4 ? tiled_shift4(e) : tiled_shift8(e));
    } else if (n < N) {
        if (perm == nullptr) {
            const uint32_t s = (uint32_t)src[(size_t)(kbase / PF) * N + n];
#pragma unroll
            for (int e = 0; e < PF; ++e) {
                const uint32_t code = (s >> (BITS * e)) & MASK;
                w |= code << (BITS == 4 ? tiled_shift4(e) : tiled_shift8(e));
            }
        } else {
#pragma unroll
            for (int e = 0; e < PF; ++e) {
                const int k = perm[kbase + e];
                const uint32_t s = (uint32_t)src[(size_t)(k / PF) * N + n];
                const uint32_t code = (s >> (BITS * (k % PF))) & MASK;
                w |= code << (BITS == 4 ? tiled_shift4(e) : tiled_shift8(e));
            }
        }
    }
    dst[idx] = w;
}

template <int BITS>
__global__ __launch_bounds__(256) void build_meta_kernel(const int32_t* __restrict__ qz,
                                                         const uint16_t* __restrict__ scales,
                                                         uint32_t* __restrict__ meta, int N, int G, size_t total) {
    constexpr int PF = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx & 15);
    const int g = (int)((idx >> 4) % G);
    const int tile = (int)((idx >> 4) / G);
    const int n = tile * kTileN + c;
    uint32_t sb = 0, z = 0;
    if (n < N) {
        sb = scales[(size_t)g * N + n];
        z = ((uint32_t)qz[(size_t)g * (N / PF) + n / PF] >> (BITS * (n % PF))) & MASK;
    }
    meta[idx] = sb | ((0xE400u | z) << 16);
}

int launch_repack_tiled(const int32_t* qweight, const int32_t* qzeros, const void* scales, const int32_t* perm,
                        uint32_t* qweight_t, uint32_t* meta, int K, int N, int group_size, int bits,
                        hipStream_t stream) {
    const int chunks = ceil_div(K, kChunkK);
    const int tiles = ceil_div(N, kTileN);
    const int G = K / group_size;
    const size_t words = (size_t)tiles * chunks * (bits == 4 ? 256 : 512);
    const unsigned gw = (unsigned)((words + 255) / 256);
    if (qweight != nullptr) {  // NULL qweight/qweight_t: rebuild the meta constants only (e.g. new scale dtype)
        if (bits == 4) {
            hipLaunchKernelGGL(repack_tiled_kernel<4>, dim3(gw), dim3(256), 0, stream, qweight, perm, qzeros,
                               qweight_t, K, N, G, chunks, words);
        } else {
            hipLaunchKernelGGL(repack_tiled_kernel<8>, dim3(gw), dim3(256), 0, stream, qweight, perm, qzeros,
                               qweight_t, K, N, G, chunks, words);
        }
    }
    const size_t metas = (size_t)tiles * G * 16;
    const unsigned gm = (unsigned)((metas + 255) / 256);
    const uint16_t* sc = reinterpret_cast<const uint16_t*>(scales);
    if (bits == 4) {
        hipLaunchKernelGGL(build_meta_kernel<4>, dim3(gm), dim3(256), 0, stream, qzeros, sc, meta, N, G, metas);
    } else {
        hipLaunchKernelGGL(build_meta_kernel<8>, dim3(gm), dim3(256), 0, stream, qzeros, sc, meta, N, G, metas);
    }
    return check_hip(hipGetLastError(), "repack_tiled launch");
}

// ---------------------------------------------------------------------------------------------
// dequantise FROM the tile-major layout (module.dequantize_weight() after post_init).  Row k' of the tiled
// matrix is written to output row perm[k'] (checkpoint order) when perm is given.
// ---------------------------------------------------------------------------------------------
template <int BITS, int SCL, int OUT>
__global__ __launch_bounds__(256) void dequant_tiled_kernel(const uint32_t* __restrict__ qw,
                                                            const uint32_t* __restrict__ meta,
                                                            const int32_t* __restrict__ perm,
                                                            uint16_t* __restrict__ out, int K, int N, int G,
                                                            int group_size, int chunks, size_t total_words) {
    constexpr int PF = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_words) return;
    const int jj = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    size_t blk;
    int kbase;
    if constexpr (BITS == 4) {
        blk = idx >> 8;
        kbase = kChunkK * (int)(blk % chunks) + 32 * jj + 8 * (lane >> 4);
    } else {
        const int h = (int)((idx >> 8) & 1);
        blk = idx >> 9;
        kbase = kChunkK * (int)(blk % chunks) + 32 * (2 * h + (jj >> 1)) + 8 * (lane >> 4) + 4 * (jj & 1);
    }
    const int tile = (int)(blk / chunks);
    const int n = tile * kTileN + (lane & 15);
    if (n >= N || kbase >= K) return;
    const uint32_t w = qw[idx];
#pragma unroll
    for (int e = 0; e < PF; ++e) {
        const int k = kbase + e;
        const int g = k / group_size;
        const uint32_t mw = meta[((size_t)tile * G + g) * 16 + (lane & 15)];
        const int zero = (int)((mw >> 16) & 0x3FFu);  // 0xE400|z -> low 10 bits hold z
        const int code = (int)((w >> (BITS == 4 ? tiled_shift4(e) : tiled_shift8(e))) & MASK);
        const float s = bits16_to_f32<SCL>((uint16_t)(mw & 0xffffu));
        const float v = round_through<SCL>(s * (float)(code - zero));
        const int row = perm ? perm[k] : k;
        out[(size_t)row * N + n] = f32_to_16<OUT>(v);
    }
}

int launch_dequant_tiled(const uint32_t* qweight_t, const uint32_t* meta, const int32_t* perm, void* out, int K, int N,
                         int group_size, int bits, int scale_dtype, int out_dtype, hipStream_t stream) {
    const int chunks = ceil_div(K, kChunkK);
    const int tiles = ceil_div(N, kTileN);
    const int G = K / group_size;
    const size_t words = (size_t)tiles * chunks * (bits == 4 ? 256 : 512);
    const dim3 grid((unsigned)((words + 255) / 256));
    uint16_t* o = reinterpret_cast<uint16_t*>(out);
#define GPTQHIP_DQT(B, S_, O_)                                                                                    \
    hipLaunchKernelGGL((dequant_tiled_kernel<B, S_, O_>), grid, dim3(256), 0, stream, qweight_t, meta, perm, o, K, N, \
                       G, group_size, chunks, words)
    if (bits == 4) {
        if (scale_dtype == kFP16 && out_dtype == kFP16) GPTQHIP_DQT(4, kFP16, kFP16);
        else if (scale_dtype == kFP16) GPTQHIP_DQT(4, kFP16, kBF16);
        else if (out_dtype == kFP16) GPTQHIP_DQT(4, kBF16, kFP16);
        else GPTQHIP_DQT(4, kBF16, kBF16);
    } else {
        if (scale_dtype == kFP16 && out_dtype == kFP16) GPTQHIP_DQT(8, kFP16, kFP16);
        else if (scale_dtype == kFP16) GPTQHIP_DQT(8, kFP16, kBF16);
        else if (out_dtype == kFP16) GPTQHIP_DQT(8, kBF16, kFP16);
        else GPTQHIP_DQT(8, kBF16, kBF16);
    }
#undef GPTQHIP_DQT
    return check_hip(hipGetLastError(), "dequant_tiled_kernel launch");
}

// ---------------------------------------------------------------------------------------------
// out[m, k'] = x[m, perm[k']]   (ExllamaV2 gathers A through q_perm the same way,
// gptqmodel_ext/exllamav2/cuda/q_gemm_kernel_gptq.cuh:79-90)
// ---------------------------------------------------------------------------------------------
// One block stages whole rows of x in LDS with coalesced 16-byte loads, then every thread assembles 8 consecutive
// output columns from LDS (2-byte reads at perm[k']) and writes them as one 16-byte store: both HBM streams are
// fully coalesced, the random access happens on-chip.  Rows longer than kGatherMaxK halves fall back to the simple
// kernel.
constexpr int kGatherMaxK = 16384;  // 32 KiB of LDS per staged row

// byte offsets (2 * column) of 8 consecutive permutation entries, packed two per register
__device__ __forceinline__ u4_t pack_byte_offsets(const int32_t* __restrict__ p) {
    const u4_t a = *reinterpret_cast<const u4_t*>(p), b = *reinterpret_cast<const u4_t*>(p + 4);
    u4_t o;
    o.x = (a.x << 1) | (a.y << 17);
    o.y = (a.z << 1) | (a.w << 17);
    o.z = (b.x << 1) | (b.y << 17);
    o.w = (b.z << 1) | (b.w << 17);
    return o;
}
__device__ __forceinline__ uint16_t lds_u16(const uint16_t* row, uint32_t byte_off) {
    return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(row) + byte_off);
}

// R rows per pass: all R*IT 16-byte loads of a pass are issued before the first LDS write (R * K * 2 bytes in flight per
// block instead of one row's latency per row).  A thread assembles the SAME output columns in every pass, so its slice of
// the permutation lives in registers (read once per block), and the next pass's rows are requested as soon as this pass's
// have been parked in LDS: they land under the gather phase instead of in front of it (round 3: a pass used to be
// load -> park -> permutation load -> gather -> store, each step waiting for the one before: 2 TB/s of the ~5 a copy reaches).
template <int R, int IT>
__global__ __launch_bounds__(256) void gather_cols_lds_kernel(const uint16_t* __restrict__ x,
                                                              const int32_t* __restrict__ perm,
                                                              uint16_t* __restrict__ out, int M, int K) {
    extern __shared__ __attribute__((aligned(16))) uint16_t rows[];  // [R][K]
    const int k8 = K / 8;
    int piece[IT];      // this thread's 16-byte output pieces (clamped: out-of-range pieces load piece 0 and store nothing)
    u4_t off[IT];       // byte offsets of the piece's 8 source columns inside a staged row, two per register (K <= 16384: 15 bits each)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x + it * 256;
        piece[it] = i < k8 ? i : 0;
        off[it] = pack_byte_offsets(perm + 8 * piece[it]);
    }
    u4_t stage[R][IT];
    auto request = [&](int m0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = m0 + r < M ? m0 + r : M - 1;
            const u4_t* src = reinterpret_cast<const u4_t*>(x + (size_t)m * K);
#pragma unroll
            for (int it = 0; it < IT; ++it) stage[r][it] = src[piece[it]];
        }
    };
    int m0 = blockIdx.x * R;
    if (m0 < M) request(m0);
    for (; m0 < M; m0 += gridDim.x * R) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int it = 0; it < IT; ++it)
                if (threadIdx.x + it * 256 < k8) reinterpret_cast<u4_t*>(rows + (size_t)r * K)[piece[it]] = stage[r][it];
        const int m_next = m0 + gridDim.x * R;
        if (m_next < M) request(m_next);   // in flight under the gather phase
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (threadIdx.x + it * 256 < k8) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (m0 + r < M) {
                        const uint16_t* row = rows + (size_t)r * K;
                        u4_t v;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            v[j] = (uint32_t)lds_u16(row, off[it][j] & 0xffffu) | ((uint32_t)lds_u16(row, off[it][j] >> 16) << 16);
                        reinterpret_cast<u4_t*>(out + (size_t)(m0 + r) * K)[piece[it]] = v;
                    }
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gather_cols_kernel(const uint16_t* __restrict__ x,
                                                          const int32_t* __restrict__ perm,
                                                          uint16_t* __restrict__ out, int M, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int src = perm[k];
    for (int m = blockIdx.y; m < M; m += gridDim.y) out[(size_t)m * K + k] = x[(size_t)m * K + src];
}

int launch_gather_cols(const void* x, const int32_t* perm, void* out, int M, int K, hipStream_t stream) {
    const uint16_t* xs = reinterpret_cast<const uint16_t*>(x);
    uint16_t* os = reinterpret_cast<uint16_t*>(out);
    if (K % 8 == 0 && K <= kGatherMaxK && M >= 8) {
        const int it = ceil_div(K / 8, 256);  // 16-byte pieces per thread and row
        const int r = K <= 8192 ? 4 : 2;      // rows per pass (<= 64 KiB of LDS)
        const int passes = ceil_div(M, r);
        const dim3 grid(passes < 2048 ? passes : 2048);
        const size_t lds = (size_t)r * K * 2;
#define GPTQHIP_GATHER(R_, IT_) hipLaunchKernelGGL((gather_cols_lds_kernel<R_, IT_>), grid, dim3(256), lds, stream, xs, perm, os, M, K)
        if (r == 4) {
            if (it <= 1) GPTQHIP_GATHER(4, 1);
            else if (it <= 2) GPTQHIP_GATHER(4, 2);
            else GPTQHIP_GATHER(4, 4);
        } else {
            if (it <= 6) GPTQHIP_GATHER(2, 6);
            else GPTQHIP_GATHER(2, 8);
        }
#undef GPTQHIP_GATHER
    } else {
        const int gy = M < 1024 ? M : 1024;
        const dim3 grid((K + 255) / 256, gy);
        hipLaunchKernelGGL(gather_cols_kernel, grid, dim3(256), 0, stream, xs, perm, os, M, K);
    }
    return check_hip(hipGetLastError(), "gather_cols launch");
}

// ---------------------------------------------------------------------------------------------
// RMSNorm (+ act-order gather) in ONE pass over the activations: out[m, k'] = w[p] * act(h32[m, p] * rsqrt(mean_k h32[m, k]^2 + eps)),
// p = perm[k'] (perm NULL: p = k').  HF LlamaRMSNorm semantics (fp32 statistics, the normalised value rounded to the activation
// dtype, then the product with the weight rounded once) -- the caller of the quantised q|k|v and gate|up projections.  An
// act-order checkpoint's prefill otherwise pays a separate gather pass over x per linear (one extra read + write of [M, K]:
// -10 % at 4096^2, M = 65536) on top of eager HF's six small RMSNorm kernels with fp32 temporaries; here the permuted,
// normalised x leaves the kernel that had to read h anyway.  Same structure as gather_cols_lds_kernel: R rows per pass staged
// in LDS, every HBM access a coalesced 16-byte piece, the random access happens on-chip.  HBM-bound: M * K * 4 bytes.
// ---------------------------------------------------------------------------------------------
template <int ACT, int R, int IT>
__global__ __launch_bounds__(256) void rmsnorm_gather_kernel(const uint16_t* __restrict__ h, const uint16_t* __restrict__ weight,
                                                             const int32_t* __restrict__ perm, uint16_t* __restrict__ out, int M, int K,
                                                             float eps) {
    extern __shared__ __attribute__((aligned(16))) uint16_t rows[];  // [R][K] normalised rows
    __shared__ float red[4][R];
    const int k8 = K / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a thread writes the same output columns in every pass: their source columns (as byte offsets into a staged row, two per
    // register) and norm weights (raw 16-bit pairs) stay in registers
    int piece[IT];
    u4_t off[IT], wv[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x + it * 256;
        piece[it] = i < k8 ? i : 0;
        if (perm != nullptr) {
            const int32_t* pp = perm + 8 * piece[it];
            off[it] = pack_byte_offsets(pp);
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[it][j] = (uint32_t)weight[pp[2 * j]] | ((uint32_t)weight[pp[2 * j + 1]] << 16);
        } else {
            wv[it] = reinterpret_cast<const u4_t*>(weight)[piece[it]];
#pragma unroll
            for (int j = 0; j < 4; ++j) off[it][j] = (uint32_t)(16 * piece[it] + 4 * j) | ((uint32_t)(16 * piece[it] + 4 * j + 2) << 16);
        }
    }
    u4_t stage[R][IT];
    auto request = [&](int m0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = m0 + r < M ? m0 + r : M - 1;
            const u4_t* s = reinterpret_cast<const u4_t*>(h + (size_t)m * K);
#pragma unroll
            for (int it = 0; it < IT; ++it) stage[r][it] = s[piece[it]];
        }
    };
    int m0 = blockIdx.x * R;
    if (m0 < M) request(m0);
    for (; m0 < M; m0 += gridDim.x * R) {
        // statistics: fp32 sum of squares per row, fixed order (thread pieces, shuffle tree, four waves)
        float ss[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float a = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                if (threadIdx.x + it * 256 < k8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = bits16_to_f32<ACT>((uint16_t)(stage[r][it][j] & 0xffffu)), hi = bits16_to_f32<ACT>((uint16_t)(stage[r][it][j] >> 16));
                        a = __builtin_fmaf(lo, lo, a);
                        a = __builtin_fmaf(hi, hi, a);
                    }
                }
            }
#pragma unroll
            for (int mk = 32; mk >= 1; mk >>= 1) a += __shfl_xor(a, mk, 64);
            ss[r] = a;
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) red[wave][r] = ss[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float inv = rsqrtf((red[0][r] + red[1][r] + red[2][r] + red[3][r]) / (float)K + eps);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                if (threadIdx.x + it * 256 < k8) {
                    u4_t v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = bits16_to_f32<ACT>((uint16_t)(stage[r][it][j] & 0xffffu)), hi = bits16_to_f32<ACT>((uint16_t)(stage[r][it][j] >> 16));
                        v[j] = (uint32_t)f32_to_16<ACT>(lo * inv) | ((uint32_t)f32_to_16<ACT>(hi * inv) << 16);
                    }
                    reinterpret_cast<u4_t*>(rows + (size_t)r * K)[piece[it]] = v;
                }
            }
        }
        const int m_next = m0 + gridDim.x * R;
        if (m_next < M) request(m_next);   // the next pass's rows land under the gather phase
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (threadIdx.x + it * 256 < k8) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (m0 + r < M) {
                        const uint16_t* row = rows + (size_t)r * K;
                        u4_t v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = bits16_to_f32<ACT>((uint16_t)(wv[it][j] & 0xffffu)) * bits16_to_f32<ACT>(lds_u16(row, off[it][j] & 0xffffu));
                            const float b = bits16_to_f32<ACT>((uint16_t)(wv[it][j] >> 16)) * bits16_to_f32<ACT>(lds_u16(row, off[it][j] >> 16));
                            v[j] = (uint32_t)f32_to_16<ACT>(a) | ((uint32_t)f32_to_16<ACT>(b) << 16);
                        }
                        reinterpret_cast<u4_t*>(out + (size_t)(m0 + r) * K)[piece[it]] = v;
                    }
                }
            }
        }
        __syncthreads();   // (also orders this pass's reads of red[] / rows[] before the next pass's writes)
    }
}

int launch_rmsnorm_gather(const void* h, const void* weight, const int32_t* perm, void* out, int M, int K, float eps, int act_dtype,
                          hipStream_t stream) {
    const uint16_t* hs = reinterpret_cast<const uint16_t*>(h);
    const uint16_t* ws = reinterpret_cast<const uint16_t*>(weight);
    uint16_t* os = reinterpret_cast<uint16_t*>(out);
    const int it = ceil_div(K / 8, 256);  // 16-byte pieces per thread and row
    const int r = K <= 8192 ? 4 : 2;      // rows per pass
    const int passes = ceil_div(M, r);
    const dim3 grid(passes < 2048 ? passes : 2048);
    const size_t lds = (size_t)r * K * 2;
#define GPTQHIP_RMSG(A_, R_, IT_)                                                                                       \
    do {                                                                                                                  \
        auto kern = rmsnorm_gather_kernel<A_, R_, IT_>;                                                                   \
        if (lds > 64 * 1024) {                                                                                            \
            int rc_ = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), \
                                "rmsnorm_gather: LDS size");                                                              \
            if (rc_) return rc_;                                                                                          \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, hs, ws, perm, os, M, K, eps);                              \
    } while (0)
#define GPTQHIP_RMSG_A(R_, IT_)                                      \
    do {                                                             \
        if (act_dtype == kFP16) GPTQHIP_RMSG(kFP16, R_, IT_);        \
        else GPTQHIP_RMSG(kBF16, R_, IT_);                           \
    } while (0)
    if (r == 4) {
        if (it <= 1) GPTQHIP_RMSG_A(4, 1);
        else if (it <= 2) GPTQHIP_RMSG_A(4, 2);
        else GPTQHIP_RMSG_A(4, 4);
    } else {
        if (it <= 6) GPTQHIP_RMSG_A(2, 6);
        else GPTQHIP_RMSG_A(2, 8);
    }
#undef GPTQHIP_RMSG_A
#undef GPTQHIP_RMSG
    return check_hip(hipGetLastError(), "rmsnorm_gather launch");
}

// ---------------------------------------------------------------------------------------------
// embedding gather-dequant from the tiled layout: one thread per (token, column).  Reads one packed word per output
// element (8x read amplification on a tokens x dim problem that is tiny next to the table itself) instead of
// materialising the whole [vocab, dim] table like the reference does.
// ---------------------------------------------------------------------------------------------
template <int BITS, int SCL>
__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ ids, const uint32_t* __restrict__ qw,
                                                        const uint32_t* __restrict__ meta,
                                                        const int32_t* __restrict__ inv_perm, uint16_t* __restrict__ out,
                                                        int32_t* __restrict__ status, int T, int K, int N, int G,
                                                        int group_size, int chunks) {
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (n >= N) return;
    const int64_t id = ids[t];
    if (id < 0 || id >= K) {
        if (n == 0) atomicExch(status, 1);
        out[(size_t)t * N + n] = 0;
        return;
    }
    const int k = inv_perm ? inv_perm[id] : (int)id;  // row in the (group-sorted) tiled matrix
    const int tile = n >> 4, c = n & 15, chunk = k >> 7, r = k & 127;
    const int j = r >> 5, rq = (r >> 3) & 3;
    uint32_t code;
    if constexpr (BITS == 4) {
        const uint32_t w = qw[(((size_t)tile * chunks + chunk) * 64 + (rq << 4 | c)) * 4 + j];
        code = (w >> tiled_shift4(r & 7)) & MASK;
    } else {
        const int e8 = r & 7, half = e8 >> 2, h = j >> 1, jj = (j & 1) * 2 + half;
        const uint32_t w = qw[((((size_t)tile * chunks + chunk) * 2 + h) * 64 + (rq << 4 | c)) * 4 + jj];
        code = (w >> tiled_shift8(e8 & 3)) & MASK;
    }
    const uint32_t mw = meta[((size_t)tile * G + k / group_size) * 16 + c];
    const int zero = (int)((mw >> 16) & 0x3FFu);
    const float s = bits16_to_f32<SCL>((uint16_t)(mw & 0xffffu));
    out[(size_t)t * N + n] = f32_to_16<SCL>(s * (float)((int)code - zero));  // exact product, one rounding
}

int launch_embedding(const int64_t* ids, const uint32_t* qw, const uint32_t* meta, const int32_t* inv_perm, void* out,
                     int32_t* status, int T, int K, int N, int group_size, int bits, int scale_dtype,
                     hipStream_t stream) {
    const int chunks = ceil_div(K, kChunkK);
    const int G = K / group_size;
    const dim3 grid((N + 255) / 256, T);
    uint16_t* o = reinterpret_cast<uint16_t*>(out);
#define GPTQHIP_EMB(B, S_) \
    hipLaunchKernelGGL((embedding_kernel<B, S_>), grid, dim3(256), 0, stream, ids, qw, meta, inv_perm, o, status, T, K, N, G, group_size, chunks)
    if (bits == 4) {
        if (scale_dtype == kFP16) GPTQHIP_EMB(4, kFP16); else GPTQHIP_EMB(4, kBF16);
    } else {
        if (scale_dtype == kFP16) GPTQHIP_EMB(8, kFP16); else GPTQHIP_EMB(8, kBF16);
    }
#undef GPTQHIP_EMB
    return check_hip(hipGetLastError(), "embedding_kernel launch");
}

// ---------------------------------------------------------------------------------------------
// quantise-and-pack (reference: gptqmodel_ext/pack_block_cpu.cpp:105-190).  Thread (r, n) packs the pf codes of
// packed row r, column n.  IEEE fp32 semantics of the CPU packer: correctly rounded divide, rint, and NO contraction of the multiply
// and the add into an fma (HIP's __fmul_rn / __fadd_rn are plain operators, which hipcc fuses by default).
// ---------------------------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(256) void pack_qweight_kernel(const float* __restrict__ weight,
                                                           const float* __restrict__ scales,
                                                           const int32_t* __restrict__ zeros,
                                                           const int32_t* __restrict__ g_idx,
                                                           int32_t* __restrict__ qweight, int K, int N, int G) {
#pragma clang fp contract(off)      // the reference multiplies, THEN adds (two roundings): an fma here changes codes at rounding ties
    constexpr int PF = 32 / BITS;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= N) return;
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int k = r * PF + j;
        int g = g_idx[k];
        if (g < 0) g += G;
        g = g < 0 ? 0 : (g >= G ? G - 1 : g);  // range is validated on the host; never index out of bounds
        float scale = scales[(size_t)g * N + n];
        const float offset = (float)zeros[(size_t)g * N + n] * scale;      // plain operators: the pragma above governs THEM (the
        if (scale == 0.0f) scale = 1e-6f;                                   // __fmul_rn / __fadd_rn wrappers carry the header's flags)
        const float sum = weight[(size_t)n * K + k] + offset;
        float q = rintf(sum / scale);
        q = fmaxf(0.0f, fminf(q, (float)((1 << BITS) - 1)));
        w |= ((uint32_t)(int)q) << (BITS * j);
    }
    qweight[(size_t)r * N + n] = (int32_t)w;
}

template <int BITS>
__global__ __launch_bounds__(256) void pack_qzeros_kernel(const int32_t* __restrict__ zeros,
                                                          int32_t* __restrict__ qzeros, int N, size_t words) {
    constexpr int PF = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    const size_t g = i / (N / PF), c = i % (N / PF);
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < PF; ++j) w |= ((uint32_t)zeros[g * N + c * PF + j] & MASK) << (BITS * j);
    qzeros[i] = (int32_t)w;
}

// the other bit widths / the planar layouts (gptqhip_codes.h): thread (group of 32 rows, n) quantises its 32 codes and writes the
// group's `bits` words; thread (g, group of 32 columns) does the same for the zero-points.  Same fp32 arithmetic as above.
__global__ __launch_bounds__(256) void pack_qweight_any_kernel(const float* __restrict__ weight, const float* __restrict__ scales,
                                                               const int32_t* __restrict__ zeros, const int32_t* __restrict__ g_idx,
                                                               int32_t* __restrict__ qweight, int K, int N, int G, int bits, int planar) {
#pragma clang fp contract(off)      // see pack_qweight_kernel
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = blockIdx.y;
    if (n >= N) return;
    const float maxq = (float)((1 << bits) - 1);
    uint32_t c[32], out[8];
    for (int i = 0; i < 32; ++i) {
        const int k = grp * 32 + i;
        int g = g_idx[k];
        if (g < 0) g += G;
        g = g < 0 ? 0 : (g >= G ? G - 1 : g);
        float scale = scales[(size_t)g * N + n];
        const float offset = (float)zeros[(size_t)g * N + n] * scale;      // plain operators: the pragma above governs THEM (the
        if (scale == 0.0f) scale = 1e-6f;                                   // __fmul_rn / __fadd_rn wrappers carry the header's flags)
        const float sum = weight[(size_t)n * K + k] + offset;
        float q = rintf(sum / scale);
        c[i] = (uint32_t)(int)fmaxf(0.0f, fminf(q, maxq));
    }
    encode_group32(c, bits, planar, out);
    for (int t = 0; t < bits; ++t) qweight[((size_t)grp * bits + t) * N + n] = (int32_t)out[t];
}

__global__ __launch_bounds__(256) void pack_qzeros_any_kernel(const int32_t* __restrict__ zeros, int32_t* __restrict__ qzeros, int N, int G,
                                                              int bits, int planar) {
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;       // group of 32 columns
    const int g = blockIdx.y;
    if (cg >= N / 32) return;
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t c[32], out[8];
    for (int i = 0; i < 32; ++i) c[i] = (uint32_t)zeros[(size_t)g * N + cg * 32 + i] & mask;
    encode_group32(c, bits, planar, out);
    for (int t = 0; t < bits; ++t) qzeros[(size_t)g * ((size_t)N * bits / 32) + (size_t)cg * bits + t] = (int32_t)out[t];
}

int launch_pack_gptq(const float* weight, const float* scales, const int32_t* zeros, const int32_t* g_idx,
                     int32_t* qweight, int32_t* qzeros, int K, int N, int G, int bits, int planar, hipStream_t stream) {
    if (bits == 2 || bits == 4 || bits == 8) planar = 0;    // planar words of these widths ARE the continuous ones
    if (bits != 4 && bits != 8) {
        hipLaunchKernelGGL(pack_qweight_any_kernel, dim3((N + 255) / 256, K / 32), dim3(256), 0, stream, weight, scales, zeros, g_idx, qweight,
                           K, N, G, bits, planar);
        hipLaunchKernelGGL(pack_qzeros_any_kernel, dim3((N / 32 + 255) / 256, G), dim3(256), 0, stream, zeros, qzeros, N, G, bits, planar);
        return check_hip(hipGetLastError(), "pack_gptq launch");
    }
    const int pf = 32 / bits;
    const dim3 grid((N + 255) / 256, K / pf);
    const size_t words = (size_t)G * (N / pf);
    const dim3 gz((unsigned)((words + 255) / 256));
    if (bits == 4) {
        hipLaunchKernelGGL(pack_qweight_kernel<4>, grid, dim3(256), 0, stream, weight, scales, zeros, g_idx, qweight, K, N, G);
        hipLaunchKernelGGL(pack_qzeros_kernel<4>, gz, dim3(256), 0, stream, zeros, qzeros, N, words);
    } else {
        hipLaunchKernelGGL(pack_qweight_kernel<8>, grid, dim3(256), 0, stream, weight, scales, zeros, g_idx, qweight, K, N, G);
        hipLaunchKernelGGL(pack_qzeros_kernel<8>, gz, dim3(256), 0, stream, zeros, qzeros, N, words);
    }
    return check_hip(hipGetLastError(), "pack_gptq launch");
}

}  // namespace gptqhip
