// Auxiliary kernels around the hot path: standalone dequantisation (parity/debug + dequantize_weight()),
// the one-time post_init relayouts (AWQ -> canonical, act-order row sort) and the activation gather.
// All are pure HBM-bound integer/byte kernels: coalesced along N, one packed word per thread.
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
// act-order row sort: out row k' = in row perm[k'] at code granularity.
// ---------------------------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(256) void repack_rows_kernel(const int32_t* __restrict__ src,
                                                          const int32_t* __restrict__ perm,
                                                          int32_t* __restrict__ dst, int K, int N) {
    constexpr int PF = 32 / BITS;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= N) return;
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int k = perm[r * PF + j];
        const uint32_t s = (uint32_t)src[(size_t)(k / PF) * N + n];
        w |= ((s >> (BITS * (k % PF))) & MASK) << (BITS * j);
    }
    dst[(size_t)r * N + n] = (int32_t)w;
}

int launch_repack_rows(const int32_t* qweight, const int32_t* perm, int32_t* out, int K, int N, int bits,
                       hipStream_t stream) {
    const int pf = 32 / bits;
    const dim3 grid((N + 255) / 256, K / pf);
    if (bits == 4) {
        hipLaunchKernelGGL(repack_rows_kernel<4>, grid, dim3(256), 0, stream, qweight, perm, out, K, N);
    } else {
        hipLaunchKernelGGL(repack_rows_kernel<8>, grid, dim3(256), 0, stream, qweight, perm, out, K, N);
    }
    return check_hip(hipGetLastError(), "repack_rows launch");
}

// ---------------------------------------------------------------------------------------------
// out[m, k'] = x[m, perm[k']]   (ExllamaV2 gathers A through q_perm the same way,
// gptqmodel_ext/exllamav2/cuda/q_gemm_kernel_gptq.cuh:79-90)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_cols_kernel(const uint16_t* __restrict__ x,
                                                          const int32_t* __restrict__ perm,
                                                          uint16_t* __restrict__ out, int M, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int src = perm[k];
    for (int m = blockIdx.y; m < M; m += gridDim.y) out[(size_t)m * K + k] = x[(size_t)m * K + src];
}

int launch_gather_cols(const void* x, const int32_t* perm, void* out, int M, int K, hipStream_t stream) {
    const int gy = M < 1024 ? M : 1024;
    const dim3 grid((K + 255) / 256, gy);
    hipLaunchKernelGGL(gather_cols_kernel, grid, dim3(256), 0, stream, reinterpret_cast<const uint16_t*>(x), perm,
                       reinterpret_cast<uint16_t*>(out), M, K);
    return check_hip(hipGetLastError(), "gather_cols launch");
}

}  // namespace gptqhip
