// Dev probe (not product): how fast can a launch of the skinny kernel's geometry merely READ its weights?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
template <int CPW, bool NT>
__global__ __launch_bounds__(1024) void probe(const u4_t* __restrict__ src, uint32_t* __restrict__ out, int chunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const u4_t* base = src + (size_t)blockIdx.x * chunks * 64 + lane;
    u4_t v[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = wave + i * W;
        if (c < chunks) v[i] = NT ? __builtin_nontemporal_load(base + (size_t)c * 64) : base[(size_t)c * 64];
        else v[i] = u4_t{0, 0, 0, 0};
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < CPW; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
extern "C" int probe_launch(const void* src, void* out, int tiles, int chunks, int waves, int nt, void* stream) {
    const int cpw = (chunks + waves - 1) / waves;
    dim3 g(tiles), b(64 * waves);
    hipStream_t s = (hipStream_t)stream;
#define L(C) if (nt) hipLaunchKernelGGL((probe<C, true>), g, b, 0, s, (const u4_t*)src, (uint32_t*)out, chunks); \
             else hipLaunchKernelGGL((probe<C, false>), g, b, 0, s, (const u4_t*)src, (uint32_t*)out, chunks);
    if (cpw <= 1) { L(1) } else if (cpw <= 2) { L(2) } else if (cpw <= 4) { L(4) } else if (cpw <= 8) { L(8) } else if (cpw <= 16) { L(16) } else return -1;
    return (int)hipGetLastError();
}
