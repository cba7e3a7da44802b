// Dev probe (not product): what would ONE persistent launch per decoder layer buy the batch-1 decode chain?  
// A layer = four dependent ops (o, gate_up, down, next qkv); an op = every block streams its share of the op's packed weights
// (pure reads, the decode kernel's geometry: 256 blocks x 16 waves, 1 KiB per wave load) and publishes a few bytes.  Three forms:
//   mode 0  four launches per layer (what the product does: dependent-kernel boundaries between the ops)
//   mode 1  one persistent launch, an XCD-hierarchical grid barrier between the ops
//   mode 2  mode 1 + before arriving at the barrier a wave requests the first PF KiB of its share of the NEXT op (weights do not
//           depend on activations) and consumes them after the barrier
// No dequant, no MFMA, no activations: an upper bound for what the barrier form can reach, measured before building the real thing.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));

struct Bar {
    uint32_t xcc[8][64];   // per-XCD arrival counters (monotonic), 256 B apart
    uint32_t top[64];      // XCD leaders
    uint32_t gen[8][64];   // per-XCD generation, written by the releasing leader
    uint32_t census[8][64];
    uint32_t status[64];
};

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }

// all blocks of the grid; `k` = the number of barriers passed so far + 1; blocks_x = blocks on this XCD (census), nx = XCDs in use
__device__ __forceinline__ void grid_barrier(Bar* b, int x, uint32_t k, uint32_t blocks_x, uint32_t nx) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t old = __hip_atomic_fetch_add(&b->xcc[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == k * blocks_x - 1u) {   // last block of this XCD
            const uint32_t t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == k * nx - 1u) {       // last XCD: release everybody
                for (int i = 0; i < 8; ++i) __hip_atomic_store(&b->gen[i][0], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned spins = 0;
        while (__hip_atomic_load(&b->gen[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {
                __hip_atomic_store(&b->status[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

struct Ops {
    const u4_t* w[4];   // per op: flat array of 1 KiB blocks (64 lanes x 16 B)
    int kib[4];         // KiB blocks per op
};

constexpr int PF = 4;   // KiB per wave requested ahead of the barrier (the decode kernel's ring depth)

template <int MODE>
__global__ __launch_bounds__(1024) void layer_probe(Ops ops, size_t layer_stride_u4, int layers, int op_first, int op_last, Bar* bar,
                                                    uint32_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int gw = blockIdx.x * W + wave, GW = gridDim.x * W;   // this wave among all waves of the grid
    int x = 0;
    uint32_t blocks_x = 0, nx = 0, k = 0;
    if (MODE >= 1) {
        // census (once per launch): how many blocks does each XCD hold?  One counter barrier on `top` would need the answer, so the
        // census itself is awaited with a plain global count
        x = xcc_id();
        __shared__ uint32_t sh[2];
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&bar->census[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar->census[0][32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&bar->census[0][32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
            uint32_t n = 0;
            for (int i = 0; i < 8; ++i) n += __hip_atomic_load(&bar->census[i][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            sh[0] = __hip_atomic_load(&bar->census[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh[1] = n;
        }
        __syncthreads();
        blocks_x = sh[0];
        nx = sh[1];
    }
    uint32_t acc = 0;
    u4_t pre[PF];
    bool have_pre = false;
    for (int l = 0; l < layers; ++l) {
        for (int op = op_first; op <= op_last; ++op) {
            const u4_t* base = ops.w[op] + (size_t)l * layer_stride_u4 + lane;
            const int n = ops.kib[op];
            // wave gw owns KiB blocks gw, gw + GW, ... (every wave a strided share; consecutive waves read consecutive KiB)
            int i = gw;
            if (MODE == 2 && have_pre) {
#pragma unroll
                for (int j = 0; j < PF; ++j) acc ^= pre[j].x ^ pre[j].y ^ pre[j].z ^ pre[j].w;
                i += PF * GW;
            }
            for (; i + 3 * GW < n; i += 4 * GW) {
                u4_t v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(base + (size_t)(i + j * GW) * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
            }
            for (; i < n; i += GW) {
                const u4_t v = __builtin_nontemporal_load(base + (size_t)i * 64);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
            // "publish" the op's output: a few bytes per block (write-through)
            if (threadIdx.x == 0) __hip_atomic_store(&out[(op * 256 + blockIdx.x) & 1023], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE >= 1) {
                const bool last = l == layers - 1 && op == op_last;
                if (MODE == 2 && !last) {
                    // next op's first PF KiB of this wave: requested BEFORE the barrier, consumed after it
                    const int nop = op == op_last ? op_first : op + 1;
                    const int nl = op == op_last ? l + 1 : l;
                    const u4_t* nb = ops.w[nop] + (size_t)nl * layer_stride_u4 + lane;
                    const int nn = ops.kib[nop];
#pragma unroll
                    for (int j = 0; j < PF; ++j) {
                        const int ii = gw + j * GW;
                        pre[j] = __builtin_nontemporal_load(nb + (size_t)(ii < nn ? ii : nn - 1) * 64);
                    }
                    have_pre = true;
                }
                if (!last) grid_barrier(bar, x, ++k, blocks_x, nx);
            }
        }
    }
    if (acc == 0x12345678u) out[1023] = acc;
}

extern "C" int layer_probe_launch(int mode, const void* w0, const void* w1, const void* w2, const void* w3, int k0, int k1, int k2, int k3,
                                  size_t layer_stride_bytes, int layers, void* bar, void* out, void* stream) {
    Ops ops;
    ops.w[0] = (const u4_t*)w0; ops.w[1] = (const u4_t*)w1; ops.w[2] = (const u4_t*)w2; ops.w[3] = (const u4_t*)w3;
    ops.kib[0] = k0; ops.kib[1] = k1; ops.kib[2] = k2; ops.kib[3] = k3;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(256), b(1024);
    if (mode == 0) {
        for (int l = 0; l < layers; ++l)
            for (int op = 0; op < 4; ++op) {
                Ops o1 = ops;
                for (int i = 0; i < 4; ++i) o1.w[i] = (const u4_t*)((const char*)ops.w[i] + (size_t)l * layer_stride_bytes);
                hipLaunchKernelGGL((layer_probe<0>), g, b, 0, s, o1, (size_t)0, 1, op, op, (Bar*)bar, (uint32_t*)out);
            }
    } else {
        hipMemsetAsync(bar, 0, sizeof(Bar), s);
        if (mode == 1) hipLaunchKernelGGL((layer_probe<1>), g, b, 0, s, ops, layer_stride_bytes / 16, layers, 0, 3, (Bar*)bar, (uint32_t*)out);
        else hipLaunchKernelGGL((layer_probe<2>), g, b, 0, s, ops, layer_stride_bytes / 16, layers, 0, 3, (Bar*)bar, (uint32_t*)out);
    }
    return (int)hipGetLastError();
}
