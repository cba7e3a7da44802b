// One-shot all-reduce for the tensor-parallel decode step (the 70B config's row-parallel o_proj / down_proj
// exchange M*hidden*4 bytes -- 32 KB at batch 1 -- 160 times per token: latency-bound, where a ring/tree collective pays
// several hops and a library launch per call).  xGMI is point-to-point and every GPU of the node can map every other GPU's
// memory, so the exchange is ONE kernel with no intermediate hop:
//
//   push    rank r stores its fp32 partial vector into slot [parity][r] of EVERY rank's communication buffer (peer-mapped
//           through IPC handles; 7 remote + 1 local 16-byte system-scope stores per 4 elements), drains them, then raises
//           flag [parity][r][block] = epoch in every buffer;
//   wait    each block polls its OWN buffer's `world` flags (local memory, one lane per source rank, bounded);
//   reduce  sums the `world` slots in RANK ORDER (every rank adds the same numbers in the same order: bit-identical results on
//           all ranks, unlike a ring whose association depends on the rank), then the reference's rounding chain in the same
//           pass: y = act(sum); y = act(y + bias); out = act(residual + y)  (torch.py:337-342 + the caller's residual add).
//
// Epochs are counted in device memory (one word per block, owned by that block), so a captured launch replays correctly;
// slots are double-buffered by epoch parity: a rank can only enter call t+2 after every peer has pushed call t+1, i.e. has
// finished reading call t.  The buffers are allocated UNCACHED (fine-grained) by gptqhip_comm_alloc so that neither the
// writer's nor the reader's L2 can hold a stale line; all cross-rank accesses are system-scope.
//
// No reference interface is replaced: the reference has no tensor parallelism.  NOT yet run across more than one
// physical GPU (the build / test boxes have one): tests/test_gpu_comm.py drives the protocol with two processes that share
// GPU 0 through real IPC mappings.
#include <string.h>

#include "../../include/gptqhip.h"
#include "gptqhip_device.h"
#include "gptqhip_host.h"

namespace gptqhip {

constexpr int kCommMaxWorld = 8;
constexpr int kCommMaxBlocks = 64;
constexpr int kCommBlockFloats = 1024;  // 256 threads x 4 floats
constexpr unsigned kCommMaxSpins = 1u << 22;

struct CommHeader {
    uint32_t epoch[kCommMaxBlocks];                          // owned by block b of the LOCAL rank
    uint32_t status;                                         // |= 1 when a bounded wait gave up
    uint32_t pad[63];
    uint32_t flags[2][kCommMaxWorld][kCommMaxBlocks];        // written by peers
};

__host__ __device__ inline size_t comm_data_offset() { return (sizeof(CommHeader) + 255) / 256 * 256; }
__host__ __device__ inline size_t comm_slot_floats(int n_max) { return (size_t)(n_max + kCommBlockFloats - 1) / kCommBlockFloats * kCommBlockFloats; }

struct PeerTable {
    char* base[kCommMaxWorld];
};

__device__ __forceinline__ void store16_system(float* dst, f4_t v) {
    // two 8-byte system-scope stores (sc0 sc1: write-through to the destination's memory, never parked in a local L2)
    unsigned long long* q = reinterpret_cast<unsigned long long*>(dst);
    const u4_t u = __builtin_bit_cast(u4_t, v);
    __hip_atomic_store(q, (unsigned long long)u.x | ((unsigned long long)u.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(q + 1, (unsigned long long)u.z | ((unsigned long long)u.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ f4_t load16_system(const float* src) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<float*>(src));
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const u4_t u = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    return __builtin_bit_cast(f4_t, u);
}

template <int ACT>
__global__ __launch_bounds__(256) void allreduce_oneshot_kernel(const float* __restrict__ partial, PeerTable peers, int rank, int world,
                                                                int n, size_t slot_floats, const void* __restrict__ bias,
                                                                const void* __restrict__ residual, void* __restrict__ out,
                                                                float* __restrict__ stats_out) {
    __shared__ uint32_t s_epoch;
    const int b = blockIdx.x, tid = threadIdx.x;
    CommHeader* mine = reinterpret_cast<CommHeader*>(peers.base[rank]);
    if (tid == 0) s_epoch = __hip_atomic_load(&mine->epoch[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
    __syncthreads();
    const uint32_t e = s_epoch;
    const int par = (int)(e & 1u);
    const int i = b * kCommBlockFloats + tid * 4;  // n % 4 == 0 (checked by the host)
    const bool live = i < n;

    // ---- push -------------------------------------------------------------------------------------------------------
    if (live) {
        const f4_t v = *reinterpret_cast<const f4_t*>(partial + i);
        for (int p = 0; p < world; ++p) {
            float* dst = reinterpret_cast<float*>(peers.base[p] + comm_data_offset()) + ((size_t)par * kCommMaxWorld + rank) * slot_floats + i;
            store16_system(dst, v);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its remote writes
    __syncthreads();
    if (tid < world) {
        CommHeader* peer = reinterpret_cast<CommHeader*>(peers.base[tid]);
        __hip_atomic_store(&peer->flags[par][rank][b], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

    // ---- wait: lane q polls "rank q has pushed block b of this epoch" in the LOCAL buffer ---------------------------------
    if (tid < world) {
        unsigned spins = 0;
        while (__hip_atomic_load(&mine->flags[par][tid][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            if (++spins > kCommMaxSpins) {
                atomicOr(&mine->status, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();

    // ---- reduce in rank order + the reference's rounding chain ----------------------------------------------------------
    float sq = 0.f;
    if (live) {
        const float* slots = reinterpret_cast<const float*>(peers.base[rank] + comm_data_offset()) + (size_t)par * kCommMaxWorld * slot_floats + i;
        f4_t s = load16_system(slots);
        for (int q = 1; q < world; ++q) s += load16_system(slots + (size_t)q * slot_floats);
        uint16_t r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = round_through<ACT>(s[j]);
            if (bias != nullptr) y = round_through<ACT>(y + load16_as_f32<ACT>(bias, (size_t)i + j));
            if (residual != nullptr) y = load16_as_f32<ACT>(residual, (size_t)i + j) + y;
            r[j] = f32_to_16<ACT>(y);
            const float h = bits16_to_f32<ACT>(r[j]);
            sq = __builtin_fmaf(h, h, sq);
        }
        u2_t ov;
        ov.x = (uint32_t)r[0] | ((uint32_t)r[1] << 16);
        ov.y = (uint32_t)r[2] | ((uint32_t)r[3] << 16);
        *reinterpret_cast<u2_t*>(reinterpret_cast<uint16_t*>(out) + i) = ov;
    }
    if (stats_out != nullptr) {
        // RMSNorm statistic of the op that consumes `out` (gptqhip_decode_linear stats_in): sum of out^2 per 16 outputs = 4 lanes
        sq += __shfl_xor(sq, 1, 64);
        sq += __shfl_xor(sq, 2, 64);
        if (live && (tid & 3) == 0) stats_out[i >> 4] = sq;
    }
    if (tid == 0) __hip_atomic_store(&mine->epoch[b], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace gptqhip

using namespace gptqhip;

extern "C" {

size_t gptqhip_comm_bytes(int world, int n_max) {
    if (world < 1 || world > kCommMaxWorld || n_max <= 0 || (size_t)n_max > (size_t)kCommMaxBlocks * kCommBlockFloats) return 0;
    return comm_data_offset() + (size_t)2 * kCommMaxWorld * comm_slot_floats(n_max) * sizeof(float);
}

int gptqhip_comm_alloc(size_t bytes, void** dev_ptr, unsigned char* handle_out) {
    if (!dev_ptr || !handle_out || bytes == 0) {
        set_error("gptqhip_comm_alloc: bad arguments");
        return GPTQHIP_EINVAL;
    }
    void* p = nullptr;
    // fine-grained / uncached: cross-GPU stores and local polls must never be served from a stale L2 line
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    }
    int rc = check_hip(e, "gptqhip_comm_alloc: hipExtMallocWithFlags");
    if (rc) return rc;
    rc = check_hip(hipMemset(p, 0, bytes), "gptqhip_comm_alloc: hipMemset");
    if (rc) return rc;
    rc = check_hip(hipDeviceSynchronize(), "gptqhip_comm_alloc: sync");
    if (rc) return rc;
    hipIpcMemHandle_t h;
    rc = check_hip(hipIpcGetMemHandle(&h, p), "gptqhip_comm_alloc: hipIpcGetMemHandle");
    if (rc) {
        (void)hipFree(p);
        return rc;
    }
    static_assert(sizeof(hipIpcMemHandle_t) == GPTQHIP_IPC_HANDLE_BYTES, "IPC handle size");
    memcpy(handle_out, &h, sizeof(h));
    *dev_ptr = p;
    return GPTQHIP_OK;
}

int gptqhip_comm_open(const unsigned char* handle, void** dev_ptr) {
    if (!handle || !dev_ptr) {
        set_error("gptqhip_comm_open: bad arguments");
        return GPTQHIP_EINVAL;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    int rc = check_hip(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "gptqhip_comm_open: hipIpcOpenMemHandle");
    if (rc) return rc;
    *dev_ptr = p;
    return GPTQHIP_OK;
}

int gptqhip_comm_close(void* dev_ptr) { return check_hip(hipIpcCloseMemHandle(dev_ptr), "gptqhip_comm_close"); }

int gptqhip_comm_free(void* dev_ptr) { return check_hip(hipFree(dev_ptr), "gptqhip_comm_free"); }

int gptqhip_comm_status(void* own_buf, uint32_t* status_out) {
    if (!own_buf || !status_out) {
        set_error("gptqhip_comm_status: bad arguments");
        return GPTQHIP_EINVAL;
    }
    return check_hip(hipMemcpy(status_out, &reinterpret_cast<CommHeader*>(own_buf)->status, sizeof(uint32_t), hipMemcpyDeviceToHost),
                     "gptqhip_comm_status");
}

int gptqhip_allreduce_oneshot(const float* partial, void* const* peer_bufs, int rank, int world, int n, int n_max, const void* bias,
                              const void* residual, void* out, float* stats_out, int act_dtype, gptqhip_stream_t stream) {
    if (!partial || !peer_bufs || !out || world < 1 || world > kCommMaxWorld || rank < 0 || rank >= world || n <= 0 || n % 4 != 0 ||
        n > n_max || gptqhip_comm_bytes(world, n_max) == 0 || (stats_out != nullptr && n % 16 != 0)) {
        set_error("gptqhip_allreduce_oneshot: bad arguments (world <= %d, n %% 4 == 0, n <= n_max <= %d)", kCommMaxWorld,
                  kCommMaxBlocks * kCommBlockFloats);
        return GPTQHIP_EINVAL;
    }
    if (act_dtype != GPTQHIP_FP16 && act_dtype != GPTQHIP_BF16) {
        set_error("gptqhip_allreduce_oneshot: act_dtype must be GPTQHIP_FP16/BF16");
        return GPTQHIP_EINVAL;
    }
    PeerTable t;
    for (int p = 0; p < kCommMaxWorld; ++p) t.base[p] = p < world ? reinterpret_cast<char*>(peer_bufs[p]) : nullptr;
    for (int p = 0; p < world; ++p) {
        if (!t.base[p]) {
            set_error("gptqhip_allreduce_oneshot: peer buffer %d is NULL", p);
            return GPTQHIP_EINVAL;
        }
    }
    const dim3 grid(ceil_div(n, kCommBlockFloats)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (act_dtype == GPTQHIP_FP16) {
        hipLaunchKernelGGL((allreduce_oneshot_kernel<kFP16>), grid, block, 0, s, partial, t, rank, world, n, comm_slot_floats(n_max), bias, residual, out, stats_out);
    } else {
        hipLaunchKernelGGL((allreduce_oneshot_kernel<kBF16>), grid, block, 0, s, partial, t, rank, world, n, comm_slot_floats(n_max), bias, residual, out, stats_out);
    }
    return check_hip(hipGetLastError(), "allreduce_oneshot_kernel launch");
}

}  // extern "C"
