// One-shot all-reduce for the tensor-parallel decode step (the 70B config's row-parallel o_proj / down_proj
// exchange M*hidden*4 bytes -- 32 KB at batch 1 -- 160 times per token: latency-bound, where a ring/tree collective pays
// several hops and a library launch per call).  xGMI is point-to-point and every GPU of the node can map every other GPU's
// memory, so the exchange is ONE kernel with no intermediate hop:
//
//   push    rank r stores its fp32 partial vector into slot [parity][r] of EVERY rank's communication buffer (peer-mapped
//           through IPC handles; 7 remote + 1 local 16-byte system-scope write-through stores per 4 elements); every storing
//           wave then executes a system-scope RELEASE fence (buffer_wbl2 sc0 sc1 + s_waitcnt vmcnt(0): all its stores have been
//           acknowledged by the destination), the block meets at a barrier, and `world` lanes raise flag [parity][r][block] =
//           epoch in every buffer with a system-scope store-RELEASE;
//   wait    each block polls its OWN buffer's `world` flags (local memory, one lane per source rank), bounded by the 100 MHz
//           wall clock (CommHeader::timeout_ticks), then executes a system-scope ACQUIRE fence (buffer_inv sc0 sc1) before the
//           block barrier that lets the other waves read the slots (which they do with system-scope loads);
//   reduce  sums the `world` slots in RANK ORDER (every rank adds the same numbers in the same order: bit-identical results on
//           all ranks, unlike a ring whose association depends on the rank), then the reference's rounding chain in the same
//           pass: y = act(sum); y = act(y + bias); out = act(residual + y)  (torch.py:337-342 + the caller's residual add).
//
// The flag edge is therefore a textbook message-passing pattern in the HSA / LLVM-AMDGPU memory model: data stores
// happen-before the release store of the flag (same wave: program order + fence; other waves: fence, workgroup barrier, release
// store), the acquire side observes the flag and then the data.  A wait that exceeds its bound does NOT reduce stale slots: the
// block writes NaN to its part of `out` (and of stats_out) and sets the sticky status word (gptqhip_comm_status), so a lost peer
// is loud -- downstream activations become NaN -- instead of silently corrupt.
//
// Epochs are counted in device memory (one word per block, owned by that block), so a captured launch replays correctly;
// slots are double-buffered by epoch parity: a rank can only enter call t+2 after every peer has pushed call t+1, i.e. has
// finished reading call t.  The buffers are allocated UNCACHED (fine-grained) by gptqhip_comm_alloc so that neither the
// writer's nor the reader's L2 can hold a stale line; all cross-rank accesses are system-scope.
//
// No reference interface is replaced: the reference has no tensor parallelism.  Hardware status: the build / test
// boxes have ONE GPU; tests/test_gpu_comm.py drives the protocol with two processes that share GPU 0 through real IPC mappings
// (incl. a 10^5-epoch back-to-back stress with per-epoch payloads).  utils.xgmi_allreduce.OneShotAllReduce.self_test() validates a
// freshly built communicator against the process group's own collective before anybody relies on it, and bench.py falls back to
// RCCL when that fails -- the first contact with real xGMI links is therefore checked at run time.
#include <stdlib.h>
#include <string.h>

#include "../../include/gptqhip.h"
#include "gptqhip_device.h"
#include "gptqhip_host.h"

namespace gptqhip {

constexpr int kCommMaxWorld = 8;
constexpr int kCommMaxBlocks = 64;
constexpr int kCommBlockFloats = 1024;  // 256 threads x 4 floats
constexpr unsigned long long kCommDefaultTimeoutTicks = 10ull * 100000000ull;   // 10 s of the 100 MHz wall clock

struct CommHeader {
    uint32_t epoch[kCommMaxBlocks];                          // owned by block b of the LOCAL rank
    uint32_t status;                                         // |= 1 when a bounded wait gave up
    uint32_t timeout_lo, timeout_hi;                         // wait bound in wall-clock ticks (100 MHz), set by gptqhip_comm_alloc
    uint32_t gepoch;                                         // all-gather calls (gptqhip_allgather_select): their own epoch, flags and
    uint32_t gflags[2][kCommMaxWorld];                       //   slot region, so the two collectives never alias each other's parity
    uint32_t pad[44];
    uint32_t flags[2][kCommMaxWorld][kCommMaxBlocks];        // written by peers
};

__host__ __device__ inline size_t comm_data_offset() { return (sizeof(CommHeader) + 255) / 256 * 256; }
__host__ __device__ inline size_t comm_slot_floats(int n_max) { return (size_t)(n_max + kCommBlockFloats - 1) / kCommBlockFloats * kCommBlockFloats; }
// all-gather region behind the all-reduce slots: [2 parities][world ranks] slots of n_max 16-bit elements each
__host__ __device__ inline size_t comm_gather_offset(int n_max) { return comm_data_offset() + (size_t)2 * kCommMaxWorld * comm_slot_floats(n_max) * sizeof(float); }
__host__ __device__ inline size_t comm_gslot_bytes(int n_max) { return ((size_t)n_max * 2 + 255) / 256 * 256; }
static_assert(sizeof(CommHeader) == (64 + 64 + 2 * 8 * 64) * 4, "CommHeader layout");

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
    __shared__ uint32_t s_lost;
    const int b = blockIdx.x, tid = threadIdx.x;
    CommHeader* mine = reinterpret_cast<CommHeader*>(peers.base[rank]);
    if (tid == 0) {
        s_epoch = __hip_atomic_load(&mine->epoch[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
        s_lost = 0u;
    }
    __syncthreads();
    const uint32_t e = s_epoch;
    const int par = (int)(e & 1u);
    const int i = b * kCommBlockFloats + tid * 4;  // n % 4 == 0 (checked by the host)
    const bool live = i < n;

    // bias / residual do not depend on the peers: requested up front as aligned 8-byte words, their latency hides under the push
    u2_t braw = {0u, 0u}, rraw = {0u, 0u};
    if (live && bias != nullptr) braw = *reinterpret_cast<const u2_t*>(reinterpret_cast<const uint16_t*>(bias) + i);
    if (live && residual != nullptr) rraw = *reinterpret_cast<const u2_t*>(reinterpret_cast<const uint16_t*>(residual) + i);

    // ---- push -------------------------------------------------------------------------------------------------------
    if (live) {
        const f4_t v = *reinterpret_cast<const f4_t*>(partial + i);
        for (int p = 0; p < world; ++p) {
            float* dst = reinterpret_cast<float*>(peers.base[p] + comm_data_offset()) + ((size_t)par * kCommMaxWorld + rank) * slot_floats + i;
            store16_system(dst, v);
        }
    }
    // every storing wave: system-scope release fence (write-back + wait until its remote stores are acknowledged) ...
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    // ... then the flag, itself a system-scope store-release (ordered after everything the barrier collected)
    if (tid < world) {
        CommHeader* peer = reinterpret_cast<CommHeader*>(peers.base[tid]);
        __hip_atomic_store(&peer->flags[par][rank][b], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }

    // ---- wait: lane q polls "rank q has pushed block b of this epoch" in the LOCAL buffer ---------------------------------
    if (tid < world) {
        const unsigned long long limit = (unsigned long long)mine->timeout_lo | ((unsigned long long)mine->timeout_hi << 32);
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(&mine->flags[par][tid][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            if ((++spins & 255u) == 0u && wall_clock64() - t0 > limit) {
                atomicOr(&mine->status, 1u);
                atomicOr(&s_lost, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: nothing read below may come from a line cached before the flags
    __syncthreads();
    const bool lost = s_lost != 0u;

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
            if (bias != nullptr) y = round_through<ACT>(y + bits16_to_f32<ACT>((uint16_t)(braw[j >> 1] >> ((j & 1) * 16))));
            if (residual != nullptr) y = bits16_to_f32<ACT>((uint16_t)(rraw[j >> 1] >> ((j & 1) * 16))) + y;
            if (lost) y = __builtin_nanf("");   // a peer never arrived: poison, do not publish a sum of stale slots
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

// One-shot ALL-GATHER + select for act-order row-parallel shards: a row shard cut from the GLOBALLY group-sorted
// rows (utils.tp.shard_gptq_row(act_order="global_sort"), the Marlin rule gptqmodel/utils/marlin.py:296-305,368-372) needs the input
// features index[0..n_out) of the FULL activation vector, which is scattered over all ranks' column shards (attention heads).  Same
// protocol as above with its own epoch / flags / slots: every rank pushes its n_local 16-bit elements into every rank's buffer,
// release -> flag -> acquire, then out[j] = full[index[j]] straight from the local slots (index NULL: the whole vector).  One block:
// the message is <= 16 KB and every output element may come from any rank.
__global__ __launch_bounds__(256) void allgather_select_kernel(const uint16_t* __restrict__ x_local, PeerTable peers, int rank, int world,
                                                               int n_local, size_t gather_off, size_t gslot_bytes,
                                                               const int32_t* __restrict__ index, int n_out, uint16_t* __restrict__ out,
                                                               uint32_t nan_bits) {
    __shared__ uint32_t s_epoch;
    __shared__ uint32_t s_lost;
    const int tid = threadIdx.x;
    CommHeader* mine = reinterpret_cast<CommHeader*>(peers.base[rank]);
    if (tid == 0) {
        s_epoch = __hip_atomic_load(&mine->gepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
        s_lost = 0u;
    }
    __syncthreads();
    const uint32_t e = s_epoch;
    const int par = (int)(e & 1u);
    for (int i = tid * 8; i < n_local; i += 256 * 8) {   // n_local % 8 == 0 (host-checked): 16 bytes per lane
        const u4_t v = *reinterpret_cast<const u4_t*>(x_local + i);
        for (int p = 0; p < world; ++p) {
            char* slot = peers.base[p] + gather_off + ((size_t)par * kCommMaxWorld + rank) * gslot_bytes;
            store16_system(reinterpret_cast<float*>(slot + (size_t)i * 2), __builtin_bit_cast(f4_t, v));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (tid < world) {
        CommHeader* peer = reinterpret_cast<CommHeader*>(peers.base[tid]);
        __hip_atomic_store(&peer->gflags[par][rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long limit = (unsigned long long)mine->timeout_lo | ((unsigned long long)mine->timeout_hi << 32);
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(&mine->gflags[par][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            if ((++spins & 255u) == 0u && wall_clock64() - t0 > limit) {
                atomicOr(&mine->status, 1u);
                atomicOr(&s_lost, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    const bool lost = s_lost != 0u;
    const char* slots = peers.base[rank] + gather_off + (size_t)par * kCommMaxWorld * gslot_bytes;
    for (int j = tid; j < n_out; j += 256) {
        const int src = index != nullptr ? index[j] : j;
        // an index outside the concatenation would read another slot region or past the buffer: poison the element and raise the
        // sticky status word (status bit 2), like a lost peer -- never a silent wrong value
        if ((unsigned)src >= (unsigned)(n_local * world)) {
            out[j] = (uint16_t)nan_bits;
            atomicOr(&mine->status, 2u);
            continue;
        }
        const int r = src / n_local, off = src - r * n_local;
        uint16_t* q = reinterpret_cast<uint16_t*>(const_cast<char*>(slots) + (size_t)r * gslot_bytes) + off;
        const uint16_t v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        out[j] = lost ? (uint16_t)nan_bits : v;
    }
    if (tid == 0) __hip_atomic_store(&mine->gepoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace gptqhip

using namespace gptqhip;

extern "C" {

size_t gptqhip_comm_bytes(int world, int n_max) {
    if (world < 1 || world > kCommMaxWorld || n_max <= 0 || (size_t)n_max > (size_t)kCommMaxBlocks * kCommBlockFloats) return 0;
    return comm_gather_offset(n_max) + (size_t)2 * kCommMaxWorld * comm_gslot_bytes(n_max);
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
    {
        // wait bound of the kernels that poll this buffer (GPTQHIP_COMM_TIMEOUT_MS, default 10 s; 100 MHz wall clock)
        unsigned long long ticks = kCommDefaultTimeoutTicks;
        const char* v = getenv("GPTQHIP_COMM_TIMEOUT_MS");
        if (v && *v && atoll(v) > 0) ticks = (unsigned long long)atoll(v) * 100000ull;
        const uint32_t t[2] = {(uint32_t)ticks, (uint32_t)(ticks >> 32)};
        rc = check_hip(hipMemcpy(&reinterpret_cast<CommHeader*>(p)->timeout_lo, t, sizeof(t), hipMemcpyHostToDevice),
                       "gptqhip_comm_alloc: timeout");
        if (rc) return rc;
    }
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

int gptqhip_comm_set_timeout(void* own_buf, unsigned int timeout_ms) {
    if (!own_buf || timeout_ms == 0) {
        set_error("gptqhip_comm_set_timeout: bad arguments");
        return GPTQHIP_EINVAL;
    }
    const unsigned long long ticks = (unsigned long long)timeout_ms * 100000ull;     // 100 MHz wall clock
    const uint32_t t[2] = {(uint32_t)ticks, (uint32_t)(ticks >> 32)};
    return check_hip(hipMemcpy(&reinterpret_cast<CommHeader*>(own_buf)->timeout_lo, t, sizeof(t), hipMemcpyHostToDevice),
                     "gptqhip_comm_set_timeout");
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
    if (((uintptr_t)partial & 15u) || ((uintptr_t)out & 7u) || ((uintptr_t)bias & 7u) || ((uintptr_t)residual & 7u)) {
        set_error("gptqhip_allreduce_oneshot: partial must be 16-byte aligned, out / bias / residual 8-byte aligned");
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

int gptqhip_allgather_select(const void* x_local, void* const* peer_bufs, int rank, int world, int n_local, int n_max,
                             const int32_t* index, int n_out, void* out, int act_dtype, gptqhip_stream_t stream) {
    if (!x_local || !peer_bufs || !out || world < 1 || world > kCommMaxWorld || rank < 0 || rank >= world || n_local <= 0 ||
        n_local % 8 != 0 || n_local > n_max || n_out <= 0 || gptqhip_comm_bytes(world, n_max) == 0 || ((uintptr_t)x_local & 15u)) {
        set_error("gptqhip_allgather_select: bad arguments (world <= %d, n_local %% 8 == 0, n_local <= n_max, x_local 16-byte aligned)",
                  kCommMaxWorld);
        return GPTQHIP_EINVAL;
    }
    if (index == nullptr && n_out != n_local * world) {
        set_error("gptqhip_allgather_select: without an index the output is the whole vector (n_out = n_local * world)");
        return GPTQHIP_EINVAL;
    }
    if (act_dtype != GPTQHIP_FP16 && act_dtype != GPTQHIP_BF16) {
        set_error("gptqhip_allgather_select: act_dtype must be GPTQHIP_FP16/BF16");
        return GPTQHIP_EINVAL;
    }
    PeerTable t;
    for (int p = 0; p < kCommMaxWorld; ++p) t.base[p] = p < world ? reinterpret_cast<char*>(peer_bufs[p]) : nullptr;
    for (int p = 0; p < world; ++p) {
        if (!t.base[p]) {
            set_error("gptqhip_allgather_select: peer buffer %d is NULL", p);
            return GPTQHIP_EINVAL;
        }
    }
    hipLaunchKernelGGL(allgather_select_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint16_t*>(x_local), t, rank, world, n_local, comm_gather_offset(n_max), comm_gslot_bytes(n_max),
                       index, n_out, reinterpret_cast<uint16_t*>(out), act_dtype == GPTQHIP_FP16 ? 0x7E00u : 0x7FC0u);
    return check_hip(hipGetLastError(), "allgather_select_kernel launch");
}

}  // extern "C"
