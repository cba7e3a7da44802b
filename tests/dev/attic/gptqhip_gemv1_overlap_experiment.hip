// Batch-1 decode GEMV with fused decoder-layer glue and in-launch dependency flags ("decode chain" op).
//
// Why a second decode kernel: profiles/r01_* show the skinny kernel streams at the HBM ceiling INSIDE a launch but loses
// ~3.5 us per launch to the dependent-kernel boundary plus its own ramp (first HBM round trip) and tail (reduce, store):
// 45 % of a Llama-3-8B token.  A real decoder is a dependency CHAIN (qkv -> attention -> o -> norm -> gate_up -> act ->
// down -> norm -> ...), so consecutive linears cannot simply run side by side.  This kernel makes the dependency
// explicit and fine-grained instead of relying on the stream order:
//   * every op is launched with a FIXED small footprint -- at most one 16-wave block per CU, <= 64 VGPRs, < 80 KiB of
//     LDS -- so TWO consecutive ops are always co-resident on the chip (2 x 16 waves = the CU's 32 wave slots);
//   * op i+1 is enqueued on a second stream and starts immediately: it issues its first D KiB-blocks of PACKED WEIGHTS per
//     wave (they depend on nothing), then ONE wave polls op i's arrival counters (relaxed agent-scope loads + s_sleep,
//     bounded), and only then reads op i's output; by then its weights sit in registers;
//   * op i publishes its outputs with 8-byte agent-scope (write-through, sc1) stores and arrives on a sharded counter
//     (64 words, one 256-B line) after draining them; the consumer reads them with 8-byte agent-scope (L1-bypassing)
//     loads -- "agent atomics on both sides" (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md valid forms), so
//     no release / acquire fence (~1.7 us each) sits on the dependency edge.
//   Deadlock freedom does not depend on dispatch order: a graph holds two chains (even ops / odd ops), so at most ops
//   {i, i+1} are in flight, each at most 256 blocks of half a CU -- the spinning op can never occupy more than half of
//   the chip's 512 block slots, the producer always finds room, and every spin is bounded (status word set on give-up).
//   * the elementwise glue between the linears rides in the GEMV instead of separate launches: RMSNorm or SiLU(gate)*up
//     while the input vector is staged into LDS (after the dependency resolves, under the weight prefetch), and the
//     residual add in the epilogue.  Semantics = HF LlamaRMSNorm / LlamaMLP / residual adds in the activation dtype.
//
// Arithmetic is the skinny kernel's: tile-major words are MFMA B fragments, dequantised in registers with the reference's
// single rounding (gptqhip_device.h), fp32 accumulation, one rounding of the result, then bias / residual adds each
// rounded in the activation dtype like torch does (torch.py:326-347).
//
// Work split: block b owns column tiles b, b+grid, b+2*grid, ...; its W waves split every tile's K range (chunk =
// wave + i*W).  A wave walks the flat sequence of its (tile, chunk) units with ONE D-deep register ring that runs across
// tile boundaries, so the HBM stream never drains between tiles; the W partial sums of a tile meet in a double-buffered
// 64-byte-per-wave LDS slot (batch 1: only accumulator row 0 is live) behind a single plain s_barrier; the reducing wave
// rotates from tile to tile.
#include "gptqhip_device.h"
#include "gptqhip_host.h"

namespace gptqhip {

constexpr int kGlueNone = 0;
constexpr int kGlueRmsNorm = 1;
constexpr int kGlueSiluMul = 2;
constexpr int kCounterShards = 64;        // arrival counters per op: one 256-byte line, polled by one wave in one load
constexpr unsigned kMaxSpins = 1u << 17;  // ~0.1-0.3 s of polling, then give up loudly (status word) instead of hanging

struct Gemv1Params {
    const uint32_t* qw;
    const uint32_t* meta;
    const void* bias;
    const void* x;         // [K] (none / rmsnorm) or [2K] gate|up (silu_mul), activation dtype
    const void* norm_w;    // [K] RMSNorm weight
    const void* residual;  // [N] or nullptr
    void* out;             // [N]
    uint32_t* wait_ctr;    // [64] or nullptr
    uint32_t* signal_ctr;  // [64] or nullptr
    uint32_t* status;      // device word, |= 1 when a bounded spin gave up
    uint32_t wait_total;
    float eps;
    int K, N, G;
    int chunks;  // K / 128
    int tiles;   // ceil(N / 16)
    int cpw;     // chunks per wave per tile = chunks / W
    int cpg_shift;
    int in_glue;
};

template <int BITS>
struct RingStage {
    u4_t w[BITS == 4 ? 1 : 2];
    uint32_t meta;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
}

// 16 bytes of a vector another op of the SAME launch window may have just written: two 8-byte agent-scope loads
// (L1-bypassing; plain loads could hit a stale L1 line of a previous token)
__device__ __forceinline__ u4_t load16_agent(const void* base, int i) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<void*>(base)) + 2 * (size_t)i;
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return u4_t{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
}

template <int ACT>
__device__ __forceinline__ float lo16(uint32_t u) { return bits16_to_f32<ACT>((uint16_t)(u & 0xffffu)); }
template <int ACT>
__device__ __forceinline__ float hi16(uint32_t u) { return bits16_to_f32<ACT>((uint16_t)(u >> 16)); }
template <int ACT>
__device__ __forceinline__ uint32_t pack16(float a, float b) {
    return (uint32_t)f32_to_16<ACT>(a) | ((uint32_t)f32_to_16<ACT>(b) << 16);
}

// ---- input staging: x (with its glue) -> LDS, activation dtype, natural k order, zero padded to chunks*128 -------------
template <int ACT>
__device__ __forceinline__ void stage_input(const Gemv1Params& p, u4_t* xbuf, float* scratch) {
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    const int n16 = p.K >> 3;            // 16-byte pieces of the vector (K % 8 == 0)
    const int n16p = p.chunks * 16;      // padded
    const void* xs = p.x;
    if (p.in_glue == kGlueNone) {
        for (int i = tid; i < n16p; i += nthr) xbuf[i] = i < n16 ? load16_agent(xs, i) : u4_t{0u, 0u, 0u, 0u};
        return;
    }
    if (p.in_glue == kGlueSiluMul) {
        // HF LlamaMLP: act_fn(gate) * up, both in the activation dtype: silu evaluated in fp32 and rounded, then one
        // rounded multiply
        for (int i = tid; i < n16p; i += nthr) {
            u4_t r = {0u, 0u, 0u, 0u};
            if (i < n16) {
                const u4_t g = load16_agent(xs, i), u = load16_agent(xs, n16 + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g0 = lo16<ACT>(g[j]), g1 = hi16<ACT>(g[j]);
                    const float s0 = round_through<ACT>(g0 / (1.0f + __expf(-g0)));
                    const float s1 = round_through<ACT>(g1 / (1.0f + __expf(-g1)));
                    r[j] = pack16<ACT>(s0 * lo16<ACT>(u[j]), s1 * hi16<ACT>(u[j]));
                }
            }
            xbuf[i] = r;
        }
        return;
    }
    // HF LlamaRMSNorm: h32 = h.float(); h32 * rsqrt(mean(h32^2) + eps) -> activation dtype -> * weight (rounded)
    float ss = 0.f;
    for (int i = tid; i < n16; i += nthr) {
        const u4_t h = load16_agent(xs, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = lo16<ACT>(h[j]), b = hi16<ACT>(h[j]);
            ss = __builtin_fmaf(a, a, ss);
            ss = __builtin_fmaf(b, b, ss);
        }
    }
    ss = wave_sum(ss);
    const int wave = tid >> 6, W = nthr >> 6;
    if ((tid & 63) == 0) scratch[wave] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < W; ++w) tot += scratch[w];  // fixed order: deterministic
    const float inv = rsqrtf(tot / (float)p.K + p.eps);
    const u4_t* ws = reinterpret_cast<const u4_t*>(p.norm_w);
    for (int i = tid; i < n16p; i += nthr) {
        u4_t r = {0u, 0u, 0u, 0u};
        if (i < n16) {
            const u4_t h = load16_agent(xs, i), g = ws[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = round_through<ACT>(lo16<ACT>(h[j]) * inv), b = round_through<ACT>(hi16<ACT>(h[j]) * inv);
                r[j] = pack16<ACT>(lo16<ACT>(g[j]) * a, hi16<ACT>(g[j]) * b);
            }
        }
        xbuf[i] = r;
    }
}

// Register-staged variant for vectors of at most 2 x 16 bytes per thread (every Llama-3-8B/70B shape except the 70B
// down_proj): the loads are issued in ONE burst (x / h / gate in `a`, norm weight / up in `b`), so in stream-ordered launches
// they can be issued BEFORE the weight ring and waited for on their own (vmcnt retires in order), and RMSNorm needs no
// second pass over memory.
struct XRegs {
    u4_t a[2], b[2];
};
__device__ __forceinline__ void load_x_regs(const Gemv1Params& p, XRegs& r) {
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    const int n16 = p.K >> 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * nthr;
        const int ci = idx < n16 ? idx : 0;  // clamped (unconditional loads keep the waits countable); masked at use
        r.a[i] = load16_agent(p.x, ci);
        if (p.in_glue == kGlueSiluMul) {
            r.b[i] = load16_agent(p.x, n16 + ci);
        } else if (p.in_glue == kGlueRmsNorm) {
            r.b[i] = reinterpret_cast<const u4_t*>(p.norm_w)[ci];
        }
    }
}
template <int ACT>
__device__ __forceinline__ void stage_from_regs(const Gemv1Params& p, const XRegs& r, u4_t* xbuf, float* scratch) {
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    const int n16 = p.K >> 3, n16p = p.chunks * 16;
    float inv = 0.f;
    if (p.in_glue == kGlueRmsNorm) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (tid + i * nthr < n16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = lo16<ACT>(r.a[i][j]), b = hi16<ACT>(r.a[i][j]);
                    ss = __builtin_fmaf(a, a, ss);
                    ss = __builtin_fmaf(b, b, ss);
                }
            }
        }
        ss = wave_sum(ss);
        if ((tid & 63) == 0) scratch[tid >> 6] = ss;
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < (nthr >> 6); ++w) tot += scratch[w];  // fixed order: deterministic
        inv = rsqrtf(tot / (float)p.K + p.eps);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * nthr;
        if (idx < n16p) {
            u4_t o = {0u, 0u, 0u, 0u};
            if (idx < n16) {
                if (p.in_glue == kGlueNone) {
                    o = r.a[i];
                } else if (p.in_glue == kGlueSiluMul) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float g0 = lo16<ACT>(r.a[i][j]), g1 = hi16<ACT>(r.a[i][j]);
                        const float s0 = round_through<ACT>(g0 / (1.0f + __expf(-g0)));
                        const float s1 = round_through<ACT>(g1 / (1.0f + __expf(-g1)));
                        o[j] = pack16<ACT>(s0 * lo16<ACT>(r.b[i][j]), s1 * hi16<ACT>(r.b[i][j]));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = round_through<ACT>(lo16<ACT>(r.a[i][j]) * inv);
                        const float b = round_through<ACT>(hi16<ACT>(r.a[i][j]) * inv);
                        o[j] = pack16<ACT>(lo16<ACT>(r.b[i][j]) * a, hi16<ACT>(r.b[i][j]) * b);
                    }
                }
            }
            xbuf[idx] = o;
        }
    }
}

// WAIT = 1: the op waits on its producer's arrival counters inside the kernel (two-stream overlap); WAIT = 0: ordinary
// stream-ordered launch -- the input vector is then requested BEFORE the weight ring.
// Register budget: WAIT = 1 ops must fit TWO blocks per CU (<= 64 VGPRs at 16 waves each); a stream-ordered op has the CU
// to itself (128 VGPRs: room for the register-staged input vector next to the ring).
template <int BITS, int ACT, int SCL, int D, int WAIT>
__global__ __launch_bounds__(1024, WAIT ? 8 : 4) void gemv1_kernel(Gemv1Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    // [xbuf: chunks*256 B][red: 2 x 16 waves x 16 floats][scratch: 16 floats][ostage: 16 waves x 16 halves]
    u4_t* xbuf = reinterpret_cast<u4_t*>(lds_raw);
    float* red = reinterpret_cast<float*>(lds_raw + (size_t)p.chunks * 256);
    float* scratch = red + 2 * 16 * 16;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int c = lane & 15, rq = lane >> 4;
    const int b = blockIdx.x, nb = gridDim.x;
    uint16_t* ostage = reinterpret_cast<uint16_t*>(scratch + 16) + wave * 16;  // private to the wave: no cross-wave reuse hazard
    const int ntiles = (p.tiles - b + nb - 1) / nb;  // tiles b, b+nb, ...
    const int cpw = p.cpw;
    const int n = ntiles * cpw;                      // (tile, chunk) units of this wave; the plan guarantees n >= D
    constexpr int WPC = BITS == 4 ? 1 : 2;
    constexpr int kBlockBytes = WPC * 1024;

    // ---- 0. stream-ordered launch: the input vector (L2-resident) is requested first, so staging it does not wait for HBM
    // (8-bit ring stages are twice as large: no registers to spare for the staged vector -> looped staging there)
    const bool xfast = BITS == 4 && (p.K >> 3) <= 2 * (int)blockDim.x && p.chunks * 16 <= 2 * (int)blockDim.x;
    XRegs xr;
    if constexpr (WAIT == 0) {
        if (xfast) load_x_regs(p, xr);
    }

    // ---- 1. weight ring prologue: depends on nothing ------------------------------------------------------------------
    const char* wbase = reinterpret_cast<const char*>(p.qw) + (uint32_t)lane * 16u;
    const char* mbase = reinterpret_cast<const char*>(p.meta) + (uint32_t)c * 4u;
    int lt = 0, lc = 0;  // load cursor: tile index within the block's list, chunk index within the wave's list
    auto issue = [&](RingStage<BITS>& st) {
        const int tile = b + lt * nb;
        const int chunk = wave + lc * W;
        const char* src = wbase + ((size_t)tile * p.chunks + chunk) * kBlockBytes;
#pragma unroll
        for (int h = 0; h < WPC; ++h) st.w[h] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(src + h * 1024));
        st.meta = *reinterpret_cast<const uint32_t*>(mbase + (((size_t)tile * p.G + (chunk >> p.cpg_shift)) << 6));
        if (++lc == cpw) {
            lc = 0;
            ++lt;
        }
    };
    RingStage<BITS> st[D];
#pragma unroll
    for (int d = 0; d < D; ++d) issue(st[d]);

    // ---- 2. dependency: ONE wave polls the producer's arrival counters, then the block reads its output (agent loads) ----
    if constexpr (WAIT == 1) {
        if (wave == 0) {
            unsigned spins = 0;
            for (;;) {
                const uint32_t v = __hip_atomic_load(p.wait_ctr + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wave_sum_u32(v) >= p.wait_total) break;
                if (++spins > kMaxSpins) {
                    if (lane == 0) atomicOr(p.status, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();
    }

    // ---- 3. stage the input vector (+ glue) into LDS --------------------------------------------------------------
    if (xfast) {
        if constexpr (WAIT == 1) load_x_regs(p, xr);
        stage_from_regs<ACT>(p, xr, xbuf, scratch);
    } else {
        stage_input<ACT>(p, xbuf, scratch);
    }
    __syncthreads();

    // ---- 4. ring: compute unit j, refill its stage with unit j + D ---------------------------------------------------
    const DequantConsts dk = make_dequant_consts<BITS>();
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
    int ct = 0, cc = 0;  // compute cursor
    // The reduce + epilogue of tile t belongs to wave (t mod W) -- rotating, so no single wave's weight stream pays for
    // all the tails.  Lanes 0..15 of that wave hold the RAW bits of the tile's residual / bias, loaded one tile ahead and
    // only converted in the epilogue (converting here would make the compiler wait for the load on the spot).
    uint32_t res_raw = 0u, bias_raw = 0u;
    int rdr = 0;  // reducer wave of the tile being accumulated = ct mod W
    auto prefetch_epilogue_operands = [&](int t, int owner) {
        if (wave == owner && lane < 16 && t < ntiles) {
            // the aligned 32-bit pair holding this lane's column (a 16-bit load would be zero-extended by an ALU op right
            // behind it, i.e. waited for immediately); the half is picked in the epilogue.  N % 8 == 0: pairs never straddle N
            const int col = (b + t * nb) * kTileN + lane;
            const size_t pi = (size_t)((col < p.N ? col : 0) >> 1);
            if (p.residual != nullptr) {
                uint32_t* rp = reinterpret_cast<uint32_t*>(const_cast<void*>(p.residual)) + pi;
                res_raw = __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (p.bias != nullptr) bias_raw = reinterpret_cast<const uint32_t*>(p.bias)[pi];
        }
    };
    prefetch_epilogue_operands(0, 0);
    auto compute = [&](const RingStage<BITS>& s) {
        const int chunk = wave + cc * W;
        const ColConst k = expand_meta<BITS, SCL>(s.meta);
        const u4_t* xa = xbuf + chunk * 16 + rq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u4_t bf;
            if constexpr (BITS == 4) {
                bf = dequant_word4<ACT, SCL>(s.w[0][j], k, dk);
            } else {
                bf = dequant_word8<ACT, SCL>(s.w[j >> 1][(j & 1) * 2], s.w[j >> 1][(j & 1) * 2 + 1], k, dk);
            }
            acc = mfma16<ACT>(xa[4 * j], bf, acc);  // all 16 fragment rows read x (broadcast): row 0 is the live one
        }
        if (++cc == cpw) {
            // tile done for this wave: partial of accumulator row 0 (lanes 0..15, register 0) -> LDS, one plain barrier
            float* slot = red + ((ct & 1) * 16 + wave) * 16;
            if (lane < 16) slot[lane] = acc[0];
            acc = f4_t{0.f, 0.f, 0.f, 0.f};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int nxt = rdr + 1 == W ? 0 : rdr + 1;
            if (wave == rdr) {
                const int tile = b + ct * nb;
                if (lane < 16) {
                    float v = 0.f;
                    const float* r0 = red + (ct & 1) * 256 + lane;
                    for (int w = 0; w < W; ++w) v += r0[w * 16];  // fixed order
                    float y = round_through<ACT>(v);
                    const int hs = (lane & 1) * 16;
                    if (p.bias != nullptr) y = round_through<ACT>(y + bits16_to_f32<ACT>((uint16_t)(bias_raw >> hs)));
                    if (p.residual != nullptr) y = bits16_to_f32<ACT>((uint16_t)(res_raw >> hs)) + y;  // hidden = residual + hidden, rounded by the store below
                    ostage[lane] = f32_to_16<ACT>(y);
                }
                // same-wave LDS accesses execute in order: read the 32 bytes back as four 8-byte write-through stores
                if (lane < 4 && tile * kTileN + lane * 4 < p.N) {
                    const unsigned long long v8 = reinterpret_cast<const unsigned long long*>(ostage)[lane];
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(reinterpret_cast<uint16_t*>(p.out) + (size_t)tile * kTileN) + lane,
                                       v8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            prefetch_epilogue_operands(ct + 1, nxt);
            rdr = nxt;
            cc = 0;
            ++ct;
        }
    };

    const int rounds = n / D, rem = n - rounds * D;
    for (int r = 0; r + 1 < rounds; ++r) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            compute(st[d]);
            issue(st[d]);
        }
    }
    // last full round: only `rem` more units exist
#pragma unroll
    for (int d = 0; d < D; ++d) {
        compute(st[d]);
        if (d < rem) issue(st[d]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (d < rem) compute(st[d]);
    }

    // ---- 5. arrive: every wave that stored outputs (write-through) drains them, then ONE relaxed agent-scope add ---------
    if (p.signal_ctr != nullptr) {  // (also allowed without WAIT: the first op of a chain only signals)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(p.signal_ctr + (b & (kCounterShards - 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
Gemv1Plan plan_gemv1(int K, int N, int group_size, int cu_count) {
    Gemv1Plan pl;
    pl.ok = 0;
    if (K % kChunkK != 0 || group_size % kChunkK != 0) return pl;
    const int cpg = group_size / kChunkK;
    if ((cpg & (cpg - 1)) != 0) return pl;
    pl.chunks = K / kChunkK;
    pl.tiles = ceil_div(N, kTileN);
    int w = 1;
    for (int cand = 16; cand >= 1; --cand) {
        if (pl.chunks % cand == 0) {
            w = cand;
            break;
        }
    }
    if (w < 4) return pl;  // odd K: too little memory-level parallelism per block, the skinny kernel handles it
    pl.waves = w;
    pl.cpw = pl.chunks / w;
    pl.grid = pl.tiles < cu_count ? pl.tiles : cu_count;
    const int min_units = (pl.tiles / pl.grid) * pl.cpw;
    pl.depth = min_units >= 4 ? 4 : (min_units >= 2 ? 2 : 1);
    pl.lds_bytes = (size_t)pl.chunks * 256 + 2 * 16 * 16 * 4 + 16 * 4 + 16 * 32;
    if (pl.lds_bytes > 72 * 1024) return pl;  // two ops must stay co-resident per CU (160 KiB of LDS)
    pl.ok = 1;
    return pl;
}

template <int BITS, int ACT, int SCL, int WAIT>
static int launch_gemv1_depth(const Gemv1Params& p, const Gemv1Plan& pl, hipStream_t stream) {
    const dim3 grid(pl.grid), block(64 * pl.waves);
    if constexpr (BITS == 4) {  // 8-bit stages are 9 VGPRs each: four of them spill beside the 64-VGPR co-residency budget
        if (pl.depth == 4) {
            hipLaunchKernelGGL((gemv1_kernel<BITS, ACT, SCL, 4, WAIT>), grid, block, pl.lds_bytes, stream, p);
            return check_hip(hipGetLastError(), "gemv1_kernel launch");
        }
    }
    if (pl.depth >= 2) {
        hipLaunchKernelGGL((gemv1_kernel<BITS, ACT, SCL, 2, WAIT>), grid, block, pl.lds_bytes, stream, p);
    } else {
        hipLaunchKernelGGL((gemv1_kernel<BITS, ACT, SCL, 1, WAIT>), grid, block, pl.lds_bytes, stream, p);
    }
    return check_hip(hipGetLastError(), "gemv1_kernel launch");
}

int launch_gemv1(const DecodeArgs& a, const Gemv1Plan& pl, hipStream_t stream) {
    Gemv1Params p;
    p.qw = a.qweight;
    p.meta = a.meta;
    p.bias = a.bias;
    p.x = a.x;
    p.norm_w = a.norm_weight;
    p.residual = a.residual;
    p.out = a.out;
    p.wait_ctr = a.wait_counters;
    p.signal_ctr = a.signal_counters;
    p.status = a.status;
    p.wait_total = a.wait_total;
    p.eps = a.eps;
    p.K = a.K;
    p.N = a.N;
    p.G = a.K / a.group_size;
    p.chunks = pl.chunks;
    p.tiles = pl.tiles;
    p.cpw = pl.cpw;
    int sh = 0;
    while ((kChunkK << sh) < a.group_size) ++sh;
    p.cpg_shift = sh;
    p.in_glue = a.in_glue;
#define GPTQHIP_DISPATCH(B, A_, S_)                                                        \
    return p.wait_ctr != nullptr ? launch_gemv1_depth<B, A_, S_, 1>(p, pl, stream) \
                                 : launch_gemv1_depth<B, A_, S_, 0>(p, pl, stream)
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
