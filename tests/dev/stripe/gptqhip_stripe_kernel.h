// The "stripe" kernel: a stream-K decomposition with the split-K reduction INSIDE the launch, for batches of roughly 33..512 rows.
// OPT-IN (GPTQHIP_FORCE_KERNEL=3 / gptqhip_set_tuning(.., 3, ..)): parity-tested like the other kernels, but gptqhip_gemm does not
// route to it -- measured on MI355X it lands within +-10 % of the prefill kernel's 64-row tiles + reduce launch on every shape of the
// reference's TFLOPS benchmark (scripts/benchmark_marlin_a100.py:35-44: M = 64..192 on 4096x11008 / 11008x4096 / 4096x4096) and
// ahead of it on none by more than 3 % (profiles/r04_stripe_midm.txt).  DESIGN.md 4.4 has the per-phase timeline and why: the regime
// is bound by the L2 -> LDS path and by dependent round trips, not by the dequant VALU this decomposition saves.
//
// Same contraction and rounding chain as the other two kernels (reference: TorchLinear._forward_eager,
// gptqmodel/nn_modules/qlinear/torch.py:326-347, dequant :700-717).
//
// Decomposition:
//   * a block (8 waves) owns ALL rows of a row panel (up to MT*16, MT <= 16) of a column STRIPE (BN = 64 or 128 columns) over a
//     contiguous K range: every packed word is fetched and dequantised ONCE per panel (in registers, straight into MFMA B fragments,
//     like the prefill kernel); the activation tile is the shared operand (LDS-DMA in full 256-byte rows, XOR-swizzled).
//       KG = 2: waves = 4 column tiles x 2 K-groups; a pipeline stage is TWO 128-row chunks (one per K-group), BN = 64
//       KG = 1: waves = 8 column tiles; a stage is one chunk, BN = 128
//   * work = the linearised (stripe, step) space of an XCD's stripes, cut into equal contiguous ITEMS (stream-K): an item may end in
//     the middle of a stripe and continue at the start of the next one, so every block gets the same number of steps whatever N, K.
//   * stripes are dealt to the 8 XCDs in contiguous runs; each XCD's items form a QUEUE = a 32-bit claim mask.  A block reads its
//     XCC id from the hardware register and claims items of the XCD it actually runs on (one atomic OR, speculatively: the first
//     item's loads start before the claim returns).  All contributors of a stripe therefore share one L2 BY CONSTRUCTION -- not by
//     assuming block b runs on XCD b % 8 -- whatever the dispatcher does; a queue nobody served is drained by the last block to
//     leave the launch, whole stripes at a time.
//   * split-K reduction inside the launch: contributors publish their fp32 fragment slabs (fragment-major, 1 KiB per wave store),
//     `s_waitcnt vmcnt(0)`, ticket; the last arriver sums all slabs in ITEM order (deterministic, independent of the arrival order)
//     with L1-bypassing loads and runs the reference's rounding epilogue.  No slab leaves the XCD's L2 on its way to the reducer
//     and there is no second launch.  (`write_through` = 1 publishes with sc1 stores instead: the placement-independent form of
//     cdna_hip_programming.md Guideline 16; bit-identical results, same speed -- kept as a switch.)
#pragma once
#include "gptqhip_tiled_kernel.h"

namespace gptqhip {

constexpr int kStripeQueues = 8;        // XCDs of an MI355X
constexpr int kStripeGrid = 256;        // one block per CU
constexpr int kStripeItemsPerQueue = 32;   // a queue is a 32-bit claim mask
constexpr int kStripeHeadStride = 64;   // ints between queue heads: 256 B, so the 8 XCDs' dequeues do not serialise on one line / channel
constexpr int kStripeHeadInts = (kStripeQueues + 1) * kStripeHeadStride;

struct StripeParams {
    TiledParams t;      // x, qw, meta, bias, out, M, K, N, G, group_size, ldo, chunks, tiles, cpg_shift
    int vstripes;       // column stripes x row panels
    int panels;         // row panels of MT*16 rows (M > MT*16)
    int sps;            // pipeline steps per stripe = chunks / KG
    int nq, items;      // queues, items per queue
    int write_through;  // publish slabs with sc1 stores
    float* slabs;       // [nq][items][2][MT*16*BN] fp32 fragment slabs
    int* heads;         // [nq + 1][kStripeHeadStride]: queue heads, then the exit counter   (all zero between launches)
    int* tickets;       // [vstripes] arrival counters                        (all zero between launches)
    unsigned long long* stamps;   // dev builds (-DGPTQHIP_STRIPE_STAMPS): [grid][16] phase stamps of each block's FIRST segment
};

#ifdef GPTQHIP_STRIPE_STAMPS
#define STRIPE_STAMP(i)                                                                                          \
    do {                                                                                                          \
        if (p.stamps && tid == 0 && !stamped) p.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define STRIPE_STAMP(i) \
    do {                \
    } while (0)
#endif

// ---- the partition arithmetic, shared by the kernel, the planner and the CPU self-check ------------------------------
struct StripeGeom {
    int vstripes, sps, nq, items;
};
// (32-bit arithmetic on purpose -- a 64-bit division is a ~300-instruction routine on the GPU; the planner guarantees
// items * (steps of a queue + 1) < 2^31 and nq * vstripes < 2^31)
__host__ __device__ inline void stripe_queue_range(const StripeGeom& g, int q, int& v0, int& nvs) {
    v0 = (int)((uint32_t)q * (uint32_t)g.vstripes / (uint32_t)g.nq);
    nvs = (int)((uint32_t)(q + 1) * (uint32_t)g.vstripes / (uint32_t)g.nq) - v0;
}
// item i of a queue with U steps covers [item_start(i), item_start(i + 1))
__host__ __device__ inline int stripe_item_start(int i, int U, int P) { return (int)((uint32_t)i * (uint32_t)U / (uint32_t)P); }
// the item that holds step x (0 <= x < U): the largest i with item_start(i) <= x
__host__ __device__ inline int stripe_item_of(int x, int U, int P) {
    return (int)((((uint32_t)x + 1u) * (uint32_t)P + (uint32_t)U - 1u) / (uint32_t)U) - 1;
}

// items of a queue with U steps: the planned count, but never more than there are steps (no empty items: the contributors of a
// stripe are then exactly the items stripe_item_of(first step) .. stripe_item_of(last step))
// (at most 32: a queue is a 32-bit claim mask)
__host__ __device__ inline int stripe_items_of_queue(int items, int U) { return U < items ? U : items; }

__device__ __forceinline__ int stripe_xcc_id() {
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // hwreg(HW_REG_XCC_ID, 0, 4)
}

template <int ACT, int SCL, int GPC, int MT, int KG, int D>
__global__ __launch_bounds__(512) void stripe_kernel(StripeParams p) {
    constexpr int CT = 8 / KG;            // column tiles (= waves) per K-group
    constexpr int BN = CT * kTileN;       // columns per stripe
    constexpr int MTW = MT / KG;          // row tiles a wave owns after the in-block K-group exchange
    constexpr int SUB = MT * 16 * 256;    // bytes of one K-group's activation sub-tile (MT*16 rows x 128 columns of K)
    constexpr int STAGE = KG * SUB;
    constexpr int NPIECE = MT * 4 / CT;   // 1 KiB LDS-DMA pieces per wave and stage
    constexpr int OPS = NPIECE + 1 + GPC; // VMEM instructions per wave and stage
    constexpr int PF = MT % 4 == 0 ? 4 : (MT % 3 == 0 ? 3 : 2);   // A fragments per read group
    constexpr int NG = 4 * MT / PF;       // read groups per stage
    constexpr int PPG = 2;                // DMA pieces issued per group (all of them before the last K-step's wait)
    constexpr int kSlabF4 = 8 * MTW * 64; // float4 per slab
    constexpr int kOutPitch = BN * 2 + 16;
    constexpr int kExch = KG == 2 ? 8 * MTW * 1024 : 0;   // bytes of the K-group exchange area
    static_assert(MT % KG == 0 && (MT * 4) % CT == 0, "row tiles must split over the K-groups / DMA pieces over the waves");
    static_assert((NPIECE + PPG - 1) / PPG <= 3 * NG / 4 + 1, "every DMA piece is issued before the stage's vmcnt wait");
    static_assert(kExch + MT * 16 * kOutPitch <= D * STAGE, "exchange + output staging reuse the stage buffers");
    // ONE shared object (a second one makes hipcc drain vmcnt before every ds_read of an LDS-DMA pipeline, cdna_hip_programming.md)
    __shared__ __attribute__((aligned(16))) char lds[D * STAGE + 64];
    int* const bcast = reinterpret_cast<int*>(lds + D * STAGE);

    const TiledParams& tp = p.t;
    const int tid = threadIdx.x;
#ifdef GPTQHIP_STRIPE_STAMPS
    bool stamped = false;
    if (p.stamps && tid == 0) {
        p.stamps[(size_t)blockIdx.x * 16 + 0] = __builtin_amdgcn_s_memrealtime();
        p.stamps[(size_t)blockIdx.x * 16 + 15] = (unsigned long long)stripe_xcc_id();
    }
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = KG == 1 ? 0 : wave / CT;
    const int ct = KG == 1 ? wave : wave % CT;
    const int c = lane & 15;
    const int rq = lane >> 4;

    f4_t acc[MT];
    const DequantConsts dk = make_dequant_consts<4>();
    BStage<4, GPC, 1> bst[D];
    const size_t qw_bytes = (size_t)tp.tiles * tp.chunks * 1024, meta_bytes = (size_t)tp.tiles * tp.G * 64;
    const BSrc bsrc = make_b_src(tp, lane, qw_bytes, meta_bytes);
    const uint32_t lds_row_base = (uint32_t)(uintptr_t)lds + (uint32_t)(kg * SUB + c * 256);

    // ---- one pipeline stage's loads: this wave's share of its K-group's activation sub-tile (LDS-DMA) + its weight block.
    // Every stage issues exactly OPS instructions; a stage past the segment's end goes through zero-length descriptors
    // (hardware returns zeros, no memory traffic), so all vmcnt waits are the same compile-time count on every path.
    struct Seg {
        int tile;      // this wave's 16-column weight tile (clamped for a ragged last stripe)
        int u0, n;     // the segment: steps [u0, u0 + n) of the stripe
        const char* abase;
        int arows;     // valid rows of the panel
    };
    // logical step i (0 .. n-1) -> the chunk this wave's K-group works on
    // (walking the steps of a segment in a per-stripe rotated order -- so that blocks sharing a K range read different column slices
    // at any moment -- was measured: no gain at K = 4096, 25 % slower at K = 11008; blocks in lockstep share their L2 misses.)
    auto chunk_of = [&](const Seg& g, int i, bool valid) __attribute__((always_inline)) { return (g.u0 + (valid ? i : 0)) * KG + kg; };
    // (Also measured and dropped: warming the L2 ahead of the LDS-DMA with one dword load per 128-byte line, requested in bulk for the
    // next 8 steps.  A stage holds 64 KiB of the 160 KiB of LDS, so only one stage of activations is in flight while another is
    // multiplied and every stage waits a miss latency; but a 64-lane dword load costs the texture addresser as much as a 1 KiB DMA
    // piece, and the extra 9 instructions per stage -- even through a zero-length descriptor -- took the step from 1.3 to 2.3 us.)
    auto issue = [&](auto sc, const Seg& g, int step) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const bool valid = step < g.n;
        const int chunk = chunk_of(g, step, valid);
        const __amdgpu_buffer_rsrc_t ars =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g.abase), 0, valid ? g.arows * tp.K * 2 : 0, 0x00020000);
        const int rl = ct * 4 + rq;
        const uint32_t voff = (uint32_t)(rl * tp.K * 2 + ((c ^ (rl & 15)) << 4));
        typedef __attribute__((address_space(3))) void* lptr_t;
        static_for<NPIECE>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            char* dst = lds + s * STAGE + kg * SUB + (I * CT + ct) * 4 * 256;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lptr_t)dst, 16, voff, chunk * (kChunkK * 2) + I * (CT * 4 * tp.K * 2), 0, 0);
        });
        const __amdgpu_buffer_rsrc_t brs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(tp.qw), 0, valid ? (int)qw_bytes : 0, 0x00020000);
        bst[s].w[0][0] = __builtin_amdgcn_raw_buffer_load_b128(brs, bsrc.l16, (uint32_t)(g.tile * tp.chunks + chunk) * 1024u, 0);
#pragma unroll
        for (int j = 0; j < GPC; ++j)
            bst[s].meta[0][j] = __builtin_amdgcn_raw_buffer_load_b32(
                bsrc.meta, bsrc.c4, (uint32_t)(g.tile * tp.G + tiled_group_of(tp, chunk * kChunkK + j * (kChunkK / GPC))) * 64u, 0);
    };
    // the same, spread over a stage: weights right after the first fragment reads, DMA pieces PPG per read group
    auto issue_b = [&](auto sc, const Seg& g, int step) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const bool valid = step < g.n;
        const int chunk = chunk_of(g, step, valid);
        const __amdgpu_buffer_rsrc_t brs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(tp.qw), 0, valid ? (int)qw_bytes : 0, 0x00020000);
        bst[s].w[0][0] = __builtin_amdgcn_raw_buffer_load_b128(brs, bsrc.l16, (uint32_t)(g.tile * tp.chunks + chunk) * 1024u, 0);
#pragma unroll
        for (int j = 0; j < GPC; ++j)
            bst[s].meta[0][j] = __builtin_amdgcn_raw_buffer_load_b32(
                bsrc.meta, bsrc.c4, (uint32_t)(g.tile * tp.G + tiled_group_of(tp, chunk * kChunkK + j * (kChunkK / GPC))) * 64u, 0);
    };

    u4_t bnow, bnext;
    // (the group constants are expanded once per chunk, at its K-step 0, like in the prefill kernel)
    ColConst ccs;
    auto dequant_step = [&](const BStage<4, GPC, 1>& bs, int j, u4_t& b) __attribute__((always_inline)) {
        if (GPC == 4 || j == 0) ccs = expand_meta<4, SCL>(bs.meta[0][GPC == 4 ? j : 0]);
        b = dequant_word4<ACT, SCL>(bs.w[0][0][j], ccs, dk);
    };

    // One stage: barrier (the stage's sub-tiles have landed -- every wave waited for its own pieces -- and everybody is done
    // with the buffer this stage's DMA overwrites), issue step + D - 1, multiply the stage.  Same choreography as the prefill
    // kernel's stage (gptqhip_tiled_kernel.h): fragment reads one group ahead of the MFMAs (inline asm + counted lgkmcnt),
    // the next K-step's dequant under the MFMAs, the wait for the next stage's loads at the start of the last K-step.
    auto stage = [&](auto sc, auto first_c, const Seg& g, int step) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        constexpr bool kFirst = decltype(first_c)::value;
        constexpr int sn = (s + 1) % D, si = (s + D - 1) % D;
        u4_t abuf[2][PF];
        __builtin_amdgcn_s_barrier();
        const uint32_t abase = lds_row_base + (uint32_t)(s * STAGE);
        uint32_t aaddr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) aaddr[j] = abase + (uint32_t)((j * 64 + rq * 16) ^ (c << 4));
        static_for<PF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            lds_read_b128<(i % MT) * 4096>(abuf[0][i], aaddr[i / MT]);
        });
        const int nstep = step + D - 1;
        const bool nvalid = nstep < g.n;
        const int nchunk = chunk_of(g, nstep, nvalid);
        const __amdgpu_buffer_rsrc_t ars =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g.abase), 0, nvalid ? g.arows * tp.K * 2 : 0, 0x00020000);
        const int rl = ct * 4 + rq;
        const uint32_t voff = (uint32_t)(rl * tp.K * 2 + ((c ^ (rl & 15)) << 4));
        // (the first stage of a segment issues nothing: the prologue has put D stages in flight, one per buffer)
        if constexpr (!kFirst) issue_b(std::integral_constant<int, si>{}, g, nstep);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NG>([&](auto gc) {
            constexpr int gi = decltype(gc)::value;
            if constexpr (gi + 1 < NG) {
                static_for<PF>([&](auto ic) {
                    constexpr int idx = (gi + 1) * PF + decltype(ic)::value;
                    lds_read_b128<(idx % MT) * 4096>(abuf[(gi + 1) & 1][decltype(ic)::value], aaddr[idx / MT]);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<PPG>([&](auto pc) {
                constexpr int I = gi * PPG + decltype(pc)::value;
                if constexpr (I < NPIECE && !kFirst) {
                    typedef __attribute__((address_space(3))) void* lptr_t;
                    char* dst = lds + si * STAGE + kg * SUB + (I * CT + ct) * 4 * 256;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lptr_t)dst, 16, voff,
                                                             nchunk * (kChunkK * 2) + I * (CT * 4 * tp.K * 2), 0, 0);
                }
            });
            constexpr int j = (gi * PF) / MT;
            constexpr bool last_of_step = ((gi + 1) * PF) % MT == 0;
            if constexpr ((gi * PF) % MT == 0) {
                if constexpr (j < 3) {
                    dequant_step(bst[s], j + 1, bnext);
                } else {
                    vm_wait<OPS, 0, D - 2>(D - 2, false);
                    dequant_step(bst[sn], 0, bnext);
                }
            }
            lds_wait<(gi + 1 < NG) ? PF : 0, PF>(abuf[gi & 1]);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int mt = (gi * PF + i) % MT;
                if constexpr (kFirst && j == 0) {
                    acc[mt] = mfma16<ACT>(abuf[gi & 1][i], bnow, f4_t{0.f, 0.f, 0.f, 0.f});
                } else {
                    acc[mt] = mfma16<ACT>(abuf[gi & 1][i], bnow, acc[mt]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (last_of_step) bnow = bnext;
        });
    };

    // ---- the accumulators of one segment: steps [u0, u1) of one virtual stripe
    auto run_segment = [&](const Seg& g) __attribute__((always_inline)) {
        const int nsteps = g.n;
        constexpr int u0 = 0;   // (logical step indices from here on)
        // Prologue.  A segment of at least D steps starts with ALL D buffers in flight (its first stage then issues nothing): at
        // 64 KiB per stage only two stages fit the LDS, every stage of a short segment waits a full cold-miss latency (1.3 us against
        // 0.45 us of MFMA work, profiles/r04_stripe_stamps.txt), and this takes one of those latencies off each segment.
        int i = 0;
        if (nsteps >= D) {
            static_for<D>([&](auto dc) { issue(dc, g, u0 + decltype(dc)::value); });
            vm_wait<OPS, 0, D - 1>(D - 1, false);
        } else {
            static_for<D - 1>([&](auto dc) { issue(dc, g, u0 + decltype(dc)::value); });
            vm_wait<OPS, 0, D - 2>(D - 2, false);
        }
        STRIPE_STAMP(2);   // first stage's loads have landed
        dequant_step(bst[0], 0, bnow);
        if (nsteps >= D) {
            static_for<D>([&](auto sc) {
                stage(sc, std::integral_constant<bool, decltype(sc)::value == 0>{}, g, u0 + decltype(sc)::value);
            });
            i = D;
            while (i + D <= nsteps) {
                static_for<D>([&](auto sc) { stage(sc, std::false_type{}, g, u0 + i + decltype(sc)::value); });
                i += D;
            }
        } else {
            f4_t zero = {0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+v"(zero));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = zero;
        }
        static_for<D - 1>([&](auto sc) {
            if (i + decltype(sc)::value < nsteps) stage(sc, std::false_type{}, g, u0 + i + decltype(sc)::value);
        });
        // the trailing (zero-length) DMA writes must have landed before the stage buffers are reused below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        STRIPE_STAMP(3);   // main loop done
    };

    // ---- in-block K-group exchange (KG = 2): wave (kg, ct) keeps row tiles [kg * MTW, (kg + 1) * MTW) and adds its partner's
    auto exchange = [&](f4_t (&mine)[MTW]) __attribute__((always_inline)) {
        if constexpr (KG == 1) {
#pragma unroll
            for (int m = 0; m < MTW; ++m) mine[m] = acc[m];
        } else {
            f4_t* ex = reinterpret_cast<f4_t*>(lds);
            const int wd = (1 - kg) * CT + ct;   // destination wave
            if (kg == 0) {
#pragma unroll
                for (int m = 0; m < MTW; ++m) ex[(wd * MTW + m) * 64 + lane] = acc[MTW + m];
            } else {
#pragma unroll
                for (int m = 0; m < MTW; ++m) ex[(wd * MTW + m) * 64 + lane] = acc[m];
            }
            __syncthreads();
            if (kg == 0) {
#pragma unroll
                for (int m = 0; m < MTW; ++m) mine[m] = acc[m] + ex[(wave * MTW + m) * 64 + lane];
            } else {
#pragma unroll
                for (int m = 0; m < MTW; ++m) mine[m] = ex[(wave * MTW + m) * 64 + lane] + acc[MTW + m];
            }
        }
    };

    // ---- reference rounding chain + row-major 16-byte stores (transposed through LDS: BN columns of a row are contiguous)
    auto epilogue = [&](const f4_t (&v)[MTW], int stripe, int m0) __attribute__((always_inline)) {
        char* ob = lds + kExch;
        const int n = stripe * BN + ct * kTileN + c;
        const float bias = (tp.bias != nullptr && n < tp.N) ? load16_as_f32<ACT>(tp.bias, (size_t)n) : 0.f;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int row0 = (kg * MTW + m) * 16 + 4 * rq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float y = round_through<ACT>(v[m][i]);
                if (tp.bias != nullptr) y = y + bias;
                *reinterpret_cast<uint16_t*>(ob + (row0 + i) * kOutPitch + (ct * kTileN + c) * 2) = f32_to_16<ACT>(y);
            }
        }
        __syncthreads();
        constexpr int PR = BN / 8;          // 16-byte pieces per row
        constexpr int RP = 512 / PR;        // rows per pass
#pragma unroll
        for (int pass = 0; pass < (MT * 16 + RP - 1) / RP; ++pass) {
            const int row = pass * RP + tid / PR, piece = tid % PR;
            const int gr = m0 + row, gn = stripe * BN + piece * 8;
            if (row < MT * 16 && gr < tp.M && gn < tp.N) {
                const u4_t val = *reinterpret_cast<const u4_t*>(ob + row * kOutPitch + piece * 16);
                *reinterpret_cast<u4_t*>(reinterpret_cast<char*>(tp.out) + ((size_t)gr * tp.ldo + gn) * 2) = val;
            }
        }
        __syncthreads();
    };

    const __amdgpu_buffer_rsrc_t slab_rs =
        __builtin_amdgcn_make_buffer_rsrc(p.slabs, 0, (int)((size_t)p.nq * p.items * 2 * kSlabF4 * 16), 0x00020000);
    const uint32_t slab_lane = (uint32_t)((wave * MTW * 64 + lane) * 16);

    int pre = 0;             // thread 0: value returned by the claim atomic in flight (see the work loop)
    bool spec = false;       // the current item was started BEFORE its claim was confirmed
    // ---- one segment from start to finish.  my_item < 0: the caller owns the whole stripe (no cross-block sum).
    auto do_segment = [&](int q, int v, int ub, int ue, int my_item, int i_first, int i_last, int U, int P, int vbase_steps, bool first_seg,
                          bool last_seg, int pre_q) __attribute__((always_inline)) -> bool {
        const int stripe = v / p.panels, panel = v - stripe * p.panels;
        const int m0 = panel * (MT * 16);
        Seg g;
        int tile = stripe * CT + ct;
        g.tile = tile < tp.tiles ? tile : tp.tiles - 1;
        g.u0 = ub;
        g.n = ue - ub;
        g.abase = reinterpret_cast<const char*>(tp.x) + (size_t)m0 * tp.K * 2;
        g.arows = min(tp.M - m0, MT * 16);
        run_segment(g);
        if (spec) {
            // the claim of this item was issued before its loads and has long returned: nothing has left the block yet, so a lost
            // claim (another block of this XCD holds the item: only possible when the dispatcher did not place block b on XCD
            // b % 8) simply drops the accumulators
            if (tid == 0) bcast[2] = (int)(((uint32_t)pre >> my_item) & 1u);
            __syncthreads();
            const bool lost = bcast[2] != 0;
            __syncthreads();
            spec = false;
            if (lost) return false;
        }
        // a fresh view of the claim mask for the next item, requested here so that its round trip runs under the exchange /
        // publish / reduce / epilogue below instead of after them
        if (last_seg && my_item >= 0 && tid == 0)
            pre = __hip_atomic_fetch_or(&p.heads[pre_q * kStripeHeadStride], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f4_t mine[MTW];
        exchange(mine);
        STRIPE_STAMP(4);   // K-group exchange done
        const int contrib = my_item < 0 ? 1 : i_last - i_first + 1;
        if (contrib > 1) {
            // publish this block's fragments: [wave][row tile][lane] float4, 1 KiB per wave store
            const uint32_t my_slab = (uint32_t)(((q * p.items + my_item) * 2 + (first_seg ? 0 : 1)) * (kSlabF4 * 16));
            if (p.write_through) {
#pragma unroll
                for (int m = 0; m < MTW; ++m)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, mine[m]), slab_rs, slab_lane + m * 1024, my_slab, 16);
            } else {
#pragma unroll
                for (int m = 0; m < MTW; ++m)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, mine[m]), slab_rs, slab_lane + m * 1024, my_slab, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slab stores are in the L2 (or in memory: sc1)
            __syncthreads();
            STRIPE_STAMP(5);   // slab published
            if (tid == 0) bcast[0] = __hip_atomic_fetch_add(&p.tickets[v], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int ticket = bcast[0];
            STRIPE_STAMP(6);   // ticket drawn
            if (ticket != contrib - 1) {
                __syncthreads();   // (bcast is reused by the next claim)
                return true;
            }
            if (tid == 0) __hip_atomic_store(&p.tickets[v], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // every contributor's ticket precedes ours, its slab stores precede its ticket: sum the slabs in ITEM order (own slab
            // read back like the others, so the order -- hence the result -- does not depend on who arrived last).  sc1 loads
            // bypass this CU's L1 and are served by the L2 all contributors share.
            const int stripe_first_step = vbase_steps;   // first step of this stripe in the queue's step space
            auto slab_of = [&](int ii) {
                const int started_here = stripe_item_start(ii, U, P) >= stripe_first_step;
                return (uint32_t)(((q * p.items + ii) * 2 + (started_here ? 0 : 1)) * (kSlabF4 * 16));
            };
            const int slab_bytes_all = (int)((size_t)p.nq * p.items * 2 * kSlabF4 * 16);
            constexpr int CMAX = 4, MG = MTW < 4 ? MTW : 4;   // up to 4 contributors x 4 row tiles in flight per wave (64 registers)
            if (contrib <= CMAX) {
                // every slab load is requested before the first addition (one loaded L2 round trip instead of one per contributor);
                // the slots beyond the last contributor go through a zero-length descriptor and are not added
#pragma unroll
                for (int mg = 0; mg < MTW; mg += MG) {
                    f4_t buf[CMAX][MG];
#pragma unroll
                    for (int k = 0; k < CMAX; ++k) {
                        const bool on = k < contrib;
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slabs, 0, on ? slab_bytes_all : 0, 0x00020000);
                        const uint32_t so = slab_of(on ? i_first + k : i_first);
#pragma unroll
                        for (int m = 0; m < MG; ++m)
                            buf[k][m] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, slab_lane + (mg + m) * 1024, so, 16));
                    }
#pragma unroll
                    for (int m = 0; m < MG; ++m) {
                        f4_t sum = buf[0][m];
#pragma unroll
                        for (int k = 1; k < CMAX; ++k)
                            if (k < contrib) sum = sum + buf[k][m];
                        mine[mg + m] = sum;
                    }
                }
            } else {
                f4_t cur[MTW], nxt[MTW];
#pragma unroll
                for (int m = 0; m < MTW; ++m)
                    cur[m] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(slab_rs, slab_lane + m * 1024, slab_of(i_first), 16));
                for (int ii = i_first; ii <= i_last; ++ii) {
                    const bool more = ii < i_last;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slabs, 0, more ? slab_bytes_all : 0, 0x00020000);
                    const uint32_t so = slab_of(more ? ii + 1 : ii);
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
                        nxt[m] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, slab_lane + m * 1024, so, 16));
                    if (ii == i_first) {
#pragma unroll
                        for (int m = 0; m < MTW; ++m) mine[m] = cur[m];
                    } else {
#pragma unroll
                        for (int m = 0; m < MTW; ++m) mine[m] = mine[m] + cur[m];
                    }
#pragma unroll
                    for (int m = 0; m < MTW; ++m) cur[m] = nxt[m];
                }
            }
            STRIPE_STAMP(7);   // slabs summed (last arriver only)
        }
        epilogue(mine, stripe, m0);
        STRIPE_STAMP(8);       // output stored
        return true;
    };

    // ---- work loop.
    // Queue of XCD q = a 32-bit CLAIM MASK (bit i: item i is taken).  A block claims an item with one returning atomic OR; it owns
    // the item iff the bit was clear before.  All items of queue q are therefore processed by blocks that read XCC id q from the
    // hardware -- the contributors of a stripe share one L2 by construction, wherever the dispatcher put them.
    //   * first item: block b guesses item b / nq (every block a different one when block b runs on XCD b % 8, as observed) and
    //     starts its loads IMMEDIATELY; the claim's round trip (1.7 us at kernel start, measured) runs under the first miss latency
    //     and the main loop, and is checked before anything leaves the block.  A lost claim only costs the work done so far.
    //   * further items (fewer blocks than items on this XCD, or a lost claim): lowest clear bit of the freshest known mask, retried
    //     until owned or the mask is full.
    //   * exit: the LAST block to leave the launch clears the masks; a mask that is still zero belongs to an XCD no block of this
    //     launch ran on: mode 1 drains that queue here, whole stripes at a time, no cross-block sums.
    // (One call site of do_segment on purpose: the pipeline is a few thousand instructions.)
    const StripeGeom geom = {p.vstripes, p.sps, p.nq, p.items};
    int q = stripe_xcc_id() % p.nq;
    int v0, nvs;
    stripe_queue_range(geom, q, v0, nvs);
    int U = nvs * p.sps;
    const int P = stripe_items_of_queue(p.items, U);   // every item holds at least one step; P <= 32
    int mode = 0, item = -1, s0 = 0, s = 0, s1 = 0, ot = 0;
    uint32_t known = 0;   // thread 0: the claim mask as far as this block has seen it
    if (P > 0) {
        item = (int)((blockIdx.x / (unsigned)p.nq) % (unsigned)P);
        if (tid == 0) pre = __hip_atomic_fetch_or(&p.heads[q * kStripeHeadStride], 1 << item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spec = true;
    }
    bool have_item = P > 0;
    for (;;) {
        int v, ub, ue, my_item, i_first = 0, i_last = 0, vbase;
        bool first_seg = true;
        if (mode == 0) {
            if (s >= s1) {
                if (!have_item) {
                    // claim the lowest clear bit of the freshest mask (thread 0; `pre` holds the mask read after the last main loop,
                    // or the mask returned by a lost speculative claim)
                    if (tid == 0) {
                        int got = -1;
                        known |= (uint32_t)pre;
                        const uint32_t full = P >= 32 ? 0xFFFFFFFFu : (1u << P) - 1u;
                        while ((known & full) != full) {
                            const int cand = __builtin_ctz(~known & full);
                            const uint32_t old = __hip_atomic_fetch_or(reinterpret_cast<uint32_t*>(&p.heads[q * kStripeHeadStride]), 1u << cand,
                                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            known |= old | (1u << cand);
                            if (!((old >> cand) & 1u)) {
                                got = cand;
                                break;
                            }
                        }
                        bcast[0] = got;
                    }
                    __syncthreads();
                    item = bcast[0];
                    __syncthreads();
                }
#ifdef GPTQHIP_STRIPE_STAMPS
                if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + (stamped ? 9 : 1)] = __builtin_amdgcn_s_memrealtime();
#endif
                if (item < 0) {
                    if (tid == 0)
                        bcast[0] = __hip_atomic_fetch_add(&p.heads[p.nq * kStripeHeadStride], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    const bool last_out = bcast[0] == (int)gridDim.x - 1;
                    __syncthreads();
#ifdef GPTQHIP_STRIPE_STAMPS
                    if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
#endif
                    if (!last_out) return;
                    if (tid <= p.nq) bcast[1 + tid] = __hip_atomic_exchange(&p.heads[tid * kStripeHeadStride], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    mode = 1;
                    q = -1;
                    nvs = 0;
                    ot = 0;
                    continue;
                }
                have_item = false;
                s0 = stripe_item_start(item, U, P);
                s1 = stripe_item_start(item + 1, U, P);
                s = s0;
                if (s >= s1) continue;
            }
            const int t = s / p.sps;
            ub = s - t * p.sps;
            ue = min(p.sps, ub + (s1 - s));
            i_first = stripe_item_of(t * p.sps, U, P);
            i_last = stripe_item_of((t + 1) * p.sps - 1, U, P);
            v = v0 + t;
            vbase = t * p.sps;
            my_item = item;
            first_seg = s == s0;
            s += ue - ub;
        } else {
            while (ot >= nvs) {   // next unserved queue
                ++q;
                if (q >= p.nq) return;
                ot = 0;
                nvs = 0;
                if (bcast[1 + q] == 0) stripe_queue_range(geom, q, v0, nvs);
            }
            v = v0 + ot;
            ub = 0;
            ue = p.sps;
            vbase = ot * p.sps;
            my_item = -1;
            U = nvs * p.sps;
            ++ot;
        }
        const bool ok = do_segment(q, v, ub, ue, my_item, i_first, i_last, U, P, vbase, first_seg, mode == 0 && s >= s1, q);
        if (!ok) s = s1;   // lost speculative claim: forget the item
#ifdef GPTQHIP_STRIPE_STAMPS
        stamped = true;
#endif
    }
}

}  // namespace gptqhip
