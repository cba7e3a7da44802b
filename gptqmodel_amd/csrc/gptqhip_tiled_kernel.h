// Prefill / large-batch fused dequant-GEMM (M > 32): MFMA-bound.  Kernel template + per-bit-width launcher; instantiated
// by gptqhip_tiled.hip (4-bit) and gptqhip_tiled8.hip (8-bit) so the two halves compile in parallel.
//
// Replaces, for large M, the reference's "dequantise the whole [K,N] weight to fp16, then aten matmul"
// (gptqmodel/nn_modules/qlinear/torch.py:326-347) and plays the role Marlin / ExllamaV2's reconstruct+GEMM play
// on NVIDIA (gptqmodel_ext/marlin/gptq_marlin.cu, gptqmodel_ext/exllamav2/cuda/q_gemm.cu:118-137) -- designed
// for CDNA4 instead of translated:
//
//   block = 8 waves (2 per SIMD), output tile BM x 256 (BM = 256, 128, 64, or -- round 5, 4-bit / one group constant per chunk -- any
//   multiple of 16 up to 128, so that a batch of 72 or 136 rows is not rounded up to 128 / 192), K advanced one 128-row chunk at a
//   time; one block per CU loops over the output tiles (persistent).
//   * B (weights) never touches LDS: wave w owns column tiles 2w, 2w+1 of the block (32 columns) for ALL BM rows,
//     so every packed word is fetched (one dwordx4 per lane per tile-chunk, 1 KiB contiguous) and dequantised
//     exactly once per block, in registers, straight into mfma_f32_16x16x32 B fragments.
//   * A (activations) is the shared operand: the BM x 128 tile goes HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds) in
//     full 256-byte rows, XOR-swizzled on the source side ((row&15)<<4) so the column-slice ds_read_b128 of the A
//     fragments is bank-conflict free (cdna_hip_programming.md T2 / rule 21).
//   * D-stage chunk pipeline (D = 2 on 256-row tiles, 3 on 128-row tiles): one s_barrier per chunk, counted vmcnt
//     waits, the next chunk's first K-step dequantised before the barrier, DMA pieces issued between MFMA groups.
//   * per K-step (32 rows) a wave issues 2 dequants (~26 VALU) + BM/16 ds_read_b128 + 2*BM/16 MFMAs.
//   * epilogue: round like the reference (round(acc), + bias, round) or keep fp32 (split-K slabs, tensor-parallel
//     partial sums), transposed through LDS into 16-byte buffer stores; it overlaps the next tile's first loads.
// docs/history/DESIGN_rounds_1-5.md section 4.2 has the measurements behind each of these choices.
#pragma once
#include "gptqhip_device.h"
#include "gptqhip_host.h"

#include <utility>

// Cache policy of the fp32 (split-K slab / TP partial) stores: sc1 = write-through.  The slabs are read by the reduce kernel on other XCDs
// right after the launch; written through, they do not wait in the writers' L2 for the end-of-kernel write-back (A/B round 5, same box, graph
// replay: 11008x4096 at M = 128 with 8 slabs 23.1 -> 21.9 us, 14336x4096 26.7 -> 25.4, 4096^2 with 3 slabs 13.9 -> 13.7; `nt` instead:
// no gain on the large slabs).  0 = plain, 2 = nt (dev A/B builds).
#ifndef GPTQHIP_SLAB_AUX
#define GPTQHIP_SLAB_AUX 16
#endif

#ifndef GPTQHIP_TILED_D64   // pipeline stages of the 64-row-tile instantiations (dev A/B builds override it)
#define GPTQHIP_TILED_D64 2
#endif

namespace gptqhip {

constexpr int kTiledD64 = GPTQHIP_TILED_D64;

struct TiledParams {
    const void* x;
    const uint32_t* qw;
    const uint32_t* meta;
    const void* bias;
    void* out;
    int M, K, N, G, group_size;
    int ldo;    // output row stride (elements)
    int chunks;
    int tiles;  // ceil(N/16)
    int out_f32;
    int cpg_shift;
    int splits;            // grid.z: K split across blocks (small grids); partials go to `slabs`
    int chunks_per_split;
    float* slabs;          // [splits][M][N] fp32 when splits > 1
};

__device__ __forceinline__ int tiled_group_of(const TiledParams& p, int k) {
    int g;
    if (p.cpg_shift >= 0) {
        g = k >> (7 + p.cpg_shift);
    } else {
        g = k / p.group_size;
    }
    return g < p.G ? g : p.G - 1;
}

template <int BITS, int GPC, int TPW>
struct BStage {
    u4_t w[TPW][BITS == 4 ? 1 : 2];
    uint32_t meta[TPW][GPC];
};

// B loads go through buffer (MUBUF) instructions: ONE per-lane 32-bit offset register (lane * 16, or (lane & 15) * 4 for the
// group constants) + a wave-uniform scalar offset per (tile, chunk).  With flat/global addressing hipcc hoisted one 64-bit VGPR
// pointer per (tile, tensor) out of the chunk loop and re-added the chunk offset with 64-bit vector adds; on the 256-row tile those
// 8+ registers did not fit beside the 128 accumulators and came back as scratch reloads with a FULL vmcnt drain in the main loop
// (profiles/r03_isa_audit.txt).  The descriptors cover the whole tensors (< 2 GiB each: checked by the launcher).
struct BSrc {
    __amdgpu_buffer_rsrc_t qw, meta;
    uint32_t l16, c4;   // per-lane byte offsets
};
__device__ __forceinline__ BSrc make_b_src(const TiledParams& p, int lane, size_t qw_bytes, size_t meta_bytes) {
    BSrc b;
    b.qw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.qw), 0, (int)qw_bytes, 0x00020000);
    b.meta = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.meta), 0, (int)meta_bytes, 0x00020000);
    b.l16 = (uint32_t)lane * 16u;
    b.c4 = (uint32_t)(lane & 15) * 4u;
    return b;
}

template <int BITS, int GPC, int TPW>
__device__ __forceinline__ void load_b(BStage<BITS, GPC, TPW>& st, const TiledParams& p, const BSrc& bs, int tile0, int chunk) {
    constexpr int WPC = BITS == 4 ? 1 : 2;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        int tile = tile0 + t;
        tile = tile < p.tiles ? tile : p.tiles - 1;  // ragged N: clamp (those columns are never stored)
        const uint32_t soff = (uint32_t)(tile * p.chunks + chunk) * (uint32_t)(WPC * 1024);
#if defined(GPTQHIP_ABLATE_BLOAD)   // dev timing build (tests/dev/tiled_ablate_build.sh; WRONG results): no weight loads at all
#pragma unroll
        for (int h = 0; h < WPC; ++h) st.w[t][h] = u4_t{soff, bs.l16, soff ^ bs.l16, soff + bs.l16};
#else
#pragma unroll
        for (int h = 0; h < WPC; ++h) st.w[t][h] = __builtin_amdgcn_raw_buffer_load_b128(bs.qw, bs.l16, soff + h * 1024, 0);
#endif
        const uint32_t mrow = (uint32_t)(tile * p.G);
#if defined(GPTQHIP_ABLATE_BLOAD) || defined(GPTQHIP_ABLATE_META)   // dev timing build: the group constants without their loads
#pragma unroll
        for (int j = 0; j < GPC; ++j) st.meta[t][j] = 0xE4082000u + mrow;
#else
#pragma unroll
        for (int j = 0; j < GPC; ++j)
            st.meta[t][j] = __builtin_amdgcn_raw_buffer_load_b32(bs.meta, bs.c4,
                                                                 (mrow + (uint32_t)tiled_group_of(p, chunk * kChunkK + j * (kChunkK / GPC))) * 64u, 0);
#endif
    }
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS-DMA staging (buffer_load_dwordx4 ... lds): the A tile goes HBM/L2 -> LDS without passing through VGPRs.  The
// hardware writes wave-uniform base + lane*16, i.e. the LDS image is lane-linear: one instruction fills 4 rows x 256 B.
// The XOR swizzle therefore moves to the SOURCE address (lane (r, pos) fetches segment pos ^ (row & 15)) and the
// fragment reads apply the same XOR (cdna_hip_programming.md rule 21: linear destination + swizzled source + swizzled
// read).  MUBUF rather than global_load_lds on purpose: hipcc treats the FLAT-encoded global_load_lds as touching both
// VMEM and LDS and then turns EVERY later vmcnt/lgkmcnt wait into a full drain, which defeats the multi-stage pipeline.
// The buffer descriptor covers exactly this block's valid rows, so rows >= M and columns >= K come back as zeros
// (no clamps, no address VALU: the lane offset is chunk-invariant, the chunk advances through the scalar offset).
struct ATileSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;        // this lane's byte offset inside a 32-row piece: row * K * 2 + swizzled segment * 16
    uint32_t piece_step;  // bytes between consecutive pieces of one wave (NT/64 * 4 rows)
};

template <int BM, int NT>
__device__ __forceinline__ ATileSrc make_a_src(const TiledParams& p, int m0, int wave, int lane) {
    ATileSrc a;
    const int rows = min(p.M - m0, BM);
    const char* base = reinterpret_cast<const char*>(p.x) + (size_t)m0 * p.K * 2;
    a.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, rows * p.K * 2, 0x00020000);
    const int rl = wave * 4 + (lane >> 4);
    a.voff = (uint32_t)(rl * p.K * 2 + (((lane & 15) ^ (rl & 15)) << 4));
    a.piece_step = (uint32_t)((NT / 64) * 4 * p.K * 2);
    return a;
}

template <int BM, int NT, int I>
__device__ __forceinline__ void stage_a_piece(const ATileSrc& a, char* lds_buf, int chunk, int wave) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int r0 = (I * (NT / 64) + wave) * 4;  // wave-uniform first row of this 1 KiB piece (r0 & 15 == 4*wave & 15)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a.rsrc, (lptr_t)(lds_buf + r0 * 256), 16, a.voff,
                                             chunk * (kChunkK * 2) + I * a.piece_step, 0, 0);
}

// (BMP: the tile height rounded up to whole 32-row DMA pieces; rows past the descriptor come back as zeros and are never read)
template <int BMP, int NT>
__device__ __forceinline__ void stage_a_dma(const ATileSrc& a, char* lds_buf, int chunk, int wave) {
    static_for<BMP * 16 / NT>([&](auto ic) { stage_a_piece<BMP, NT, decltype(ic)::value>(a, lds_buf, chunk, wave); });
}

// ds_read_b128 the compiler does not track: completion is awaited by lds_wait<CNT>, whose "+v" operands make every
// consumer of the fragments depend on the wait.
template <int OFF>
__device__ __forceinline__ void lds_read_b128(u4_t& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT, int N>
__device__ __forceinline__ void lds_wait(u4_t (&frag)[N]) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag[0]) : "n"(CNT));
#pragma unroll
    for (int i = 1; i < N; ++i) asm volatile("" : "+v"(frag[i]));
}

// s_waitcnt vmcnt(stages * OPS + (plus_stores ? S : 0)) for block-uniform run-time arguments, stages in [0, MAXS]
// (the immediate must be a constant: one compare per possible stage count, unrolled at compile time)
template <int OPS, int S, int MAXS>
__device__ __forceinline__ void vm_wait(int stages, bool plus_stores) {
    static_assert(MAXS >= 0 && MAXS <= 4, "at most 6 pipeline stages");
    static_assert(MAXS * OPS + S <= 63, "vmcnt is a 6-bit field");
    if constexpr (MAXS >= 1) {
        if (stages < MAXS) {
            vm_wait<OPS, S, MAXS - 1>(stages, plus_stores);
            return;
        }
    }
    if (plus_stores) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXS * OPS + S) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXS * OPS) : "memory");
    }
}

// Per-tile context of the persistent tile loop
struct TileCtx {
    int m0;     // first row of the tile
    int tile0;  // this wave's first 16-column weight tile
    ATileSrc a;
};

// OUTF = 0: 16-bit output with the reference's rounding chain; 1: fp32 accumulators (split-K slabs, TP partial sums).  A
// template parameter rather than a run-time branch: the two epilogues issue different numbers of stores, and a branch
// between them inside the tile loop makes hipcc assume the smaller count (zero, after its CFG lowering) in every wait.
// BN: columns per block -- 256 (two column tiles per wave), or 128 (ONE column tile per wave; round 5): half the dequant VALU and MFMAs per
// wave and chunk, twice the blocks for the same (M, N), so that wide layers at 64..192 rows fill the chip WITHOUT split-K slabs.
template <int BITS, int ACT, int SCL, int GPC, int BM, int WAVES, int D, int OUTF, int BN = kTiledBN>
__global__ __launch_bounds__(64 * WAVES) void tiled_kernel(TiledParams p) {
    static_assert(BM % 16 == 0 && BM >= 16 && BM <= 256, "row tiles of 16");
    constexpr int MT = BM / 16;
    constexpr int NT = 64 * WAVES;
    constexpr int BMP = (BM + 31) / 32 * 32;  // LDS image / DMA height: whole 32-row pieces (NT / 64 waves x 4 rows each)
    static_assert(NT == 512, "the 32-row DMA piece assumes 8 waves");
    constexpr int TPW = BN / kTileN / WAVES;  // column tiles per wave
    static_assert(TPW == 1 || TPW == 2, "one or two column tiles per wave");
    // epilogue geometry: a wave's output is BM rows x WC columns, staged through LDS and stored in 16-byte pieces
    constexpr int WC = 16 * TPW;
    constexpr int LPR16 = WC / 8, RPP16 = 64 / LPR16;    // 16-bit: lanes per row, rows per store instruction (4, 16 | 2, 32)
    constexpr int LPR32 = WC / 4, RPP32 = 64 / LPR32;    // fp32:                                              (8, 8 | 4, 16)
    // D stage buffers for the A tile + a separate staging area for the epilogue transposes (BM x 16 B per wave), so a
    // tile's output can leave while the NEXT tile's first stages are already landing in the stage buffers
    __shared__ __attribute__((aligned(16))) char lds_all[D * BMP * 256];
    // staging per wave: TG row tiles per epilogue pass (16-bit: 64 B per row; fp32: 128 B per row)
    constexpr int TG16 = TPW == 1 ? 2 : (MT >= 4 ? MT / 4 : 1), TG32 = TPW == 1 ? 1 : (MT >= 8 ? MT / 8 : 1);
    constexpr int kEpi = OUTF ? TG32 * 16 * WC * 4 : TG16 * 16 * WC * 2;  // bytes per wave
    __shared__ __attribute__((aligned(16))) char lds_epi[WAVES * kEpi];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15;
    const int rq = lane >> 4;

    // Persistent tile loop: block b works on virtual blocks b, b + G, b + 2G, ... (G = gridDim.x, one block per CU).
    // XCD-aware order (cdna_hip_programming.md T1): hardware places block b on XCD b % 8 and G % 8 == 0 or G == ntiles,
    // so virtual block v also runs on XCD v % 8; remap so each XCD works on a contiguous run of (bm, bn) pairs -> the
    // blocks sharing one A row-panel reuse it from ONE L2.
    const int nbx = ceil_div(p.N, BN);
    const int ntiles = nbx * ceil_div(p.M, BM);
    const int G = gridDim.x;
    auto make_ctx = [&](int v) __attribute__((always_inline)) {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective for any ntiles
        const int bm = lin / nbx;
        const int bn = lin - bm * nbx;
        TileCtx t;
        t.m0 = bm * BM;
        t.tile0 = bn * (BN / kTileN) + wave * TPW;
        t.a = make_a_src<BM, NT>(p, t.m0, wave, lane);
        return t;
    };

    f4_t acc[MT][TPW];
    const DequantConsts dk = make_dequant_consts<BITS>();
    BStage<BITS, GPC, TPW> bst[D];
    const BSrc bsrc = make_b_src(p, lane, (size_t)p.tiles * p.chunks * (BITS == 4 ? 1024 : 2048), (size_t)p.tiles * p.G * 64);

    const int c_begin = blockIdx.z * p.chunks_per_split;
    const int c_end = min(p.chunks, c_begin + p.chunks_per_split);

    // LDS byte address of this lane's fragment row (the low 32 bits of a generic LDS pointer are the LDS offset)
    const uint32_t lds_row_base = (uint32_t)(uintptr_t)lds_all + (uint32_t)(c * 256);

    // D-stage pipeline over the 128-deep K chunks: chunk i lives in LDS buffer / register stage i % D, the loads of
    // chunks i+1 .. i+D-1 are in flight while chunk i is multiplied.  One chunk of a 128-row tile is only ~1000
    // matrix-pipe cycles per wave -- shorter than a loaded HBM round trip -- so D = 3 there; 256-row tiles (2 x 64 KiB
    // of LDS) keep D = 2.
#if defined(GPTQHIP_ABLATE_BLOAD)
    constexpr int OPS = BMP * 16 / NT;
#elif defined(GPTQHIP_ABLATE_META)
    constexpr int OPS = BMP * 16 / NT + TPW * (BITS == 4 ? 1 : 2);
#else
    constexpr int OPS = BMP * 16 / NT + TPW * ((BITS == 4 ? 1 : 2) + GPC);  // VMEM instructions per stage and wave
#endif
    // 16-byte store instructions per wave and tile in the epilogue (one column tile per wave on an odd number of row tiles: the last
    // 32-row store is half out of the tile's descriptor and dropped by the hardware)
    constexpr int NST = OUTF ? BM / RPP32 : (BM + RPP16 - 1) / RPP16;
    // Every issue is UNCONDITIONAL (a chunk index past the end is clamped and re-fetches the last chunk into a stage
    // nobody reads): the instruction stream between any load and its use is then the same on every path, which is what
    // lets both the hand-written and hipcc's own vmcnt waits be exact counts instead of full drains.
    auto issue = [&](auto sc, const TileCtx& t, int chunk) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const int ck = min(chunk, c_end - 1);
        stage_a_dma<BMP, NT>(t.a, lds_all + s * (BMP * 256), ck, wave);
        load_b<BITS, GPC, TPW>(bst[s], p, bsrc, t.tile0, ck);
    };
    auto prologue = [&](const TileCtx& t) __attribute__((always_inline)) {
        static_for<D - 1>([&](auto dc) { issue(dc, t, c_begin + decltype(dc)::value); });
    };

    // K-step fragments of B: bnow is carried ACROSS chunks -- K-step 0 of the next chunk is dequantised under the last
    // K-step's MFMAs of the current one, so that after the barrier the matrix pipe restarts after one LDS round trip
    // instead of after a VMEM issue burst + a dequant pass (all 8 waves leave the barrier together: nobody covers).
    u4_t bnow[TPW], bnext[TPW];
    // The per-(group, column) constants are expanded from the meta word ONCE per chunk (K-step 0) and kept for its other three
    // K-steps.  (Round 4: the stage's sched_barrier fences kept hipcc from sharing the expansion between the K-steps, so it ran four
    // times per chunk and tile -- ~9 of the ~22 VALU a K-step of one tile costs; the PMC pass that looked for the "B-load cost"
    // found VALU / SALU issue slots, not memory: profiles/r04_pmc_tiled_vmem.jsonl.)
    // (Not on 256-row tiles with bf16 scales or the fp32 epilogue: the kept constants -- four or five registers per tile there -- do not
    // fit beside the 128 accumulators and come back as scratch traffic; those instantiations keep the per-K-step expansion.)
    constexpr bool kHoistMeta = GPC == 1 && !(BM == 256 && (SCL == kBF16 || OUTF == 1));
    ColConst ccs[TPW];
    auto dequant_step = [&](const BStage<BITS, GPC, TPW>& bs, int j, u4_t (&b)[TPW]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
#ifdef GPTQHIP_TILED_NO_HOIST      // dev A/B build: the round-3 form everywhere
            constexpr bool kHoist = false;
#else
            constexpr bool kHoist = kHoistMeta;
#endif
            ColConst cnow;
            if constexpr (kHoist) {
                if (j == 0) ccs[t] = expand_meta<BITS, SCL>(bs.meta[t][0]);
            } else {
                cnow = expand_meta<BITS, SCL>(bs.meta[t][GPC == 4 ? j : 0]);
            }
            const ColConst& cc = kHoist ? ccs[t] : cnow;
            if constexpr (BITS == 4) {
                b[t] = dequant_word4<ACT, SCL>(bs.w[t][0][j], cc, dk);
            } else {
                b[t] = dequant_word8<ACT, SCL>(bs.w[t][j >> 1][(j & 1) * 2], bs.w[t][j >> 1][(j & 1) * 2 + 1], cc, dk);
            }
        }
    };
    // tile start: chunk 0's loads (a prologue stage: the previous tile's stores are younger) have landed -> K-step 0
    auto pre_first = [&]() __attribute__((always_inline)) {
        vm_wait<OPS, NST, D - 2>(D - 2, true);
        dequant_step(bst[0], 0, bnow);
    };

    // One pipeline stage = barrier, multiply chunk (stage slot s) while issuing chunk + D - 1 and preparing chunk + 1.
    //   barrier: every wave's pieces of this chunk's A tile have landed (each wave waited for its own loads before it
    //   got here) AND every wave is done reading buffer (s-1) % D, which this stage's DMA overwrites.  (Plain s_barrier:
    //   __syncthreads() would make hipcc drain vmcnt to 0.)
    //   issue: B loads right after the first fragment reads, the A-tile DMA pieces one per MFMA group (an LDS-DMA issue
    //   costs 60-185 cycles, MI355X_MICROARCH.md -- as one burst after the barrier it kept the matrix pipe idle).
    //   wait for chunk + 1, at the start of the last K-step: issue order per tile is [prologue stages 0..D-2] [NST output
    //   stores of the previous tile] [one stage per executed pipeline stage] and vmcnt retires in issue order, so chunk + 1
    //   has landed once at most AHEAD younger stages (+ the stores, if chunk + 1 is a prologue stage) are outstanding.
    // KIND 0: first round of a tile (stage 0 starts the accumulators from C = 0), 1: steady round, 2: drain (no issue).
    // Everything is compile-time unrolled with sched_barrier(0) fences; A fragments: groups of PF ds_read_b128 (inline
    // asm hipcc does not track; lds_wait ties the consumers to the counted wait), the reads of group g+1 in flight
    // under the MFMAs of group g; the next K-step's dequant VALU rides in the MFMA stream.
    auto stage = [&](auto sc, const TileCtx& t, int chunk, auto kind_c, auto pos_c) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        constexpr int kind = decltype(kind_c)::value;
        constexpr int pos = decltype(pos_c)::value;  // position inside the round / the drain
        constexpr bool kFirst = kind == 0 && pos == 0;
        constexpr bool kIssue = kind != 2;
        constexpr bool kNext = !(kind == 2 && pos == D - 2);  // loads of a next chunk exist (clamped past the end)
        constexpr int sn = (s + 1) % D, si = (s + D - 1) % D;
        // fragment group: 8 at BM = 128, 4 at 256 / 64, a whole K-step's MT fragments on the odd heights (PF must divide MT)
        constexpr int PF = BM == 128 ? 8 : (MT % 4 == 0 ? 4 : MT);  // measured: deeper spills at BM=256, helps at BM=128; 64-row tiles: MT = 4.  (Round 4: reading the
        // fragments two at a time on the 256-row 8-bit / per-K-step-constant instantiations, to free the registers they spill, made hipcc
        // spill MORE -- 60-132 bytes instead of 20-72 -- and was dropped.)
        constexpr int NG = 4 * MT / PF;        // fragment groups per chunk
        constexpr int NPIECE = BMP * 16 / NT;
#ifdef GPTQHIP_TILED_INTERLEAVE
        constexpr int kInterleaveValu = GPTQHIP_TILED_INTERLEAVE;   // dev A/B builds
#else
        // (round 5 A/B over the in-between heights: +2 % at 64 / 80 rows, 0 at 96, -2 % at 112; with one column tile per wave +2..3 % at
        // 48..96 rows, -1..2 % at 32 and 128)
        constexpr int kInterleaveValu = (TPW == 1 ? (BM >= 48 && BM <= 96) : BM <= 80) ? 4 : 0;
#endif
        static_assert(NPIECE <= NG, "one DMA piece per fragment group");
        u4_t abuf[2][PF];

        __builtin_amdgcn_s_barrier();
        // A fragment idx = j * MT + mt lives at row mt*16 + c, 16-byte segment (4j + rq) ^ c of this buffer
        const uint32_t abase = lds_row_base + (uint32_t)(s * (BMP * 256));
        uint32_t aaddr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) aaddr[j] = abase + (uint32_t)((j * 64 + rq * 16) ^ (c << 4));
        static_for<PF>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            lds_read_b128<(i % MT) * 4096>(abuf[0][i], aaddr[i / MT]);
        });
        const int ck = min(chunk + D - 1, c_end - 1);
        if constexpr (kIssue) load_b<BITS, GPC, TPW>(bst[si], p, bsrc, t.tile0, ck);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 1 < NG) {
                static_for<PF>([&](auto ic) {
                    constexpr int idx = (g + 1) * PF + decltype(ic)::value;
                    lds_read_b128<(idx % MT) * 4096>(abuf[(g + 1) & 1][decltype(ic)::value], aaddr[idx / MT]);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kIssue && g < NPIECE) stage_a_piece<BMP, NT, g>(t.a, lds_all + si * (BMP * 256), ck, wave);
            constexpr int j = (g * PF) / MT;            // K-step of this group (PF divides MT)
            constexpr bool last_of_step = ((g + 1) * PF) % MT == 0;
            if constexpr ((g * PF) % MT == 0) {  // VALU under this group's MFMAs
                if constexpr (j < 3) {
                    dequant_step(bst[s], j + 1, bnext);
                } else if constexpr (kNext) {
                    if constexpr (kind == 2) {
                        vm_wait<OPS, NST, D - 2>(D - 3 - pos, chunk + 1 - c_begin <= D - 2);
                    } else {
                        vm_wait<OPS, NST, D - 2>(D - 2, kind == 0 && pos + 1 <= D - 2);
                    }
                    dequant_step(bst[sn], 0, bnext);
                }
            }
            lds_wait<(g + 1 < NG) ? PF : 0, PF>(abuf[g & 1]);
            // 64-row tiles: the next K-step's dequant VALU (26 per 8 MFMAs there) is interleaved with this group's MFMAs instead of
            // running as one block in front of them (s_setprio is a scheduling boundary for hipcc, so such a group goes without it).
            // Measured (round 3, profiles/r03_tiled_ablation.txt): 4096x28672 at M=128 45 -> 41 us; no gain on 128- / 256-row tiles.
            constexpr bool kInterleaved = kInterleaveValu > 0 && (g * PF) % MT == 0 && (j < 3 || kNext);
            if constexpr (!kInterleaved) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int mt = (g * PF + i) % MT;
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) {
                    if constexpr (kFirst && j == 0) {
                        acc[mt][tt] = mfma16<ACT>(abuf[g & 1][i], bnow[tt], f4_t{0.f, 0.f, 0.f, 0.f});
                    } else {
                        acc[mt][tt] = mfma16<ACT>(abuf[g & 1][i], bnow[tt], acc[mt][tt]);
                    }
                }
            }
            if constexpr (kInterleaved) {
                static_for<PF * TPW>([&](auto) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA ...
                    __builtin_amdgcn_sched_group_barrier(0x002, kInterleaveValu, 0);   // ... then up to kInterleaveValu VALU
                });
            }
            if constexpr (!kInterleaved) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (last_of_step && (j < 3 || kNext)) {
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) bnow[tt] = bnext[tt];
            }
        });
    };

    // ---- epilogue pieces: buffer stores (hardware bounds check drops rows >= M, the lane offset of columns >= N is
    // pushed out of range) so that EVERY tile issues exactly the same number of store instructions -- the counted
    // waits above rely on it.
    // bias of the tile's columns: fetched BEFORE the next tile's prologue is issued (the wait for these few bytes would
    // otherwise sit behind that prologue's loads)
    auto load_bias = [&](const TileCtx& t, float (&bias)[TPW]) __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int n = (t.tile0 + tt) * kTileN + c;
            bias[tt] = (p.bias != nullptr && n < p.N) ? load16_as_f32<ACT>(p.bias, (size_t)n) : 0.f;
        }
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) asm volatile("" : "+v"(bias[tt]));  // loaded (and waited for) here, not later
    };
    auto store_tile = [&](const TileCtx& t, const float (&bias)[TPW]) __attribute__((always_inline)) {
        const int rows = min(p.M - t.m0, BM);
        // opaque copy of the lane id: keeps hipcc from hoisting the epilogue's address arithmetic out of the tile loop,
        // where it would occupy registers all through the main loop
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int c_e = lane_e & 15, rq_e = lane_e >> 4;
        if constexpr (OUTF == 0) {
            // 16-bit output: round like the reference, transpose through this wave's staging area (passes of TG16 row
            // tiles) and store whole 16-byte row pieces instead of 2-byte scattered elements
            uint16_t* slab = reinterpret_cast<uint16_t*>(lds_epi + wave * kEpi);
            char* base = reinterpret_cast<char*>(p.out) + (size_t)t.m0 * p.ldo * 2;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, rows * p.ldo * 2, 0x00020000);
            const int n0 = t.tile0 * kTileN + (lane_e % LPR16) * 8;
            const uint32_t lane_off = n0 < p.N ? (uint32_t)((lane_e / LPR16) * p.ldo * 2 + n0 * 2) : 0xFFFFFF00u;
#pragma unroll
            for (int h = 0; h < (MT + TG16 - 1) / TG16; ++h) {
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                    for (int mh = 0; mh < TG16; ++mh)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (h * TG16 + mh >= MT) continue;      // (odd MT with two row tiles per pass: rows past the tile, dropped on the way out)
                            float y = round_through<ACT>(acc[h * TG16 + mh < MT ? h * TG16 + mh : MT - 1][tt][i]);
                            if (p.bias != nullptr) y = y + bias[tt];
                            slab[(mh * 16 + 4 * rq_e + i) * WC + tt * 16 + c_e] = f32_to_16<ACT>(y);
                        }
                // same-wave LDS accesses execute in order: no barrier between this wave's writes and reads
#pragma unroll
                for (int pass = 0; pass < TG16 * 16 / RPP16; ++pass) {
                    const int row = pass * RPP16 + lane_e / LPR16;
                    const u4_t v = *reinterpret_cast<const u4_t*>(slab + row * WC + (lane_e % LPR16) * 8);
                    const uint32_t off = lane_off + (uint32_t)((h * (TG16 * 16) + pass * RPP16) * p.ldo * 2);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane_off >= 0xFFFFFF00u ? lane_off : off, 0, 0);
                }
            }
        } else {
            // fp32 accumulators (split-K partials or tensor-parallel partial sums): passes of TG32 row tiles, rows of 32
            // floats (128 B) leave as 16-byte pieces
            float* slab = reinterpret_cast<float*>(lds_epi + wave * kEpi);
            const size_t ld = p.splits > 1 ? (size_t)p.N : (size_t)p.ldo;
            char* base = p.splits > 1 ? reinterpret_cast<char*>(p.slabs + ((size_t)blockIdx.z * p.M + t.m0) * p.N)
                                      : reinterpret_cast<char*>(p.out) + (size_t)t.m0 * p.ldo * 4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(rows * ld * 4), 0x00020000);
            const int n0 = t.tile0 * kTileN + (lane_e % LPR32) * 4;
            const uint32_t lane_off = n0 < p.N ? (uint32_t)((lane_e / LPR32) * ld * 4 + n0 * 4) : 0xFFFFFF00u;
#pragma unroll
            for (int h = 0; h < MT / TG32; ++h) {
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                    for (int mh = 0; mh < TG32; ++mh)
#pragma unroll
                        for (int i = 0; i < 4; ++i) slab[(mh * 16 + 4 * rq_e + i) * WC + tt * 16 + c_e] = acc[h * TG32 + mh][tt][i];
#pragma unroll
                for (int pass = 0; pass < TG32 * 16 / RPP32; ++pass) {
                    const int row = pass * RPP32 + lane_e / LPR32;
                    const f4_t v = *reinterpret_cast<const f4_t*>(slab + row * WC + (lane_e % LPR32) * 4);
                    const uint32_t off = lane_off + (uint32_t)((h * (TG32 * 16) + pass * RPP32) * ld * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rs,
                                                           lane_off >= 0xFFFFFF00u ? lane_off : off, 0, GPTQHIP_SLAB_AUX);
                }
            }
        }
    };

    TileCtx cur = make_ctx(blockIdx.x);
    prologue(cur);
    {
        // the first tile has no predecessor whose stores sit between its prologue and its later stages: issue the same
        // number of (out-of-range, dropped) stores so that every tile sees the same instruction sequence
        const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);
        const u4_t z = {0u, 0u, 0u, 0u};
        static_for<NST>([&](auto ic) {  // NST distinct instructions (a rolled-up loop would hide their count from hipcc)
            __builtin_amdgcn_raw_buffer_store_b128(z, none, 16 * decltype(ic)::value, 0, 0);
        });
    }
    for (int v = blockIdx.x;; v += G) {
        int chunk0 = c_begin;
        pre_first();
        if (c_end - c_begin >= D) {
            static_for<D>([&](auto sc) {
                stage(sc, cur, chunk0 + decltype(sc)::value, std::integral_constant<int, 0>{}, sc);
            });
            chunk0 += D;
            while (chunk0 + D <= c_end) {
                static_for<D>([&](auto sc) {
                    stage(sc, cur, chunk0 + decltype(sc)::value, std::integral_constant<int, 1>{}, sc);
                });
                chunk0 += D;
            }
        } else {
            f4_t zero = {0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+v"(zero));  // fewer than D chunks (tiny K): rare path, plain zeroing
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[mt][t] = zero;
        }
        // drain: the last c_end - chunk0 < D chunks; chunk0 - c_begin is a multiple of D, so chunk0 + i uses stage i
        static_for<D - 1>([&](auto ic) {
            if (chunk0 + decltype(ic)::value < c_end) {
                stage(ic, cur, chunk0 + decltype(ic)::value, std::integral_constant<int, 2>{}, ic);
            }
        });
        // every wave is done reading the stage buffers -> the next tile's first stages may land in them while this
        // tile's output is rounded, transposed and stored; those stores then drain under the next tile's first chunks
        __builtin_amdgcn_s_barrier();
        float bias[TPW];
        load_bias(cur, bias);
        if (v + G >= ntiles) {
            store_tile(cur, bias);
            break;
        }
        const TileCtx nxt = make_ctx(v + G);
        prologue(nxt);
        store_tile(cur, bias);
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------
template <int BITS, int ACT, int SCL, int GPC, int OUTF>
inline int launch_tiled_out(const TiledParams& p, int bm, hipStream_t stream) {
    // one block per CU; with more tiles than CUs the blocks loop over tiles (persistent), keeping the next tile's first
    // loads in flight across the epilogue.  Split-K launches have few tiles by construction: one tile per block.
    const int ntiles = ceil_div(p.N, kTiledBN) * ceil_div(p.M, bm);
    const dim3 grid(p.splits == 1 && ntiles > 256 ? 256 : ntiles, 1, p.splits);
    if ((size_t)p.tiles * p.chunks * (BITS == 4 ? 1024 : 2048) >= ((size_t)1 << 31) || (size_t)p.tiles * p.G * 64 >= ((size_t)1 << 31)) {
        set_error("tiled kernel: packed weights of one layer must stay below 2 GiB (32-bit buffer offsets)");
        return -22;  // GPTQHIP_EINVAL
    }
    {
        // 256-row tiles for every variant.  The 8-bit and per-K-step-group-constant stages spill 11-14 VGPRs beside the 128
        // accumulator registers (profiles/r02_tiled_bm256_isa.txt) and are STILL 19-36 % faster than on 128-row tiles
        // (profiles/r02_tiled_variants.txt: 8-bit 979 -> 1213 TF, group 64 828 -> 1128 TF at M=8192 4096^2): twice the MFMA work
        // per dequantised word outweighs a few scratch accesses per chunk.
        if (bm == 256) {
            hipLaunchKernelGGL((tiled_kernel<BITS, ACT, SCL, GPC, 256, 8, 2, OUTF>), grid, dim3(512), 0, stream, p);
            return check_hip(hipGetLastError(), "tiled_kernel launch");
        }
    }
    if (bm == 64) {  // M <= 64: half the MFMA work and staging of a 128-row tile
        // 2 stages.  Same-box A/B of 2 / 3 / 4 / 5 stages (round 3, profiles/r03_tiled_stages_64row.txt): the 64-row tile only runs where
        // blocks are short (split-K: 4..14 chunks each) or the grid is one round of a wide layer, and there every extra stage is a
        // longer prologue burst that never pays back -- 4096^2 at M=128 15.5 / 16.6 / 17.6 / 17.9 us, 14336x4096 27.2 / 28.2 / 28.5 / 28.9,
        // gate_up at M=128 equal (40.7 / 40.6 / 41.4 / 43.0).  (Five stages were tried first on the theory that a block with so little
        // MFMA work per chunk is bound by its bytes in flight: it is not.)  The waits are generalised to any depth (vm_wait) and the
        // depth is one macro, so the experiment is one -D away.
        constexpr int D64 = BITS == 4 ? kTiledD64 : 2;
        hipLaunchKernelGGL((tiled_kernel<BITS, ACT, SCL, GPC, 64, 8, D64, OUTF>), grid, dim3(512), 0, stream, p);
        return check_hip(hipGetLastError(), "tiled_kernel launch");
    }
    if (bm != 128) {
        set_error("tiled kernel: no %d-row tile for bits=%d gpc=%d", bm, BITS, GPC);
        return -22;  // GPTQHIP_EINVAL
    }
    // (a 4-wave x 4-tile variant with two independent blocks per CU was measured 25-30 % slower: 660 vs 950 TF)
    // 3 stages in flight (measured on 128-row tiles, M=2048 4096^2: 951 / 1061 / 984 TF for 2 / 3 / 4 stages -- the
    // fourth only lengthens the start-up burst); the 8-bit register stages are twice as large: 2 stages there
#ifndef GPTQHIP_TILED_D128
#define GPTQHIP_TILED_D128 3
#endif
    hipLaunchKernelGGL((tiled_kernel<BITS, ACT, SCL, GPC, 128, 8, BITS == 4 ? GPTQHIP_TILED_D128 : 2, OUTF>), grid, dim3(512), 0, stream, p);
    return check_hip(hipGetLastError(), "tiled_kernel launch");
}

template <int BITS, int ACT, int SCL, int OUTF>
inline int launch_tiled_gpc(const TiledParams& p, int gpc, int bm, hipStream_t stream) {
    if (gpc == 1) return launch_tiled_out<BITS, ACT, SCL, 1, OUTF>(p, bm, stream);
    return launch_tiled_out<BITS, ACT, SCL, 4, OUTF>(p, bm, stream);
}

// one translation unit per (bit width, epilogue) so the instantiations compile in parallel:
// gptqhip_tiled.hip (4-bit, 16-bit output), gptqhip_tiled_f32.hip (4-bit, fp32 output), gptqhip_tiled8.hip (8-bit)
int launch_tiled_w4(const TiledParams& p, int act_dtype, int scale_dtype, int gpc, int bm, hipStream_t stream);
int launch_tiled_w4_f32(const TiledParams& p, int act_dtype, int scale_dtype, int gpc, int bm, hipStream_t stream);
int launch_tiled_w8(const TiledParams& p, int act_dtype, int scale_dtype, int gpc, int bm, hipStream_t stream);
// the extra tile heights (gptqhip_tiled_r<rows>.hip): 4-bit, one group constant per chunk
int launch_tiled_w4_r32(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_r48(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_r80(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_r96(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_r112(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
// 128-column blocks (gptqhip_tiled_n128_r<rows>.hip): 4-bit, one group constant per chunk, 32..128-row tiles
int launch_tiled_w4_n128_r32(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_n128_r48(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_n128_r64(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_n128_r80(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_n128_r96(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_n128_r112(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);
int launch_tiled_w4_n128_r128(const TiledParams& p, int act_dtype, int scale_dtype, int out_f32, hipStream_t stream);

template <int BITS, int OUTF>
inline int launch_tiled_bits(const TiledParams& p, int act_dtype, int scale_dtype, int gpc, int bm, hipStream_t stream) {
    if (act_dtype == kFP16 && scale_dtype == kFP16) return launch_tiled_gpc<BITS, kFP16, kFP16, OUTF>(p, gpc, bm, stream);
    if (act_dtype == kBF16 && scale_dtype == kFP16) return launch_tiled_gpc<BITS, kBF16, kFP16, OUTF>(p, gpc, bm, stream);
    if (act_dtype == kFP16 && scale_dtype == kBF16) return launch_tiled_gpc<BITS, kFP16, kBF16, OUTF>(p, gpc, bm, stream);
    return launch_tiled_gpc<BITS, kBF16, kBF16, OUTF>(p, gpc, bm, stream);
}

}  // namespace gptqhip
