// Skinny fused dequant-GEMM for decode / small batch (M <= 32): HBM-bound streaming of the packed
// weights with the dequantisation fused into the MFMA contraction.
//
// Replaces the reference hot loop TorchLinear._forward_eager (gptqmodel/nn_modules/qlinear/torch.py:326-347)
// which materialises the whole fp16 [K,N] weight (torch.py:700-717) and then calls aten matmul.
//
// Mapping (wave64, mfma_f32_16x16x32, tile-major layout of gptqhip_device.h):
//   * A block owns ONE 16-column tile; its W waves (4..16) split the tile's K range chunk by chunk
//     (chunk = 128 rows).  One dwordx4 per lane = one (tile, chunk) block = 1 KiB contiguous; wave w reads
//     chunk c0+w, c0+w+W, ... so the block streams one linear address range and every byte is fetched exactly
//     once (non-temporal: no reuse).
//   * Each int32 word is one lane's B fragment (8 consecutive k of its column): HBM -> VGPR ->
//     (and_or magic, pk_add/pk_fma, pk_mul) -> MFMA operand.  No LDS, no cross-lane traffic for weights.
//   * A D-deep register ring per wave keeps D chunks (weights + constants + activations) in flight; in the
//     "regular" case (the planner picks W so every wave owns a multiple of D chunks) the loop is straight-line with
//     unconditional loads so hipcc emits COUNTED s_waitcnt vmcnt(N) instead of draining the queue.
//   * Activations (x is M*K*2 bytes, L2 resident; natural k order matches the tiled nibble order) are loaded in
//     full cache lines with the ring, parked in the wave's private LDS slot and read back as MFMA A fragments
//     (AM_ROW1 / AM_ROW4 / AM_ROWS below).  Rows >= M only feed output rows nobody stores: no masking.
//   * Reduction: the W partial accumulators meet in LDS (in-block split-K).  Only when N/16 tiles cannot occupy
//     the chip does K also split across blocks (grid.y): fp32 slabs published with write-through (sc1) stores + one
//     relaxed agent-scope ticket, reduced by the last arriver in a fixed order (deterministic, no float atomics)
//     -- cdna_hip_programming.md §5 "in-launch split-K reduction".
#include <stdlib.h>

#include <type_traits>

#include "gptqhip_device.h"
#include "gptqhip_host.h"

#ifndef GPTQHIP_ABLATE   // dev-only timing ablations (tests/dev/ablate.sh); the product build never defines it
#define GPTQHIP_ABLATE 0
#endif
#ifndef GPTQHIP_SK1P_XREG
#define GPTQHIP_SK1P_XREG 0
#endif
#ifndef GPTQHIP_OUT_WRITE_THROUGH   // dev A/B (see finish_outputs): write-through output stores measured SLOWER (o 4.19 -> 4.37, down 7.87 -> 7.98 us)
#define GPTQHIP_OUT_WRITE_THROUGH 0
#endif

namespace gptqhip {

// Threads a block may have (the instantiation's __launch_bounds__): 1024 (16 waves, 128 VGPRs) up to 16 rows, 512 (8 waves) for
// 17..64 rows.  Measured in round 3 (profiles/r03_mid_m_sweep.txt): with the split-ring pipeline the 17..32-row instantiations
// need only 88-132 VGPRs and COULD run 16 waves, but 8 waves per block are as fast on 4096^2 / 14336x4096 and 7-25 % faster on
// 4096x6144 / 4096x28672 (twice the blocks per CU re-stage twice the activation tiles).  profiles/r03_isa_audit.txt has the table.
template <int BITS, int MT>
__host__ __device__ constexpr int skinny_launch_bound() {
    return MT >= 2 ? 512 : 1024;
}
static int skinny_max_waves(int bits, int mt) { (void)bits; return mt >= 2 ? 8 : 16; }

// Occupancy hint: none.  (Round 3 measured one: the bf16 batch-1 instantiations with the 4-deep ring need 82-84 VGPRs = 5 waves per SIMD
// where fp16 needs 67 = 7, so a Llama-3-8B gate_up -- 7 four-wave blocks per CU -- runs in two rounds.  amdgpu_waves_per_eu(6) brings
// them to 78-80 VGPRs without scratch, (7) spills 20 registers; with (6) the bf16 chain got SLOWER, 771 vs 808 tokens/s against 886 / 898
// fp16 on the same boxes: the squeezed schedule costs more than the sixth wave brings.)
template <int BITS, int ACT, int MT, int AM, int D>
__host__ __device__ constexpr int skinny_min_waves_per_simd() {
    return 1;
}

struct SkinnyParams {
    const void* x;
    const int32_t* perm;   // act-order row permutation applied to x inside the kernel (AM_ROW1P), else nullptr
    const uint32_t* qw;    // tiled words
    const uint32_t* meta;  // [tiles][G][16]
    const void* bias;
    void* out;
    float* slabs;
    int* counters;
    int M, K, N, G, group_size;
    int chunks;            // ceil(K/128)
    int chunks_per_split;  // chunks handled by one block
    int splits;
    int out_f32;           // epilogue writes raw fp32 accumulators (TP partial sums)
    int n_mine;            // regular pipeline: chunks per wave incl. padding = rounds * D
    int regular;           // every wave owns a multiple of D chunks: straight-line counted-wait pipeline
    int cpg_shift;         // log2(chunks per group) when group_size is 128 * 2^n, else -1 (integer division)
    int exact_bf16;        // GPTQHIP_GEMM_EXACT_BF16: see compute_stage
    int alg_fp16;          // decode form 2: algebraic dequant for fp16 activations (compute_stage)
    int slot_stride;       // skinny1_kernel: bytes of a wave's LDS slot
    int nb_tiles;          // skinny1p_kernel: > 0 = tiles per block, reduction rows kept per tile and combined WITHOUT block barriers
    // batch-1 decode op (gptqhip_decode_linear): decoder-layer glue fused into the GEMV (GLUE template parameter)
    const void* glue_b;    // RMSNORM: norm weight [K]; SILU_MUL: nullptr (up = x + K)
    const void* residual;  // [N] or nullptr: out = act(residual + y)
    const float* stats_in; // RMSNORM: per-producer-tile sums of h^2 ([stats_n] floats) written by the op that produced h; nullptr: the
                           // block reduces the row itself
    float* stats_out;      // [ceil(N/16)] or nullptr: sum over this tile's 16 outputs of out^2 (the next op's RMSNorm statistic)
    float eps;
    int in_glue;
    int out_glue;          // kOutSiluMul: column tiles hold 8 gate + 8 up columns; out[N/2] = act(silu(gate)) * up
    int stats_n;
};
constexpr int kOutNone = 0;
constexpr int kOutSiluMul = 1;

// Input glue of the batch-1 decode op: what a Llama-style decoder layer computes between two quantised linears, applied
// to the activation pair of each ring stage on its way into the MFMA A fragment (semantics = HF LlamaRMSNorm / LlamaMLP in
// the activation dtype).
constexpr int kGlueNone = 0;
constexpr int kGlueRmsNorm = 1;  // x' = w * act(h32 * rsqrt(mean(h32^2) + eps))
constexpr int kGlueSiluMul = 2;  // x' = act(silu(gate)) * up, x = gate | up

template <int ACT>
__device__ __forceinline__ uint32_t glue_pair(uint32_t a, uint32_t b, float inv, int glue) {
    if constexpr (ACT == kFP16) {
        if (glue == kGlueRmsNorm) {
            // fp16: three VALU ops per pair instead of ~14.  v_fma_mix{lo,hi}_f16 read the f16 half of h and the f32 inv,
            // multiply in f32 and round ONCE to f16 (== (h.float() * inv).to(fp16)); v_pk_mul_f16 by the weight pair is the
            // rounded fp16 product HF's `weight * hidden` computes.
            uint32_t t = 0u;
            asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "+v"(t) : "v"(a), "v"(inv));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(a), "v"(inv));
            return as_u32(as_h2(b) * as_h2(t));
        }
    }
    const float a0 = bits16_to_f32<ACT>((uint16_t)(a & 0xffffu)), a1 = bits16_to_f32<ACT>((uint16_t)(a >> 16));
    const float b0 = bits16_to_f32<ACT>((uint16_t)(b & 0xffffu)), b1 = bits16_to_f32<ACT>((uint16_t)(b >> 16));
    float r0, r1;
    if (glue == kGlueRmsNorm) {
        r0 = b0 * round_through<ACT>(a0 * inv);
        r1 = b1 * round_through<ACT>(a1 * inv);
    } else {
        r0 = round_through<ACT>(a0 / (1.0f + expf(-a0))) * b0;   // expf, not __expf: HF evaluates SiLU with the accurate exp
        r1 = round_through<ACT>(a1 / (1.0f + expf(-a1))) * b1;
    }
    return (uint32_t)f32_to_16<ACT>(r0) | ((uint32_t)f32_to_16<ACT>(r1) << 16);
}

// AM: how a wave gets its activations.
//   AM_ROW1  (M == 1):  ONE 4-byte load per lane per chunk (the chunk's 256 B of the single row), prefetched with the
//                       weights (1 VGPR per ring stage); at compute time the wave parks them in its private LDS slot
//                       and reads the four MFMA A fragments back with broadcast ds_read_b128 (all 16 fragment rows
//                       see row 0: rows >= M feed output rows nobody stores).
//   AM_ROW4  (M <= 4):  the same with 16 bytes per lane (lane = row*16 + segment: 4 rows x 256 B).
//   AM_ROWS  (M <= 32): 4*MT loads of 16 B per lane, each covering 4 rows x 256 B in FULL cache lines; the wave
//                       parks the 16*MT x 128 tile in its LDS slot with rows padded to 272 B so the fragment
//                       ds_read_b128 (16 rows, same 16-B column) is bank-conflict free.  (Fragment-shaped global
//                       loads touch 16 half-used lines per instruction and made M=8 2.6x slower than M=1.)
constexpr int AM_ROW4 = 0;
constexpr int AM_ROW1 = 2;
constexpr int AM_ROWS = 3;
constexpr int AM_ROWSH = 4;  // AM_ROWS with the last two (unused) row quads skipped: M <= 16*MT - 8
constexpr int AM_ROW1P = 5;  // AM_ROW1 for act-order checkpoints (no separate gather launch per linear): the block copies the
                             // x row into LDS once (coalesced), the ring carries the chunk's permutation pair per lane and the
                             // two halves are gathered from LDS at compute time (2-byte gathers straight from global memory
                             // were measured 1.4-1.9x slower than the un-permuted kernel: 64 distinct lines per wave load)
constexpr int kRowsPitch = 17;  // u4 per padded LDS row (272 B)

template <int AM, int MT>
__host__ __device__ constexpr int slot_bytes() {
    return (AM == AM_ROW1 || AM == AM_ROW1P) ? 256
                         : (AM == AM_ROW4 ? 1024 : ((AM == AM_ROWS || AM == AM_ROWSH) ? 16 * MT * kRowsPitch * 16 : 0));
}
template <int AM>
__host__ __device__ constexpr bool is_rows() {
    return AM == AM_ROWS || AM == AM_ROWSH;
}
template <int AM, int MT>
__host__ __device__ constexpr int row_quads() {  // 4-row groups actually loaded per chunk
    return AM == AM_ROWSH ? 4 * MT - 2 : 4 * MT;
}

template <int AM, int MT>
struct AStage {
    // AM_ROW4: [1] = the norm-weight segment of the decode op's RMSNorm glue; AM_ROWS with MT == 1 (decode op on 9..16 rows): [4] =
    // the same (AM_ROWSH keeps it in the unused quad [3]); unused -- and never allocated -- otherwise
    u4_t a[AM == AM_ROW4 ? 2 : (AM == AM_ROWS && MT == 1 ? 5 : 4 * MT)];
};
template <int MT>
struct AStage<AM_ROW1, MT> {
    uint32_t a[2];  // [1]: second glue operand (norm weight pair / up pair) of the decode op, unused otherwise
};
template <int MT>
struct AStage<AM_ROW1P, MT> {
    uint32_t a[2];  // perm[k], perm[k+1] of this lane's pair in the chunk
};

template <int BITS, int GPC, int MT, int AM>
struct Stage {
    u4_t w[BITS == 4 ? 1 : 2];
    uint32_t meta[GPC];
    AStage<AM, MT> x;
};

__device__ __forceinline__ int group_of(const SkinnyParams& p, int k) {  // k is wave-uniform
    int g;
    if (p.cpg_shift >= 0) {
        g = k >> (7 + p.cpg_shift);
    } else {
        g = k / p.group_size;
    }
    return g < p.G ? g : p.G - 1;  // padded rows beyond K: any finite scale (their activations are 0)
}

// Addressing: every base below is wave-uniform (kernel arguments, blockIdx, the readfirstlane'd wave id) and every
// per-lane part is a 32-bit byte offset, so the loads select the scalar-base + vector-offset form and spend no
// 64-bit VALU adds.
struct TileBases {
    const char* w;         // this tile's first (tile, chunk) block
    const uint32_t* meta;  // this tile's [G][16] constants
    const char* x;         // activations
    uint32_t lane16;       // lane * 16
    uint32_t c4;           // (lane & 15) * 4
    __amdgpu_buffer_rsrc_t xrsrc;  // 17..64-row instantiations: descriptor over the M valid rows of x
    uint32_t quad_stride;          //   bytes between row quads (4 rows)
};

template <int BITS, int GPC, int MT, int AM>
__device__ __forceinline__ void load_stage(Stage<BITS, GPC, MT, AM>& st, const SkinnyParams& p, const TileBases& tb,
                                           int chunk, int lane) {
    constexpr int WPC = BITS == 4 ? 1 : 2;
    const char* src = tb.w + (size_t)chunk * (WPC * 1024);
#pragma unroll
    for (int h = 0; h < WPC; ++h)
        st.w[h] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(src + h * 1024 + tb.lane16));
#pragma unroll
    for (int j = 0; j < GPC; ++j) {
        const uint32_t* mrow = tb.meta + group_of(p, chunk * kChunkK + j * (kChunkK / GPC)) * 16;
        st.meta[j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(mrow) + tb.c4);
    }
    // activations: no masking anywhere.  Rows >= M only feed output rows nobody stores (address clamped to row 0);
    // k >= K happens only in the zero-padded tail chunk of a ragged K, whose weights dequantise to exactly 0
    // (repack_tiled stores code == zero-point there), so the clamped address may read any finite x.
    if constexpr (AM == AM_ROW1P) {
        int k0 = chunk * kChunkK + 2 * lane;
        k0 = k0 + 1 < p.K ? k0 : 0;
        st.x.a[0] = p.perm[k0];
        st.x.a[1] = p.perm[k0 + 1];
    } else if constexpr (AM == AM_ROW1) {
        uint32_t off = (uint32_t)chunk * 256u + (uint32_t)lane * 4u;       // bytes into row 0
        off = off < (uint32_t)p.K * 2u ? off : 0u;
        st.x.a[0] = *reinterpret_cast<const uint32_t*>(tb.x + off);
    } else if constexpr (AM == AM_ROW4) {
        int row = lane >> 4;
        row = row < p.M ? row : 0;
        int k0 = chunk * kChunkK + 8 * (lane & 15);
        k0 = k0 < p.K ? k0 : 0;
        st.x.a[0] = *reinterpret_cast<const u4_t*>(tb.x + ((size_t)row * p.K + k0) * 2);
    } else if constexpr (is_rows<AM>()) {
        int k0 = chunk * kChunkK + 8 * (lane & 15);
        k0 = k0 < p.K ? k0 : 0;
#pragma unroll
        for (int i = 0; i < row_quads<AM, MT>(); ++i) {
            int row = 4 * i + (lane >> 4);
            row = row < p.M ? row : 0;
            st.x.a[i] = *reinterpret_cast<const u4_t*>(tb.x + ((size_t)row * p.K + k0) * 2);
        }
    }
}

// FAST loader for the regular pipeline (K % 128 == 0, group_size = 128 * 2^n, every wave owns a multiple of D
// chunks): a wave-uniform cursor (scalar pointers advanced by constant strides) + per-lane 32-bit offsets computed
// once.  No clamps, no integer division, no 64-bit vector address arithmetic in the loop.
struct Cursor {
    const char* w;  // next (tile, chunk) block this wave loads
    const char* x;  // activations, advanced to that chunk (row 0)
    int chunk;
    int end;        // first chunk past this block's REAL K range: chunks >= end are padding of the last ring round (the planner
                    // rounds every wave up to whole rounds); their loads are clamped to the last real chunk (wave-uniform scalar
                    // selects, no branch: the waits stay counted) and their stage is skipped at compute time
};

template <int MT, int AM>
struct LaneOffs {
    uint32_t x[is_rows<AM>() && MT == 1 ? 4 : 1];  // byte offsets of this lane's activation loads
};

// LOAD_W / LOAD_A: the split-ring pipeline of the 17..64-row instantiations (kSplitRing) loads the weight stage and the activation
// stage through two cursors; everything else loads both through one.
template <int BITS, int GPC, int MT, int AM, int GLUE = 0, bool LOAD_W = true, bool LOAD_A = true>
__device__ __forceinline__ void load_stage_fast(Stage<BITS, GPC, MT, AM>& st, const SkinnyParams& p, const TileBases& tb,
                                                const LaneOffs<MT, AM>& lo, Cursor& cu, int stride_chunks) {
    constexpr int WPC = BITS == 4 ? 1 : 2;
    const int over = cu.chunk - (cu.end - 1);
    const size_t back = over > 0 ? (size_t)over : 0;                 // 0 for every real chunk
    const char* wsrc = cu.w - back * (WPC * 1024);
    const char* xsrc = cu.x - back * 256;
    const int ch = cu.chunk - (int)back;
    if constexpr (LOAD_W) {
#pragma unroll
    for (int h = 0; h < WPC; ++h)
        st.w[h] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(wsrc + h * 1024 + tb.lane16));
    if constexpr (GPC == 1) {
        const char* mrow = reinterpret_cast<const char*>(tb.meta) + ((size_t)(ch >> p.cpg_shift) << 6);
#if GPTQHIP_ABLATE & 1
        st.meta[0] = 0x00082000u;
#else
        st.meta[0] = *reinterpret_cast<const uint32_t*>(mrow + tb.c4);
#endif
    } else {
        // group_size 32 / 64: one constant row per 32-row K-step; cpg_shift holds log2(group_size / 32) here
#pragma unroll
        for (int j = 0; j < GPC; ++j) {
            const char* mrow = reinterpret_cast<const char*>(tb.meta) + ((size_t)((ch * 4 + j) >> p.cpg_shift) << 6);
            st.meta[j] = *reinterpret_cast<const uint32_t*>(mrow + tb.c4);
        }
    }
    }   // LOAD_W
    if constexpr (!LOAD_A) {
        // (weights only)
    } else if constexpr (AM == AM_ROW1) {
#if GPTQHIP_ABLATE & 2
        st.x.a[0] = 0x3c003c00u;
#else
        st.x.a[0] = *reinterpret_cast<const uint32_t*>(xsrc + lo.x[0]);
#endif
        if constexpr (GLUE == kGlueRmsNorm) {
            st.x.a[1] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.glue_b) + (xsrc - tb.x) + lo.x[0]);
        } else if constexpr (GLUE == kGlueSiluMul) {
            st.x.a[1] = *reinterpret_cast<const uint32_t*>(xsrc + (size_t)p.K * 2 + lo.x[0]);
        }
    } else if constexpr (AM == AM_ROW1P) {
        const u2_t pr = *reinterpret_cast<const u2_t*>(reinterpret_cast<const char*>(p.perm) + (size_t)ch * 512 + lo.x[0]);
        st.x.a[0] = pr.x;
        st.x.a[1] = pr.y;
    } else if constexpr (AM == AM_ROW4) {
        st.x.a[0] = *reinterpret_cast<const u4_t*>(xsrc + lo.x[0]);
        if constexpr (GLUE == kGlueRmsNorm) {   // the same 16-byte weight segment for all four rows
            st.x.a[1] = *reinterpret_cast<const u4_t*>(reinterpret_cast<const char*>(p.glue_b) + (xsrc - tb.x) + tb.c4 * 4u);
        }
    } else if constexpr (is_rows<AM>() && MT >= 2) {
        // 17..64 rows: buffer loads -- ONE per-lane offset register (row quad 0) + a scalar offset per quad and chunk instead of
        // 4 * MT offset registers; rows >= M are outside the descriptor and come back as zeros (no clamps)
        const uint32_t soff = (uint32_t)(xsrc - tb.x);
#pragma unroll
        for (int i = 0; i < row_quads<AM, MT>(); ++i)
            st.x.a[i] = __builtin_amdgcn_raw_buffer_load_b128(tb.xrsrc, lo.x[0], soff + (uint32_t)i * tb.quad_stride, 0);
    } else if constexpr (is_rows<AM>()) {
#pragma unroll
        for (int i = 0; i < row_quads<AM, MT>(); ++i) st.x.a[i] = *reinterpret_cast<const u4_t*>(xsrc + lo.x[i]);
        if constexpr (is_rows<AM>() && MT == 1 && GLUE == kGlueRmsNorm) {
            // decode op on 5..8 rows: the norm-weight segment rides in the stage's last (unused: quads 2, 3 are skipped) register;
            // on 9..16 rows (AM_ROWS) in a fifth one
            st.x.a[AM == AM_ROWSH ? 3 : 4] = *reinterpret_cast<const u4_t*>(reinterpret_cast<const char*>(p.glue_b) + (xsrc - tb.x) + tb.c4 * 4u);
        }
    }
    cu.w += (size_t)stride_chunks * (WPC * 1024);
    cu.x += (size_t)stride_chunks * 256;
    cu.chunk += stride_chunks;
}

template <int MT, int AM, int GLUE>
__host__ __device__ constexpr bool kSplitRing() {
    return MT >= 2 && (AM == AM_ROWS || AM == AM_ROWSH) && GLUE == 0;
}

struct GlueInv {
    float v[4] = {0.f, 0.f, 0.f, 0.f};   // RMSNorm 1/rms of rows rq, 4 + rq, 8 + rq, 12 + rq (the lane's row of each loaded quad)
};

template <int BITS, int ACT, int GPC, int AM>
__host__ __device__ constexpr bool kExactBf16() {
    // bf16 activations only.  The same form for fp16 ((nibble | 0x6400) = 1024 + q) was built and measured in round 2: as a runtime
    // branch it slowed the DEFAULT fp16 kernels by 2 % on narrow layers and gained 2.8 % when on; as separate instantiations it
    // left the default alone and ran 5 % SLOWER than it -- the bit-faithful fp16 dequant (13 VALU per word) is not what holds
    // decode back, so there is no fp16 variant.
    return BITS == 4 && ACT == kBF16 && GPC == 1 && (AM == AM_ROW1 || AM == AM_ROW1P || AM == AM_ROW4);
}

// PHASE 0: the whole stage; 1: only park the stage's activations in the wave's LDS slot; 2: only dequantise + multiply (the
// activations are in the slot already) -- the split-ring pipeline (kSplitRing) runs 1 and 2 on different register stages.
template <int BITS, int ACT, int SCL, int MT, int GPC, int AM, int GLUE = 0, int PHASE = 0>
__device__ __forceinline__ void compute_stage(const Stage<BITS, GPC, MT, AM>& st, const SkinnyParams& p, int chunk,
                                              int lane, u4_t* aslot, const DequantConsts& dk, f4_t (&acc)[MT],
                                              const uint16_t* xbuf = nullptr, const GlueInv& gi = GlueInv{}) {
    const float inv = gi.v[0];
    const int c = lane & 15;
    const int rq = lane >> 4;
    int abase = 0;  // u4 index of this lane's fragment row inside the wave's LDS slot
#if GPTQHIP_ABLATE & 4
    if constexpr (AM == AM_ROW1 && BITS == 4) {
        const uint32_t t = st.w[0].x ^ st.w[0].y ^ st.w[0].z ^ st.w[0].w ^ st.meta[0] ^ st.x.a[0];
        acc[0][0] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, acc[0][0]) ^ t);
        return;
    }
#endif
    if constexpr (AM == AM_ROW1) {
        if constexpr (GLUE == kGlueNone) {
            reinterpret_cast<uint32_t*>(aslot)[lane] = st.x.a[0];
        } else {
            reinterpret_cast<uint32_t*>(aslot)[lane] = glue_pair<ACT>(st.x.a[0], st.x.a[1], inv, GLUE);
        }
    } else if constexpr (AM == AM_ROW1P) {
        reinterpret_cast<uint32_t*>(aslot)[lane] = (uint32_t)xbuf[st.x.a[0]] | ((uint32_t)xbuf[st.x.a[1]] << 16);
    } else if constexpr (AM == AM_ROW4) {
        if constexpr (GLUE == kGlueRmsNorm) {   // `inv` = this lane's row's 1/rms (rows >= M carry row 0's: nobody stores them)
            u4_t g;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = glue_pair<ACT>(st.x.a[0][j], st.x.a[1][j], inv, GLUE);
            aslot[lane] = g;
        } else {
            aslot[lane] = st.x.a[0];
        }
        abase = (c < p.M ? c : 0) << 4;  // lanes of unused rows re-read row 0: a broadcast, no extra bank traffic
    } else if constexpr (is_rows<AM>() && PHASE != 2) {
        // rows of skipped quads keep whatever the slot held: they only feed output rows >= M, which nobody stores
        if constexpr (MT == 1 && GLUE == kGlueRmsNorm) {
#pragma unroll
            for (int i = 0; i < row_quads<AM, MT>(); ++i) {   // row 4 i + rq with its own 1/rms (gi.v[i])
                u4_t g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = glue_pair<ACT>(st.x.a[i][j], st.x.a[AM == AM_ROWSH ? 3 : 4][j], gi.v[i], GLUE);
                aslot[(4 * i + rq) * kRowsPitch + c] = g;
            }
        } else {
#pragma unroll
            for (int i = 0; i < row_quads<AM, MT>(); ++i) aslot[(4 * i + rq) * kRowsPitch + c] = st.x.a[i];
        }
    }
    if constexpr (PHASE == 1) return;
    if (kExactBf16<BITS, ACT, GPC, AM>() && p.exact_bf16) {
        // OPT-IN (GPTQHIP_GEMM_EXACT_BF16; block-uniform branch).  bf16 activations, 4-bit codes, one group per chunk,
        // at most 4 rows: gfx950 has no packed bf16 VALU, so the per-weight dequant (cvt, mul, cvt_pk: 28 VALU per word)
        // is replaced by linear algebra on the matrix pipe.
        // (nibble | 0x4300) is the bf16 number 128 + q exactly, so with G = this chunk's group
        //     sum_k x_k s_G (q_k - z_G) = s_G * ( sum_k x_k (128 + q_k)  -  (128 + z_G) * sum_k x_k )
        // where both sums are MFMAs with exact bf16 x bf16 products accumulated in fp32 (the second one against a
        // fragment of ones).  7 VALU per word + 8 per chunk (bf16 decode 777 -> 934 tokens/s).  The weights are NOT
        // individually rounded to bf16 here: the result is the exact-arithmetic value, up to 2 output ulps (1.2e-2 of
        // max|y| measured) away from the reference's rounding chain -- inside the reference's own acceptance for other
        // kernels (atol 8e-3 + rtol 0.15, tests/kernels/test_gptq.py:255,321-360) but outside this repo's default
        // gate, hence a flag and not the default.
        const uint32_t mw = st.meta[0];
        const float s = bits16_to_f32<SCL>((uint16_t)(mw & 0xffffu));
        const float zc = 128.f + (float)((mw >> 16) & 0xFu);
        const uint32_t magic = dk.magic_bf;
        const u4_t ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
        f4_t ag = {0.f, 0.f, 0.f, 0.f}, sg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = st.w[0][j];
            u4_t b;
            b.x = and_or(w, dk.lo, magic);        // k0,k1
            b.y = and_or(w >> 4, dk.lo, magic);   // k2,k3
            b.z = and_or(w >> 8, dk.lo, magic);   // k4,k5
            b.w = and_or(w >> 12, dk.lo, magic);  // k6,k7
            const u4_t av = aslot[abase + 4 * j + rq];
            ag = mfma16<ACT>(av, b, ag);
            sg = mfma16<ACT>(av, ones, sg);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0][i] = __builtin_fmaf(s, __builtin_fmaf(sg[i], -zc, ag[i]), acc[0][i]);
        return;
    }
    if constexpr (BITS == 4 && ACT == kFP16 && SCL == kFP16 && GPC == 1 && (AM == AM_ROW1 || AM == AM_ROW1P || AM == AM_ROW4)) {
        if (p.alg_fp16) {
            // decode forms 2 / 3 (round 6; block-uniform branch): GROUP-FACTORED dequant.  The code pairs become the exact small integers
            // (q - z) in fp16 -- (w & 0x000F000F) | 0x6400 = 1024 + q and (w & 0x00F000F0) | 0x5400 = 64 + q, one packed add of the pre-baked
            // -(1024 + z) | -(64 + z) each -- and go into the MFMA unscaled; the group's scale multiplies the fp32 partial sum once per chunk:
            //     y = sum_g s_g * ( sum_{k in g} x_k (q_k - z_g) ).
            // 9 VALU per packed word instead of 13 (no per-weight multiply).  Exact products, fp32 accumulation: this is the exact-arithmetic
            // value of the reference's y = x @ (s (q - z)); it differs from the reference's chain only by the reference's own per-weight
            // rounding fp16(s (q - z)) (2^-12 relative, random).  (Measured first and dropped: feeding 1024 + q / 64 + q straight into the MFMA
            // and subtracting the offsets per chunk -- 5 VALU per word, +2 % on the chain, but the fp32 accumulator then carries 1024 x sum|x|
            // and an activation row with outliers loses 0.25 absolute on outputs of a few hundred: outside the reference's element-wise atol.)
            const uint32_t mw = st.meta[0];
            const float s = bits16_to_f32<SCL>((uint16_t)(mw & 0xffffu));
            const uint32_t zc = mw >> 16;                                  // 0xE400 | z = fp16 -(1024 + z)
            const h2_t zlo = as_h2(zc | (zc << 16));
            const uint32_t zh = 0xD400u | ((zc & 0xFu) << 4);              // fp16 -(64 + z)
            const h2_t zhi = as_h2(zh | (zh << 16));
            uint32_t magic_hi = 0x54005400u;
            asm volatile("" : "+v"(magic_hi));
            f4_t g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w = st.w[0][j], w8 = w >> 8;
                u4_t b;
                b.x = as_u32(as_h2(and_or(w, dk.lo, dk.magic)) + zlo);
                b.y = as_u32(as_h2(and_or(w, dk.hi, magic_hi)) + zhi);
                b.z = as_u32(as_h2(and_or(w8, dk.lo, dk.magic)) + zlo);
                b.w = as_u32(as_h2(and_or(w8, dk.hi, magic_hi)) + zhi);
                const u4_t av = aslot[abase + 4 * j + rq];
                if (j & 1) {
                    g1 = mfma16<ACT>(av, b, g1);
                } else {
                    g0 = mfma16<ACT>(av, b, g0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[0][i] = __builtin_fmaf(s, g0[i] + g1[i], acc[0][i]);
            return;
        }
    }
    ColConst cc = expand_meta<BITS, SCL>(st.meta[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (GPC == 4) {
            if (j > 0) cc = expand_meta<BITS, SCL>(st.meta[j]);
        }
        u4_t b;
        if constexpr (BITS == 4) {
            b = dequant_word4<ACT, SCL>(st.w[0][j], cc, dk);
        } else {
            b = dequant_word8<ACT, SCL>(st.w[j >> 1][(j & 1) * 2], st.w[j >> 1][(j & 1) * 2 + 1], cc, dk);
        }
        if constexpr (AM == AM_ROW1 || AM == AM_ROW1P || AM == AM_ROW4) {
            // fragment of lane (m = c, rq) = the 16 bytes at segment 4*j + rq of row m (same-wave LDS accesses
            // execute in order, so the read needs no barrier after the write above)
            const u4_t av = aslot[abase + 4 * j + rq];
            acc[0] = mfma16<ACT>(av, b, acc[0]);
        } else if constexpr (is_rows<AM>()) {
#pragma unroll
            for (int mtile = 0; mtile < MT; ++mtile) {
                const u4_t av = aslot[(16 * mtile + c) * kRowsPitch + 4 * j + rq];
                acc[mtile] = mfma16<ACT>(av, b, acc[mtile]);
            }
        }
    }
}

// Sum over the 16 lanes of a DPP row, every lane of the row gets the total (fixed order; four v_add_f32 with a DPP operand, no LDS).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
    return v;
}
// ... and over the whole wave (the four row totals through v_readlane: wave-uniform result, no LDS)
__device__ __forceinline__ float wave64_sum(float v) {
    v = row16_sum(v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// bf16 activations on the RAW-code path (decode form 5): the f16 matrix pipe is the one that takes the codes as denormals, so a wave turns its
// (glued) bf16 x pieces into fp16 -- exactly: a bf16 value has 8 significant bits -- after dividing them by a power of two chosen from the wave's
// own largest |x| (so that nothing overflows fp16's 2^15 range whatever the input; elements more than 2^29 below the wave's largest lose bits,
// their products are below fp32's resolution of the sum anyway).  The power of two goes back in through the chunk constants.
__device__ __forceinline__ uint32_t wave_max_abs_bf16(const u4_t* g, int n) {       // largest |x| of the wave's pieces, as bf16 bits (sign cleared)
    uint32_t m = 0u;
    for (int q = 0; q < n; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a = g[q][j] & 0x7fff7fffu;
            const uint32_t hi = a >> 16, lo = a & 0xffffu;
            m = m > hi ? m : hi;
            m = m > lo ? m : lo;
        }
    }
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)m, mk, 64);
        m = m > o ? m : o;
    }
    return m;
}
__device__ __forceinline__ int bf16_down_shift(uint32_t max_abs_bits) {              // k >= 0 with |x| 2^-k < 2^15 for every element
    const int e = (int)(max_abs_bits >> 7) - 127;
    return e > 14 ? e - 14 : 0;
}
__device__ __forceinline__ uint32_t bf16pair_to_f16pair(uint32_t u, float dn) {
    const float lo = __builtin_bit_cast(float, u << 16) * dn, hi = __builtin_bit_cast(float, u & 0xffff0000u) * dn;
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lo, hi));
}

// Sum of the W (<= 16) waves' reduction rows for this lane, in wave order.  A rolled loop on purpose: hipcc emits ds_read_b32 -> s_waitcnt -> v_add per
// row (W serialized LDS round trips), but issuing all sixteen reads back to back (rows past W clamped and masked) measured SLOWER on the same box --
// o 4.23 -> 4.37, gate_up 12.85 -> 13.5 us, chain 1052 -> 1020 tokens/s (profiles/r06_decode_forms.txt): sixteen more live registers and ~40
// instructions of clamped addressing in every tile's epilogue cost more than the round trips, which overlap the other waves' work.
__device__ __forceinline__ float sum_wave_rows(const float* rows, int stride, int W, int lane) {   // stride: floats between two waves' rows
    float v = 0.f;
    for (int w = 0; w < W; ++w) v += rows[(size_t)w * stride + lane];
    return v;
}
__device__ __forceinline__ float sum_wave_rows(const float (*rows)[64], int W, int lane) { return sum_wave_rows(&rows[0][0], 64, W, lane); }

// Cross-block split-K hand-off (last arriver reduces) + the epilogue: the reference's rounding chain and the decode op's output
// glue.  `v` = this lane's fp32 sum for output (m, n) of its block's K range; `live` = the lane owns a real output.
template <int ACT>
__device__ __forceinline__ void finish_outputs(const SkinnyParams& p, float v, bool live, int m, int n, int tile, int split, int wave,
                                               int lane, uint32_t res_raw, int* s_last) {
    if (p.splits > 1) {
        // publish the fp32 partial write-through (sc1): no release fence needed
        if (live) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(p.slabs + ((size_t)split * p.M + m) * p.N + n);
            __hip_atomic_store(dst, __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_last = (old == p.splits - 1);
        }
        __syncthreads();
        if (!*s_last) return;
        // last arriver: deterministic reduction over the splits; slabs read with sc1 (L1-bypassing) loads
        if (live) {
            float s = 0.f;
            for (int sp = 0; sp < p.splits; ++sp) {
                uint32_t* src = reinterpret_cast<uint32_t*>(p.slabs + ((size_t)sp * p.M + m) * p.N + n);
                s += __builtin_bit_cast(float, __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            v = s;
        }
        if (threadIdx.x == 0) __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- epilogue: round like the reference (matmul result, then += bias in the activation dtype) ----
    if (live && p.out_f32) {
        reinterpret_cast<float*>(p.out)[(size_t)m * p.N + n] = v;
    } else if (p.out_glue == kOutSiluMul || p.stats_out != nullptr) {
        // decode op (M <= 16) epilogues that combine a tile's 16 outputs of one row: reducer wave w holds row w in lanes 0..15
        // (accumulator register w of the lanes with rq == 0) and runs them wave-uniformly so the lane shuffles are legal
        if (wave < 4 && wave < p.M) {
            // reducer wave w: lanes 0..15 hold row w, lanes 16..31 row w + 4, 32..47 row w + 8, 48..63 row w + 12 (live when < M);
            // all shuffles below stay inside a 16-lane group
            const int tiles = (p.N + kTileN - 1) / kTileN;
            const int c16 = lane & 15;
            float y = round_through<ACT>(v);
            if (p.bias != nullptr && live) y = round_through<ACT>(y + load16_as_f32<ACT>(p.bias, (size_t)n));
            if (p.out_glue == kOutSiluMul) {
                // interleaved gate|up tile (fuse_gate_up_interleaved): lanes 0..7 of a group hold gate columns j, lanes 8..15 the
                // matching up columns; HF LlamaMLP: act(silu(gate)) * up, each rounded in the activation dtype
                // (DPP row_shl:8 = lane i reads lane i + 8 of its 16-lane row: no LDS round trip on the launch's tail)
                const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x108, 0xf, 0xf, false));
                const float a = round_through<ACT>(y / (1.0f + expf(-y))) * up;
                const int j = tile * 8 + c16;
#if GPTQHIP_OUT_WRITE_THROUGH
                if (live && c16 < 8 && j < p.N / 2) __hip_atomic_store(reinterpret_cast<uint16_t*>(p.out) + (size_t)m * (p.N / 2) + j, f32_to_16<ACT>(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                if (live && c16 < 8 && j < p.N / 2) reinterpret_cast<uint16_t*>(p.out)[(size_t)m * (p.N / 2) + j] = f32_to_16<ACT>(a);
#endif
            } else {
                if (p.residual != nullptr) y = bits16_to_f32<ACT>((uint16_t)(res_raw >> ((lane & 1) * 16))) + y;
                const float h = round_through<ACT>(y);
#if GPTQHIP_OUT_WRITE_THROUGH   // dev A/B: the residual-stream outputs + statistics stored write-through (sc1) -- nothing dirty left for the end-of-kernel write-back
                if (live) __hip_atomic_store(reinterpret_cast<uint16_t*>(p.out) + (size_t)m * p.N + n, f32_to_16<ACT>(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                if (live) reinterpret_cast<uint16_t*>(p.out)[(size_t)m * p.N + n] = f32_to_16<ACT>(h);
#endif
                float sq = live ? h * h : 0.f;  // RMSNorm statistic of the NEXT op: fixed rotate tree over the 16 columns (DPP row_ror)
                sq = row16_sum(sq);
#if GPTQHIP_OUT_WRITE_THROUGH
                if (c16 == 0 && m < p.M) __hip_atomic_store(reinterpret_cast<uint32_t*>(p.stats_out) + (size_t)m * tiles + tile, __builtin_bit_cast(uint32_t, sq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                if (c16 == 0 && m < p.M) p.stats_out[(size_t)m * tiles + tile] = sq;
#endif
            }
        }
    } else if (live) {
        float y = round_through<ACT>(v);
        if (p.bias != nullptr) y = y + load16_as_f32<ACT>(p.bias, (size_t)n);
        if (p.residual != nullptr) {  // decode op (M == 1): hidden = residual + linear(x), each step rounded like torch
            y = bits16_to_f32<ACT>((uint16_t)(res_raw >> ((lane & 1) * 16))) + round_through<ACT>(y);
        }
        reinterpret_cast<uint16_t*>(p.out)[(size_t)m * p.N + n] = f32_to_16<ACT>(y);
    }
}

// (Measured and dropped in round 2: a sched_barrier after every stage's loads, which keeps hipcc from sinking the four dwordx4
// weight loads of a ring round to the end of the round -- per-shape times moved by < 1.5 % either way, other waves cover.)
// LB: threads the block may have.  17..32 rows (MT == 2) hold 32 activation registers per ring stage: under the 128-VGPR budget
// of a 1024-thread block every such instantiation spilled 24-60 registers to scratch (round-2 ISA audit); they are built with
// LB = 512 (<= 8 waves per block, 256-VGPR budget) instead and the planner picks <= 8 waves for them.  33..64 rows (MT == 4, round 3:
// one launch -- the weights are read ONCE -- instead of two 32-row launches) live in the same 512-thread budget.
template <int BITS, int ACT, int SCL, int MT, int GPC, int AM, int D, int GLUE = 0, int LB = skinny_launch_bound<BITS, MT>()>
__global__ __launch_bounds__(LB) __attribute__((amdgpu_waves_per_eu(skinny_min_waves_per_simd<BITS, ACT, MT, AM, D>())))
void skinny_kernel(SkinnyParams p) {
    // ONE dynamic LDS array (16-B aligned base, no statics in front of it): per-wave activation slots during the K
    // loop, then the split-K reduction buffer red[W][MT*4][64]; the last 16 bytes hold the "last arriver" flag.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float(*red)[MT * 4][64] = reinterpret_cast<float(*)[MT * 4][64]>(lds);
    constexpr int kSlot = slot_bytes<AM, MT>() > MT * 1024 ? slot_bytes<AM, MT>() : MT * 1024;  // bytes per wave
    int* s_last = reinterpret_cast<int*>(reinterpret_cast<char*>(lds) + (blockDim.x >> 6) * kSlot);
    uint16_t* xbuf = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(lds) + (blockDim.x >> 6) * kSlot + 16);  // AM_ROW1P

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform -> scalar branches
    const int W = blockDim.x >> 6;
    const int c = lane & 15;
    const int rq = lane >> 4;
    const int tile = blockIdx.x;
    const int split = blockIdx.y;

    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.chunks, c_begin + p.chunks_per_split);

    f4_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f4_t{0.f, 0.f, 0.f, 0.f};

    TileBases tb;
    tb.w = reinterpret_cast<const char*>(p.qw) + (size_t)tile * p.chunks * (BITS == 4 ? 1024 : 2048);
    tb.meta = p.meta + (size_t)tile * p.G * 16;
    tb.x = reinterpret_cast<const char*>(p.x);
    tb.lane16 = (uint32_t)lane * 16u;
    tb.c4 = (uint32_t)c * 4u;
    if constexpr (MT >= 2) {
        tb.xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.M * p.K * 2, 0x00020000);
        tb.quad_stride = (uint32_t)p.K * 8u;
    }
    u4_t* aslot = reinterpret_cast<u4_t*>(reinterpret_cast<char*>(lds) + wave * slot_bytes<AM, MT>());
    const DequantConsts dk = make_dequant_consts<BITS>();

    // decode op: the residual of this tile's row-0 outputs (reducer lanes = wave 0, lanes 0..15) is requested up front as
    // the aligned 32-bit pair holding the column (no zero-extension ALU op behind the load -> no early wait), used in the
    // epilogue; glue_inv = RMSNorm's rsqrt(mean(h^2) + eps)
    GlueInv ginv;                            // (v[1..3]: rows 4.. of the 5..16-row decode op)
    float& glue_inv = ginv.v[0];
    uint32_t res_raw = 0u;
    if (p.residual != nullptr && wave < 4 && wave + 4 * rq < p.M) {
        // (decode op: M <= 16; reducer wave w holds output rows w (lanes 0..15), w + 4 (lanes 16..31), w + 8, w + 12)
        const int coln = tile * kTileN + c;
        res_raw = reinterpret_cast<const uint32_t*>(p.residual)[((size_t)(wave + 4 * rq) * p.N + (coln < p.N ? coln : 0)) >> 1];
    }

    // D-deep register ring: every load of a chunk (weights, constants, activations) is issued D chunks ahead,
    // so a wave keeps D KiB of HBM reads in flight and waits only for the oldest stage.
    Stage<BITS, GPC, MT, AM> st[D];
    const int n_mine = p.regular ? p.n_mine : (c_begin + wave < c_end ? (c_end - c_begin - wave + W - 1) / W : 0);  // chunks of this wave
    if (p.regular) {
        // REGULAR: every wave owns a multiple of D chunks (the planner picks W for that), K % 128 == 0 and one
        // group constant per chunk.  Straight-line prologue / steady loop / drain with unconditional loads, so the
        // compiler's s_waitcnt insertion can COUNT (vmcnt(3*(D-1)) style) instead of draining the queue -- with
        // conditional loads it falls back to vmcnt(0) before every stage, which serialises the ring.
        {
            LaneOffs<MT, AM> lo;
            if constexpr (AM == AM_ROW1) {
                lo.x[0] = (uint32_t)lane * 4u;
            } else if constexpr (AM == AM_ROW1P) {
                lo.x[0] = (uint32_t)lane * 8u;  // this lane's (perm[k], perm[k+1]) pair inside a chunk's 512 bytes
            } else if constexpr (AM == AM_ROW4) {
                const int row = rq < p.M ? rq : 0;
                lo.x[0] = (uint32_t)row * (uint32_t)p.K * 2u + (uint32_t)c * 16u;
            } else if constexpr (is_rows<AM>() && MT >= 2) {
                lo.x[0] = (uint32_t)rq * (uint32_t)p.K * 2u + (uint32_t)c * 16u;   // (row quad i: + i * quad_stride through the scalar offset)
            } else if constexpr (is_rows<AM>()) {
#pragma unroll
                for (int i = 0; i < row_quads<AM, MT>(); ++i) {
                    int row = 4 * i + rq;
                    row = row < p.M ? row : 0;
                    lo.x[i] = (uint32_t)row * (uint32_t)p.K * 2u + (uint32_t)c * 16u;
                }
            }
            int cur = c_begin + wave;
            Cursor cu;
            cu.w = tb.w + (size_t)cur * (BITS == 4 ? 1024 : 2048);
            cu.x = tb.x + (size_t)cur * 256;
            cu.chunk = cur;
            cu.end = c_end;
            if constexpr (AM == AM_ROW1P) {
                // the x row goes to LDS once per block: its (L2-hit) loads are issued BEFORE the weight ring so that
                // waiting for them does not wait for the HBM stream behind them (vmcnt retires in issue order).  Decode op:
                // the input glue (RMSNorm / SiLU*mul) is applied HERE, once per element, on the way into LDS -- the ring's
                // gather then reads finished activations.
                const u4_t* xs = reinterpret_cast<const u4_t*>(p.x);
                const u4_t* gs = GLUE == kGlueRmsNorm ? reinterpret_cast<const u4_t*>(p.glue_b) : xs + p.K / 8;  // norm weight | up half
                const int n16 = p.K / 8;
                float* scratch = reinterpret_cast<float*>(xbuf + p.K);   // 1 + 16 floats behind the row (see the launch)
                float sv[8];
                if constexpr (GLUE == kGlueRmsNorm) {
                    if (p.stats_in != nullptr && wave == 0) {   // the producer's per-tile sums of squares, in front of everything
                        const int last = p.stats_n - 1;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int idx = lane + 64 * i;
                            sv[i] = p.stats_in[idx < last ? idx : last];
                        }
                    }
                }
                u4_t xr[2], gr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = (int)threadIdx.x + i * (int)blockDim.x;
                    xr[i] = xs[idx < n16 ? idx : 0];
                    if constexpr (GLUE != kGlueNone) gr[i] = gs[idx < n16 ? idx : 0];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < D; ++d) load_stage_fast<BITS, GPC, MT, AM>(st[d], p, tb, lo, cu, W);
                if constexpr (GLUE == kGlueRmsNorm) {
                    if (p.stats_in != nullptr) {
                        if (wave == 0) {
                            float ssum = 0.f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) ssum += (lane + 64 * i < p.stats_n) ? sv[i] : 0.f;
#pragma unroll
                            for (int mk = 32; mk >= 1; mk >>= 1) ssum += __shfl_xor(ssum, mk, 64);
                            if (lane == 0) scratch[0] = rsqrtf(ssum / (float)p.K + p.eps);
                        }
                    } else {   // no producer statistics: reduce the row in the block (fixed order)
                        float ss = 0.f;
                        auto sq = [&](const u4_t& h) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float a = bits16_to_f32<ACT>((uint16_t)(h[j] & 0xffffu)), b = bits16_to_f32<ACT>((uint16_t)(h[j] >> 16));
                                ss = __builtin_fmaf(a, a, ss);
                                ss = __builtin_fmaf(b, b, ss);
                            }
                        };
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if ((int)threadIdx.x + i * (int)blockDim.x < n16) sq(xr[i]);
                        }
                        for (int idx = (int)threadIdx.x + 2 * (int)blockDim.x; idx < n16; idx += (int)blockDim.x) sq(xs[idx]);
#pragma unroll
                        for (int mk = 32; mk >= 1; mk >>= 1) ss += __shfl_xor(ss, mk, 64);
                        if (lane == 0) scratch[1 + wave] = ss;
                        __syncthreads();
                        if (threadIdx.x == 0) {
                            float tot = 0.f;
                            for (int w = 0; w < W; ++w) tot += scratch[1 + w];
                            scratch[0] = rsqrtf(tot / (float)p.K + p.eps);
                        }
                    }
                    __syncthreads();
                    glue_inv = scratch[0];
                }
                auto glued = [&](const u4_t& xv, const u4_t& gv) {
                    if constexpr (GLUE == kGlueNone) {
                        return xv;
                    } else {
                        u4_t r;
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[j] = glue_pair<ACT>(xv[j], gv[j], glue_inv, GLUE);
                        return r;
                    }
                };
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = (int)threadIdx.x + i * (int)blockDim.x;
                    if (idx < n16) reinterpret_cast<u4_t*>(xbuf)[idx] = glued(xr[i], gr[i]);
                }
                for (int idx = (int)threadIdx.x + 2 * (int)blockDim.x; idx < n16; idx += (int)blockDim.x) {
                    // (rows longer than 32 B x threads: rare, plain copy)
                    u4_t gv = {0u, 0u, 0u, 0u};
                    if constexpr (GLUE != kGlueNone) gv = gs[idx];
                    reinterpret_cast<u4_t*>(xbuf)[idx] = glued(xs[idx], gv);
                }
                __syncthreads();
            } else if constexpr ((AM == AM_ROW4 || (is_rows<AM>() && MT == 1)) && GLUE == kGlueRmsNorm) {
                // decode op on up to sixteen rows: wave w (< 4) owns the RMSNorm statistics of rows w, w + 4, w + 8, w + 12 -- the
                // producer's per-tile sums of squares (eight clamped loads per lane and row, in front of the weight ring), or a
                // wave-local reduction of the row when there is no producer (first op of a step) -- and shares 1/rms through LDS
                float* scratch = reinterpret_cast<float*>(xbuf);
                constexpr int RPW = AM == AM_ROW4 ? 1 : (AM == AM_ROWSH ? 2 : 4);   // rows per wave
                float sv[RPW][8];
                if (p.stats_in != nullptr) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int row = wave + 4 * r;
                        if (wave < 4 && row < p.M) {
                            const float* srow = p.stats_in + (size_t)row * p.stats_n;
                            const int last = p.stats_n - 1;
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int idx = lane + 64 * i;
                                sv[r][i] = srow[idx < last ? idx : last];
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int d = 0; d < D; ++d) load_stage_fast<BITS, GPC, MT, AM, GLUE>(st[d], p, tb, lo, cu, W);
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int row = wave + 4 * r;
                        if (wave < 4 && row < p.M) {
                            float ssum = 0.f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) ssum += (lane + 64 * i < p.stats_n) ? sv[r][i] : 0.f;
#pragma unroll
                            for (int mk = 32; mk >= 1; mk >>= 1) ssum += __shfl_xor(ssum, mk, 64);
                            if (lane == 0) scratch[row] = rsqrtf(ssum / (float)p.K + p.eps);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int row = wave + 4 * r;
                        if (wave < 4 && row < p.M) {
                            const u4_t* hs = reinterpret_cast<const u4_t*>(p.x) + (size_t)row * (p.K / 8);
                            float ss = 0.f;
                            for (int idx = lane; idx < p.K / 8; idx += 64) {
                                const u4_t h = hs[idx];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const float a = bits16_to_f32<ACT>((uint16_t)(h[j] & 0xffffu)), b = bits16_to_f32<ACT>((uint16_t)(h[j] >> 16));
                                    ss = __builtin_fmaf(a, a, ss);
                                    ss = __builtin_fmaf(b, b, ss);
                                }
                            }
#pragma unroll
                            for (int mk = 32; mk >= 1; mk >>= 1) ss += __shfl_xor(ss, mk, 64);
                            if (lane == 0) scratch[row] = rsqrtf(ss / (float)p.K + p.eps);
                        }
                    }
#pragma unroll
                    for (int d = 0; d < D; ++d) load_stage_fast<BITS, GPC, MT, AM, GLUE>(st[d], p, tb, lo, cu, W);
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < RPW; ++r) ginv.v[r] = scratch[4 * r + rq < p.M ? 4 * r + rq : 0];
            } else if (GLUE == kGlueRmsNorm && p.stats_in != nullptr) {
                // RMSNorm statistics handed over by the op that produced h (one partial per 16-column tile: its epilogue's
                // sum of out^2).  ONE wave per block sums them in a fixed order -- eight clamped loads per lane issued in
                // front of the weight ring, a shuffle tree -- and shares 1/rms through LDS.  (Every wave loading them itself
                // doubled the memory instructions of the 70B gate_up: 16 waves x 3584 blocks x 8 loads of the same 2 KiB.)
                float sv[8];
                if (wave == 0) {
                    const int last = p.stats_n - 1;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int idx = lane + 64 * i;
                        sv[i] = p.stats_in[idx < last ? idx : last];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // keep them in FRONT of the ring (hipcc sank one behind it -> vmcnt(0))
#pragma unroll
                for (int d = 0; d < D; ++d) load_stage_fast<BITS, GPC, MT, AM, GLUE>(st[d], p, tb, lo, cu, W);
                float* scratch = reinterpret_cast<float*>(xbuf);
                if (wave == 0) {
                    float ssum = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) ssum += (lane + 64 * i < p.stats_n) ? sv[i] : 0.f;
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) ssum += __shfl_xor(ssum, m, 64);
                    if (lane == 0) scratch[0] = rsqrtf(ssum / (float)p.K + p.eps);
                }
                __syncthreads();
                glue_inv = scratch[0];
            } else if constexpr (GLUE == kGlueRmsNorm) {
                // (no producer statistics: e.g. the first op of a step, whose input comes from outside the chain)
                // RMSNorm statistics of the whole input row, once per block: the row (L2-resident) is requested BEFORE the
                // weight ring so that waiting for it does not wait for HBM (vmcnt retires in issue order); the reduction
                // runs while the ring's first loads are in flight.  Fixed summation order: deterministic.
                const u4_t* hs = reinterpret_cast<const u4_t*>(p.x);
                const int n16 = p.K / 8;
                u4_t hr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = (int)threadIdx.x + i * (int)blockDim.x;
                    hr[i] = hs[idx < n16 ? idx : 0];
                }
#pragma unroll
                for (int d = 0; d < D; ++d) load_stage_fast<BITS, GPC, MT, AM, GLUE>(st[d], p, tb, lo, cu, W);
                float ss = 0.f;
                auto sq = [&](const u4_t& h) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = bits16_to_f32<ACT>((uint16_t)(h[j] & 0xffffu)), b = bits16_to_f32<ACT>((uint16_t)(h[j] >> 16));
                        ss = __builtin_fmaf(a, a, ss);
                        ss = __builtin_fmaf(b, b, ss);
                    }
                };
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if ((int)threadIdx.x + i * (int)blockDim.x < n16) sq(hr[i]);
                }
                for (int idx = (int)threadIdx.x + 2 * (int)blockDim.x; idx < n16; idx += (int)blockDim.x) sq(hs[idx]);
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
                float* scratch = reinterpret_cast<float*>(xbuf);  // (no act-order staging buffer in this variant)
                if (lane == 0) scratch[wave] = ss;
                __syncthreads();
                float tot = 0.f;
                for (int w = 0; w < W; ++w) tot += scratch[w];
                glue_inv = rsqrtf(tot / (float)p.K + p.eps);
            } else if constexpr (kSplitRing<MT, AM, GLUE>()) {
                // 17..64 rows: the weight ring stays D deep, the ACTIVATION stage (32 / 64 registers) is ONE deep -- st[0].x only: as
                // soon as a chunk's activations are parked in LDS the next chunk's (L2-resident) rows are requested into the same
                // registers and arrive under this chunk's dequant + MFMAs.  D activation stages needed 152-251 VGPRs at 32 rows
                // (<= 8 waves per block) and spilled at 64.
                Cursor ca = cu;    // (the first chunk's activations are requested FIRST: they are waited for first)
                load_stage_fast<BITS, GPC, MT, AM, GLUE, false, true>(st[0], p, tb, lo, ca, W);
#pragma unroll
                for (int d = 0; d < D; ++d) load_stage_fast<BITS, GPC, MT, AM, GLUE, true, false>(st[d], p, tb, lo, cu, W);
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) load_stage_fast<BITS, GPC, MT, AM, GLUE>(st[d], p, tb, lo, cu, W);
            }
            if constexpr (kSplitRing<MT, AM, GLUE>()) {
                Cursor ca;     // the activation cursor runs one chunk ahead of the chunk being multiplied
                ca.w = nullptr;
                ca.x = tb.x + (size_t)(cur + W) * 256;
                ca.chunk = cur + W;
                ca.end = c_end;
                for (int it = D; it < n_mine; it += D) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        compute_stage<BITS, ACT, SCL, MT, GPC, AM, GLUE, 1>(st[0], p, cur, lane, aslot, dk, acc, xbuf, ginv);
                        load_stage_fast<BITS, GPC, MT, AM, GLUE, false, true>(st[0], p, tb, lo, ca, W);
                        compute_stage<BITS, ACT, SCL, MT, GPC, AM, GLUE, 2>(st[d], p, cur, lane, aslot, dk, acc, xbuf, ginv);
                        load_stage_fast<BITS, GPC, MT, AM, GLUE, true, false>(st[d], p, tb, lo, cu, W);
                        cur += W;
                    }
                }
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    // (only the last ring round can hold padding chunks: wave-uniform skip; the activation prefetch past the end is
                    // clamped to the last real chunk like the weight loads)
                    if (cur < c_end) {
                        compute_stage<BITS, ACT, SCL, MT, GPC, AM, GLUE, 1>(st[0], p, cur, lane, aslot, dk, acc, xbuf, ginv);
                        if (d + 1 < D) load_stage_fast<BITS, GPC, MT, AM, GLUE, false, true>(st[0], p, tb, lo, ca, W);
                        compute_stage<BITS, ACT, SCL, MT, GPC, AM, GLUE, 2>(st[d], p, cur, lane, aslot, dk, acc, xbuf, ginv);
                    }
                    cur += W;
                }
            } else {
            for (int it = D; it < n_mine; it += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    compute_stage<BITS, ACT, SCL, MT, GPC, AM, GLUE>(st[d], p, cur, lane, aslot, dk, acc, xbuf, ginv);
                    load_stage_fast<BITS, GPC, MT, AM, GLUE>(st[d], p, tb, lo, cu, W);
                    cur += W;
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                // (only the last ring round can hold padding chunks: wave-uniform skip)
                if (cur < c_end) compute_stage<BITS, ACT, SCL, MT, GPC, AM, GLUE>(st[d], p, cur, lane, aslot, dk, acc, xbuf, ginv);
                cur += W;
            }
            }
        }
    } else {
        // generic: any chunk count per wave (ragged K, forced geometry); conservative waits
        if constexpr (AM == AM_ROW1P) {
            for (int idx = (int)threadIdx.x; idx < p.K / 8; idx += (int)blockDim.x)
                reinterpret_cast<u4_t*>(xbuf)[idx] = reinterpret_cast<const u4_t*>(p.x)[idx];
            __syncthreads();
        }
        if constexpr (kSplitRing<MT, AM, GLUE>()) {
            // 17..64 rows off the regular pipeline (ragged K, forced geometry -- rare): one stage at a time, no prefetch ring, so
            // that this fallback does not dictate the instantiation's register budget (D activation stages = 64-128 VGPRs)
            for (int cur = c_begin + wave; cur < c_end; cur += W) {
                load_stage<BITS, GPC, MT, AM>(st[0], p, tb, cur, lane);
                compute_stage<BITS, ACT, SCL, MT, GPC, AM>(st[0], p, cur, lane, aslot, dk, acc, xbuf);
            }
        } else {
        {
            int nxt = c_begin + wave;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (nxt < c_end) load_stage<BITS, GPC, MT, AM>(st[d], p, tb, nxt, lane);
                nxt += W;
            }
        }
        for (int cur = c_begin + wave; cur < c_end;) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (cur < c_end) {
                    compute_stage<BITS, ACT, SCL, MT, GPC, AM>(st[d], p, cur, lane, aslot, dk, acc, xbuf);
                    const int nxt = cur + D * W;
                    if (nxt < c_end) load_stage<BITS, GPC, MT, AM>(st[d], p, tb, nxt, lane);
                    cur += W;
                }
            }
        }
        }
    }

#if GPTQHIP_ABLATE & 8
    if (__builtin_bit_cast(uint32_t, acc[0][0]) == 0x12345678u) reinterpret_cast<uint16_t*>(p.out)[blockIdx.x] = 1;
    return;
#endif
    // ---- in-block split-K reduction through LDS -------------------------------------------------
    __syncthreads();  // the activation slots alias the reduction buffer
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][mt * 4 + i][lane] = acc[mt][i];
    }
    __syncthreads();

    if constexpr (MT == 4) {
        // 33..64 rows: 16 accumulator registers per lane, at most 8 waves -> reducer wave w owns registers w, w + W, ...; no decode
        // op glue and no cross-block split-K here (the planner keeps splits == 1), just the reference's rounding chain
        const int n4 = tile * kTileN + c;
        for (int r = wave; r < 4 * MT; r += W) {
            float v4 = 0.f;
            for (int w = 0; w < W; ++w) v4 += red[w][r][lane];
            const int m4 = 16 * (r >> 2) + 4 * rq + (r & 3);
            if (m4 < p.M && n4 < p.N) {
                if (p.out_f32) {
                    reinterpret_cast<float*>(p.out)[(size_t)m4 * p.N + n4] = v4;
                } else {
                    float y = round_through<ACT>(v4);
                    if (p.bias != nullptr) y = y + load16_as_f32<ACT>(p.bias, (size_t)n4);
                    reinterpret_cast<uint16_t*>(p.out)[(size_t)m4 * p.N + n4] = f32_to_16<ACT>(y);
                }
            }
        }
        return;
    }
    // wave w < 4*MT owns accumulator register (mt = w>>2, i = w&3): row m = 16*mt + 4*rq + i, column n
    const bool reducer = wave < 4 * MT;
    const int m = 16 * (wave >> 2) + 4 * rq + (wave & 3);
    const int n = tile * kTileN + c;
    const bool live = reducer && m < p.M && n < p.N;
    float v = 0.f;
    if (reducer) {
        v = sum_wave_rows(&red[0][wave][0], MT * 4 * 64, W, lane);
    }

    finish_outputs<ACT>(p, v, live, m, n, tile, split, wave, lane, res_raw, s_last);
}




// ------------------------------------------------------------------------------------------------
// Batch 1, "preload" form (round 6; decode forms 2 / 3 of gptqhip_set_decode_form).  Same mapping as skinny_kernel's M = 1 regular pipeline
// (one 16-column tile per block, W waves split the K range chunk by chunk, D-deep register ring of nt weight loads, in-block LDS
// reduction, finish_outputs epilogue) with the two changes the round-6 counters asked for (profiles/r06_pmc_summary.json: the M = 1
// kernel issues 82 VALU and 3 VMEM instructions per 1 KiB chunk; VALU busy 55 % of the launch on gate_up):
//   * ONE VMEM instruction per chunk.  skinny_kernel's ring stage carries a 4-byte x load and a 4-byte constant load beside the 1 KiB
//     weight load (round-2 ablation: those two cost 2.6 us of gate_up's 15).  Here a wave fetches the x pieces (16 B per lane: four chunks
//     per instruction), the norm-weight pieces and the group constants of ALL its chunks in front of the weight ring (<= 16 chunks per
//     wave: at most 12 instructions, once), applies the input glue once, and parks them in its private LDS slot; the ring then holds
//     weights only and a chunk costs four ds_read_b128 + one ds_read_b32.
//   * ALG = 1 (fp16 activations): group-factored dequant (see compute_stage's decode forms 2 / 3): exact (q - z) pairs into the MFMA,
//     9 VALU per packed word, the scale once per chunk in fp32.  ALG = 0 keeps the bit-faithful per-weight rounding.
// Replaces nothing upstream beyond what skinny_kernel does (TorchLinear._forward_eager, torch.py:326-347).
// ------------------------------------------------------------------------------------------------
// PERM: act-order checkpoints (the permutation applied inside the kernel, like skinny_kernel's AM_ROW1P): the block stages the GLUED x row in
// LDS once, every wave gathers the eight elements per lane of each of its chunks from there at park time (the eight indices = two 16-byte
// loads per lane and quad, in front of the ring) -- the main loop is unchanged.
#ifndef GPTQHIP_SK1_RING_FIRST
#define GPTQHIP_SK1_RING_FIRST 0
#endif
#ifndef GPTQHIP_SK1_ABLATE   // dev timing builds (tests/dev/sk1_ablate_build.sh; WRONG results): 1 no small-operand loads, 2 no dequant / MFMA,
#define GPTQHIP_SK1_ABLATE 0 // 4 no reduction / epilogue, 8 no weight loads, 16 no statistics load, 32 no norm-weight load
#endif
template <int ACT, int SCL, int D, int GLUE, int ALG, bool PERM = false>
__global__ __launch_bounds__(1024) void skinny1_kernel(SkinnyParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int c = lane & 15, rq = lane >> 4;
    const int tile = blockIdx.x, split = blockIdx.y;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.chunks, c_begin + p.chunks_per_split);
    const int n_mine = p.n_mine;                       // chunks per wave incl. the padding of the last ring round
    const int nq = (n_mine + 3) >> 2;                  // 16-byte x instructions (four chunks each)
    // ALG = 2 ("raw codes", decode form 5): the 4-bit codes go into the MFMA as they are -- (w & 0x000F000F) is a pair of fp16 DENORMALS q * 2^-24,
    // (w & 0x00F000F0) a pair q * 2^-20 (the matrix pipe takes fp16 denormals at face value: tests/dev/mfma_denorm_probe.hip) -- one v_and_b32 per
    // two weights, no zero-point subtraction: y = sum_g s_g (2^24 lo_g + 2^20 hi_g - z_g Sx_g) with lo / hi the two MFMA chains of a chunk and
    // Sx_g the sum of the chunk's 128 (glued) activations, taken once per wave at park time.  The two classes need their own A fragments:
    // the x pieces are parked re-paired ([k0 k1 | k4 k5] of two K-steps, [k2 k3 | k6 k7] of two K-steps) so a fragment is still one ds_read_b128.
    constexpr bool RAW = ALG == 2 && ((ACT == kFP16 && SCL == kFP16) || ACT == kBF16);
    constexpr bool XCVT = RAW && ACT == kBF16;      // bf16 activations ride the f16 matrix pipe (see bf16pair_to_f16pair)
    char* const slot = reinterpret_cast<char*>(lds) + wave * p.slot_stride;
    u4_t* const xs = reinterpret_cast<u4_t*>(slot);                                   // [4 nq][16] u4: the glued x pieces
    uint32_t* const ms = reinterpret_cast<uint32_t*>(slot + nq * 1024);               // [4 nq][16] meta words (RAW: [4 nq][16] float2 = s 2^24, s 2^20)
    int* s_last = reinterpret_cast<int*>(reinterpret_cast<char*>(lds) + W * p.slot_stride);
    float* scratch = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + W * p.slot_stride + 16);
    float(*red)[64] = reinterpret_cast<float(*)[64]>(reinterpret_cast<char*>(lds) + W * p.slot_stride + 96);   // beside the slots: one barrier
    const DequantConsts dk = make_dequant_consts<4>();
    const char* wbase = reinterpret_cast<const char*>(p.qw) + (size_t)tile * p.chunks * 1024;
    const char* mbase = reinterpret_cast<const char*>(p.meta + (size_t)tile * p.G * 16);
    const uint32_t lane16 = (uint32_t)lane * 16u, c4 = (uint32_t)c * 4u;

    // weight ring
    u4_t st[D];
    int nxt = c_begin + wave;
    auto load_w = [&](u4_t& dst) __attribute__((always_inline)) {
        const int ck = nxt < c_end ? nxt : c_end - 1;
#if GPTQHIP_SK1_ABLATE & 8
        dst = u4_t{(uint32_t)ck, lane16, (uint32_t)ck ^ lane16, (uint32_t)ck + lane16};
#else
        dst = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(wbase + (size_t)ck * 1024 + lane16));
#endif
        nxt += W;
    };
#if GPTQHIP_SK1_RING_FIRST   // dev A/B: the ring's first round requested BEFORE the small operands (they then retire behind it): SLOWER -- o 4.22 -> 4.52, qkv 5.47 -> 5.97,
                             // down 7.74 -> 8.0 us (the park no longer hides under the ring's first round trip); profiles/r06_decode_forms.txt
#pragma unroll
    for (int d = 0; d < D; ++d) load_w(st[d]);
    __builtin_amdgcn_sched_barrier(0);
#endif
    uint32_t res_raw = 0u;
    if (p.residual != nullptr && wave == 0 && rq == 0 && !(GPTQHIP_SK1_ABLATE & 1)) {
        const int coln = tile * kTileN + c;
        res_raw = reinterpret_cast<const uint32_t*>(p.residual)[(size_t)(coln < p.N ? coln : 0) >> 1];
    }
    f4_t sv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if constexpr (GLUE == kGlueRmsNorm) {
        if (p.stats_in != nullptr && !(GPTQHIP_SK1_ABLATE & 17)) {
            // the producer's per-tile sums of squares, in front of everything: ONE 16-byte load per lane covers 256 partial sums (a second
            // one up to 512); entries past stats_n are outside the descriptor and read as zeros.  EVERY wave fetches and reduces them (one
            // L2-hit instruction per wave, DPP adds): no LDS hand-over and no block barrier between the launch and its first multiply.
            // (skinny_kernel lets one wave do it: with its eight 4-byte loads per lane, per-wave copies doubled the 70B gate_up's VMEM count.)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.stats_in), 0, p.stats_n * 4, 0x00020000);
            sv[0] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, 0, 0));
            if (p.stats_n > 256) sv[1] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, 1024, 0));
        }
    }
    // this wave's x pieces / norm-weight pieces / group constants: lane (rq, c) of instruction q serves the wave's chunk 4 q + rq
    u4_t xq[4], gq[4];
    uint32_t mq[4];
    u4_t pq[PERM ? 4 : 1][2];          // PERM: perm[k'] of the lane's eight rows of each chunk
    u4_t xr[2], gr[2];                 // PERM: this thread's 16-byte pieces of the x row / norm weight (staged once per block)
    uint16_t* const xbuf = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(lds) + W * p.slot_stride + 96 + W * 256);
    const int n16 = p.K / 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < nq) {
            int ck = c_begin + wave + (4 * q + rq) * W;
            ck = ck < c_end ? ck : c_end - 1;           // padding chunks: any finite values (their stages are skipped)
#if GPTQHIP_SK1_ABLATE & 1
            xq[q] = u4_t{0x3c003c00u + (uint32_t)ck, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
            gq[q] = xq[q];
            mq[q] = 0xE4082000u + (uint32_t)ck;
            if constexpr (PERM) { pq[q][0] = xq[q]; pq[q][1] = xq[q]; }
            continue;
#endif
            if constexpr (PERM) {
                const u4_t* pp = reinterpret_cast<const u4_t*>(p.perm + (size_t)ck * 128 + c * 8);
                pq[q][0] = pp[0];
                pq[q][1] = pp[1];
            } else {
                xq[q] = *reinterpret_cast<const u4_t*>(reinterpret_cast<const char*>(p.x) + (size_t)ck * 256 + c * 16);
#if GPTQHIP_SK1_ABLATE & 32
                gq[q] = u4_t{0x3c003c00u + (uint32_t)ck, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#else
                if constexpr (GLUE == kGlueRmsNorm) gq[q] = *reinterpret_cast<const u4_t*>(reinterpret_cast<const char*>(p.glue_b) + (size_t)ck * 256 + c * 16);
#endif
            }
            mq[q] = *reinterpret_cast<const uint32_t*>(mbase + ((size_t)(ck >> p.cpg_shift) << 6) + c4);
        }
    }
    if constexpr (PERM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = (int)threadIdx.x + i * (int)blockDim.x;
            xr[i] = reinterpret_cast<const u4_t*>(p.x)[idx < n16 ? idx : 0];
            if constexpr (GLUE == kGlueRmsNorm) gr[i] = reinterpret_cast<const u4_t*>(p.glue_b)[idx < n16 ? idx : 0];
        }
    }
#if !GPTQHIP_SK1_RING_FIRST
    __builtin_amdgcn_sched_barrier(0);   // keep them in FRONT of the ring
#pragma unroll
    for (int d = 0; d < D; ++d) load_w(st[d]);
#endif

    float inv = 1.f;
    if constexpr (GLUE == kGlueRmsNorm) {
        if (p.stats_in != nullptr) {
            const float ssum = wave64_sum(((sv[0][0] + sv[0][1]) + (sv[0][2] + sv[0][3])) + ((sv[1][0] + sv[1][1]) + (sv[1][2] + sv[1][3])));   // fixed order
            inv = rsqrtf(ssum / (float)p.K + p.eps);
        } else {
            // no producer statistics (first op of a step; the planner keeps splits == 1 here): the waves' pieces cover the row exactly once
            float ss = 0.f;
            auto sq4 = [&](const u4_t& h) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = bits16_to_f32<ACT>((uint16_t)(h[j] & 0xffffu)), b = bits16_to_f32<ACT>((uint16_t)(h[j] >> 16));
                    ss = __builtin_fmaf(a, a, ss);
                    ss = __builtin_fmaf(b, b, ss);
                }
            };
            if constexpr (PERM) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if ((int)threadIdx.x + i * (int)blockDim.x < n16) sq4(xr[i]);
                for (int idx = (int)threadIdx.x + 2 * (int)blockDim.x; idx < n16; idx += (int)blockDim.x) sq4(reinterpret_cast<const u4_t*>(p.x)[idx]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < nq && c_begin + wave + (4 * q + rq) * W < c_end) sq4(xq[q]);
            }
#pragma unroll
            for (int mk = 32; mk >= 1; mk >>= 1) ss += __shfl_xor(ss, mk, 64);
            if (lane == 0) scratch[1 + wave] = ss;
            __syncthreads();
            if (threadIdx.x == 0) {
                float tot = 0.f;
                for (int w = 0; w < W; ++w) tot += scratch[1 + w];
                scratch[0] = rsqrtf(tot / (float)p.K + p.eps);
            }
            __syncthreads();
            inv = scratch[0];
        }
    }
    float tsum = 0.f;    // RAW: -sum over this lane's park chunks of s z Sx (column c of chunk 4 q + rq)
    float up = 1.f;      // XCVT: 2^k, the power of two taken out of this wave's x pieces
    u4_t gg[4];          // the wave's glued x pieces (parked below; XCVT converts them in between)
    auto convert_pieces = [&]() __attribute__((always_inline)) {
        if constexpr (XCVT) {
            const int k = bf16_down_shift(wave_max_abs_bf16(gg, nq));
            const float dn = __builtin_bit_cast(float, (uint32_t)(127 - k) << 23);
            up = __builtin_bit_cast(float, (uint32_t)(127 + k) << 23);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nq) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) gg[q][j] = bf16pair_to_f16pair(gg[q][j], dn);
                }
            }
        }
    };
    auto park = [&](int q, const u4_t& g) __attribute__((always_inline)) {
        if constexpr (RAW) {
            const int li = 4 * q + rq, j = c >> 2;        // this lane holds piece c = K-step j, k-group c & 3 of chunk li
            char* base = reinterpret_cast<char*>(xs) + li * 256 + (c & 3) * 64 + (j >> 1) * 32 + (j & 1) * 8;
            *reinterpret_cast<u2_t*>(base) = u2_t{g[0], g[2]};          // (k0 k1 | k4 k5): the low-nibble class
            *reinterpret_cast<u2_t*>(base + 16) = u2_t{g[1], g[3]};     // (k2 k3 | k6 k7): the high-nibble class
            const h2_t ones = as_h2(0x3C003C00u);
            float s8 = __builtin_amdgcn_fdot2(as_h2(g[0]), ones, 0.f, false);
            s8 = __builtin_amdgcn_fdot2(as_h2(g[1]), ones, s8, false);
            s8 = __builtin_amdgcn_fdot2(as_h2(g[2]), ones, s8, false);
            s8 = __builtin_amdgcn_fdot2(as_h2(g[3]), ones, s8, false);
            s8 = row16_sum(s8);                                          // Sx of chunk li (the row's 16 lanes hold its 16 pieces)
            const uint32_t mw = mq[q];
            const float sc = bits16_to_f32<SCL>((uint16_t)(mw & 0xffffu));
            const float z = (float)((mw >> 16) & 0xFu);                  // meta = scale16 | (0xE400 | zero) << 16
            reinterpret_cast<float2*>(ms)[q * 64 + lane] = float2{sc * 16777216.f * up, sc * 1048576.f * up};
            if (c_begin + wave + li * W < c_end) tsum = __builtin_fmaf(-(sc * z), s8, tsum);      // (padding chunks: skipped like their stages)
        } else {
            xs[q * 64 + lane] = g;
            ms[q * 64 + lane] = mq[q];
        }
    };
    if constexpr (PERM) {
        // the glued row into LDS once per block (natural order), then every lane gathers its rows of each chunk by their indices
        auto glued = [&](const u4_t& xv, const u4_t& gv) __attribute__((always_inline)) {
            u4_t r = xv;
            if constexpr (GLUE == kGlueRmsNorm) {
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = glue_pair<ACT>(xv[j], gv[j], inv, GLUE);
            }
            return r;
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = (int)threadIdx.x + i * (int)blockDim.x;
            if (idx < n16) reinterpret_cast<u4_t*>(xbuf)[idx] = glued(xr[i], gr[i]);
        }
        for (int idx = (int)threadIdx.x + 2 * (int)blockDim.x; idx < n16; idx += (int)blockDim.x) {   // (rows longer than 32 B x threads: rare)
            u4_t gv = {0u, 0u, 0u, 0u};
            if constexpr (GLUE == kGlueRmsNorm) gv = reinterpret_cast<const u4_t*>(p.glue_b)[idx];
            reinterpret_cast<u4_t*>(xbuf)[idx] = glued(reinterpret_cast<const u4_t*>(p.x)[idx], gv);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < nq) {
                u4_t g;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t i0 = pq[q][j >> 1][(j & 1) * 2], i1 = pq[q][j >> 1][(j & 1) * 2 + 1];
                    g[j] = (uint32_t)xbuf[i0] | ((uint32_t)xbuf[i1] << 16);
                }
                gg[q] = g;
            }
        }
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < nq) {
            u4_t g = xq[q];
            if constexpr (GLUE == kGlueRmsNorm) {
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = glue_pair<ACT>(xq[q][j], gq[q][j], inv, GLUE);
            }
            gg[q] = g;
        }
    }
    }
    convert_pieces();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < nq) park(q, gg[q]);
    }

    f4_t acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (RAW) {
        // column c's zero-point term over ALL the wave's chunks: the four rows hold chunks 4 q + 0..3 -> butterfly over the rows
        tsum += __shfl_xor(tsum, 16, 64);
        tsum += __shfl_xor(tsum, 32, 64);
        acc[0] = tsum * up;
    }
    uint32_t magic_hi = 0x54005400u;
    asm volatile("" : "+v"(magic_hi));
    auto compute = [&](const u4_t& wv, int li) __attribute__((always_inline)) {
#if GPTQHIP_SK1_ABLATE & 2
        asm volatile("" ::"v"(wv));     // (the loaded registers are consumed: the waits stay)
        acc[0] += (float)li;
        return;
#endif
        if constexpr (RAW) {
            const float2 ab = reinterpret_cast<const float2*>(ms)[li * 16 + c];
            const u4_t* xr = xs + li * 16 + rq * 4;                         // [lo of K-steps 0|1, hi of 0|1, lo of 2|3, hi of 2|3]
            const f4_t zero4 = {0.f, 0.f, 0.f, 0.f};
            f4_t glo = zero4, ghi = zero4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t w0 = wv[2 * h], w1 = wv[2 * h + 1], w0s = w0 >> 8, w1s = w1 >> 8;
                const u4_t blo = {w0 & dk.lo, w0s & dk.lo, w1 & dk.lo, w1s & dk.lo};
                const u4_t bhi = {w0 & dk.hi, w0s & dk.hi, w1 & dk.hi, w1s & dk.hi};
                glo = mfma16<kFP16>(xr[2 * h], blo, glo);
                ghi = mfma16<kFP16>(xr[2 * h + 1], bhi, ghi);
            }
            acc[0] = __builtin_fmaf(ab.x, glo[0], acc[0]);
            acc[0] = __builtin_fmaf(ab.y, ghi[0], acc[0]);
            return;
        }
        const uint32_t mw = ms[li * 16 + c];
        const u4_t* xa = xs + li * 16 + rq;
        if constexpr (ALG == 1 && ACT == kFP16 && SCL == kFP16) {
            // group-factored dequant (compute_stage's decode forms 2 / 3): exact (q - z) pairs into the matrix pipe (two independent chains),
            // the scale once per chunk in fp32.  Only output row 0 exists at M = 1: accumulator register 0.
            const float sc = bits16_to_f32<SCL>((uint16_t)(mw & 0xffffu));
            const uint32_t zc = mw >> 16;                                  // 0xE400 | z = fp16 -(1024 + z)
            const h2_t zlo = as_h2(zc | (zc << 16));
            const uint32_t zh = 0xD400u | ((zc & 0xFu) << 4);              // fp16 -(64 + z)
            const h2_t zhi = as_h2(zh | (zh << 16));
            f4_t g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w = wv[j], w8 = w >> 8;
                u4_t b;
                b.x = as_u32(as_h2(and_or(w, dk.lo, dk.magic)) + zlo);
                b.y = as_u32(as_h2(and_or(w, dk.hi, magic_hi)) + zhi);
                b.z = as_u32(as_h2(and_or(w8, dk.lo, dk.magic)) + zlo);
                b.w = as_u32(as_h2(and_or(w8, dk.hi, magic_hi)) + zhi);
                if (j & 1) {
                    g1 = mfma16<ACT>(xa[4 * j], b, g1);
                } else {
                    g0 = mfma16<ACT>(xa[4 * j], b, g0);
                }
            }
            acc[0] = __builtin_fmaf(sc, g0[0] + g1[0], acc[0]);
        } else {
            const ColConst cc = expand_meta<4, SCL>(mw);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma16<ACT>(xa[4 * j], dequant_word4<ACT, SCL>(wv[j], cc, dk), acc);
        }
    };
    int cur = c_begin + wave, li = 0;
    for (int it = D; it < n_mine; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            compute(st[d], li);
            load_w(st[d]);
#ifndef GPTQHIP_SK1_NOPIN
            __builtin_amdgcn_sched_barrier(0);   // keep the refill HERE: hipcc otherwise sinks a round's four loads to its end and the ring drains to zero
#endif
            cur += W;
            ++li;
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (cur < c_end) compute(st[d], li);     // (only the last ring round can hold padding chunks: wave-uniform skip)
        cur += W;
        ++li;
    }

#if GPTQHIP_SK1_ABLATE & 4
    if (__builtin_bit_cast(uint32_t, acc[0]) == 0x12345678u) reinterpret_cast<uint16_t*>(p.out)[blockIdx.x] = 1;
    return;
#endif
    // ---- in-block split-K reduction through LDS: only output row 0 exists (accumulator register 0 of the lanes with rq == 0) ----
    red[wave][lane] = acc[0];
    __syncthreads();
    const int m = 4 * rq + wave;
    const int n = tile * kTileN + c;
    const bool live = wave == 0 && m < p.M && n < p.N;
    float v = 0.f;
    if (wave == 0) v = sum_wave_rows(red, W, lane);
    finish_outputs<ACT>(p, v, live, m, n, tile, split, wave, lane, res_raw, s_last);
}

// ------------------------------------------------------------------------------------------------
// The preload form on layers with several 16-column tiles per CU (the fused gate_up: 1792 tiles = 7 per CU).  With one tile per block,
// 47 % of that layer's VMEM instructions are the per-block preloads (x pieces, norm-weight pieces, statistics, constants: 7 per wave
// next to 8 weight loads) and every one of the 1792 blocks pays a launch, a prologue and an epilogue.  Here ONE block per CU (W = K / 512
// waves, each owning four chunks of every tile = one ring round per tile) walks tiles b, b + grid, ...: the glued x pieces are parked
// once, the weight ring runs on across tile boundaries (a tile's loads are in flight while the previous one is reduced), a tile costs
// one constants load per wave (prefetched a tile ahead into registers, parked in the wave's own double-buffered LDS rows: no barrier),
// and its 16 outputs leave through finish_outputs on wave (tile index % W) after ONE block barrier (reduction rows double-buffered by
// tile parity).  No residual / bias / cross-block split here: the layers that have them (o_proj, down_proj) have one tile per CU.
// ------------------------------------------------------------------------------------------------
// PERM (act-order checkpoints, like skinny1_kernel's): the block stages the GLUED x row in LDS once (natural order, one barrier), every wave gathers
// the eight elements per lane of each of its D chunks from there by their indices -- once per block, the tile loop is unchanged.
template <int ACT, int SCL, int GLUE, int ALG, int D = 4, bool PERM = false>
__global__ __launch_bounds__(1024) void skinny1p_kernel(SkinnyParams p) {
    static_assert(D == 4 || D == 8, "ring depth = chunks per wave and tile");
    constexpr int NQ = D / 4;                       // 16-byte preload instructions per wave (four chunks each)
    constexpr bool RAW = ALG == 2 && ((ACT == kFP16 && SCL == kFP16) || ACT == kBF16);     // raw codes as fp16 denormals (see skinny1_kernel)
    constexpr bool XCVT = RAW && ACT == kBF16;      // bf16 activations ride the f16 matrix pipe (see bf16pair_to_f16pair)
    constexpr int kSlot = D * (RAW ? 512 : 384);    // x pieces (D * 256 B) + constants double-buffered by tile parity (2 * D * 64 B; RAW: float2)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int c = lane & 15, rq = lane >> 4;
    const int tiles = (p.N + kTileN - 1) / kTileN;
    char* const slot = reinterpret_cast<char*>(lds) + wave * kSlot;
    u4_t* const xs = reinterpret_cast<u4_t*>(slot);                            // [D][16] u4: this wave's D glued x pieces
    uint32_t* const ms = reinterpret_cast<uint32_t*>(slot + D * 256);          // [2][D][16] constants, double-buffered by tile parity
    float* scratch = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + W * kSlot);
    float(*red)[16][64] = reinterpret_cast<float(*)[16][64]>(reinterpret_cast<char*>(lds) + W * kSlot + 96);    // [2][W <= 16][64]  (nb_tiles: [tiles per block][16][64])
    // p.nb_tiles > 0: no block barrier per tile.  Every tile of the block has its own reduction rows and an arrival counter in LDS; a wave
    // writes its partial sums, counts itself in (release) and goes on with the next tile; the tile's owner (tile index % W) combines the rows
    // ONE TILE LATER -- after its own stages of the next tile, when the others have normally arrived -- spinning on the counter if not.
    const int nb = p.nb_tiles;
    int* const cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(lds) + W * kSlot + 96 + (size_t)(nb > 0 ? nb : 2) * 16 * 256);
    if (nb > 0) {
        if ((int)threadIdx.x < nb) cnt[threadIdx.x] = 0;
        __syncthreads();
    }
    const DequantConsts dk = make_dequant_consts<4>();
    const uint32_t lane16 = (uint32_t)lane * 16u, c4 = (uint32_t)c * 4u;
    // the chunk lane (rq, c) serves in preload instruction q: wave + (4 q + rq) * W
    const size_t tile_w = (size_t)p.chunks * 1024, tile_m = (size_t)p.G * 64;

    f4_t sv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if constexpr (GLUE == kGlueRmsNorm) {
        if (p.stats_in != nullptr) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.stats_in), 0, p.stats_n * 4, 0x00020000);
            sv[0] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, 0, 0));
            if (p.stats_n > 256) sv[1] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, 1024, 0));
        }
    }
    u4_t xq[NQ], gq[NQ];
    uint32_t mq[NQ];
    u4_t pq[PERM ? NQ : 1][2];         // PERM: perm[k'] of the lane's eight rows of each chunk
    u4_t xr[2], gr[2];                 // PERM: this thread's 16-byte pieces of the x row / norm weight (staged once per block)
    uint16_t* const xbuf = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(lds) + W * kSlot + 96 + (size_t)(nb > 0 ? nb : 2) * 16 * 256 + 64);
    const int n16 = p.K / 8;
    int tile = blockIdx.x;
    const char* mrow[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int ck_lane = wave + (4 * q + rq) * W;
        if constexpr (PERM) {
            const u4_t* pp = reinterpret_cast<const u4_t*>(p.perm + (size_t)ck_lane * 128 + c * 8);
            pq[q][0] = pp[0];
            pq[q][1] = pp[1];
        } else {
            xq[q] = *reinterpret_cast<const u4_t*>(reinterpret_cast<const char*>(p.x) + (size_t)ck_lane * 256 + c * 16);
            gq[q] = u4_t{0u, 0u, 0u, 0u};
            if constexpr (GLUE == kGlueRmsNorm) gq[q] = *reinterpret_cast<const u4_t*>(reinterpret_cast<const char*>(p.glue_b) + (size_t)ck_lane * 256 + c * 16);
        }
        mrow[q] = reinterpret_cast<const char*>(p.meta) + (size_t)tile * tile_m + ((size_t)(ck_lane >> p.cpg_shift) << 6) + c4;
        mq[q] = *reinterpret_cast<const uint32_t*>(mrow[q]);
    }
    if constexpr (PERM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = (int)threadIdx.x + i * (int)blockDim.x;
            xr[i] = reinterpret_cast<const u4_t*>(p.x)[idx < n16 ? idx : 0];
            gr[i] = u4_t{0u, 0u, 0u, 0u};
            if constexpr (GLUE == kGlueRmsNorm) gr[i] = reinterpret_cast<const u4_t*>(p.glue_b)[idx < n16 ? idx : 0];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const char* wsrc = reinterpret_cast<const char*>(p.qw) + (size_t)tile * tile_w + (size_t)wave * 1024 + lane16;
    const size_t wstep = (size_t)W * 1024, tstep_w = (size_t)gridDim.x * tile_w, tstep_m = (size_t)gridDim.x * tile_m;
    u4_t st[D];
#pragma unroll
    for (int d = 0; d < D; ++d) st[d] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(wsrc + d * wstep));

    float inv = 1.f;
    if constexpr (GLUE == kGlueRmsNorm) {
        if (p.stats_in != nullptr) {
            const float ssum = wave64_sum(((sv[0][0] + sv[0][1]) + (sv[0][2] + sv[0][3])) + ((sv[1][0] + sv[1][1]) + (sv[1][2] + sv[1][3])));
            inv = rsqrtf(ssum / (float)p.K + p.eps);
        } else {
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = bits16_to_f32<ACT>((uint16_t)(xq[q][j] & 0xffffu)), b = bits16_to_f32<ACT>((uint16_t)(xq[q][j] >> 16));
                    ss = __builtin_fmaf(a, a, ss);
                    ss = __builtin_fmaf(b, b, ss);
                }
            }
            ss = wave64_sum(ss);
            if (lane == 0) scratch[1 + wave] = ss;
            __syncthreads();
            if (threadIdx.x == 0) {
                float tot = 0.f;
                for (int w = 0; w < W; ++w) tot += scratch[1 + w];
                scratch[0] = rsqrtf(tot / (float)p.K + p.eps);
            }
            __syncthreads();
            inv = scratch[0];
        }
        if constexpr (!PERM) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xq[q][j] = glue_pair<ACT>(xq[q][j], gq[q][j], inv, GLUE);
        }
        }
    }
    if constexpr (PERM) {
        // (the launcher sends RMSNorm ops here only with producer statistics: the in-block reduction above reads the waves' natural pieces)
        auto glued = [&](const u4_t& xv, const u4_t& gv) __attribute__((always_inline)) {
            u4_t r = xv;
            if constexpr (GLUE == kGlueRmsNorm) {
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = glue_pair<ACT>(xv[j], gv[j], inv, GLUE);
            }
            return r;
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = (int)threadIdx.x + i * (int)blockDim.x;
            if (idx < n16) reinterpret_cast<u4_t*>(xbuf)[idx] = glued(xr[i], gr[i]);
        }
        for (int idx = (int)threadIdx.x + 2 * (int)blockDim.x; idx < n16; idx += (int)blockDim.x) {   // (rows longer than 32 B x threads: rare)
            u4_t gv = {0u, 0u, 0u, 0u};
            if constexpr (GLUE == kGlueRmsNorm) gv = reinterpret_cast<const u4_t*>(p.glue_b)[idx];
            reinterpret_cast<u4_t*>(xbuf)[idx] = glued(reinterpret_cast<const u4_t*>(p.x)[idx], gv);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t i0 = pq[q][j >> 1][(j & 1) * 2], i1 = pq[q][j >> 1][(j & 1) * 2 + 1];
                xq[q][j] = (uint32_t)xbuf[i0] | ((uint32_t)xbuf[i1] << 16);
            }
        }
    }
    float sx[NQ];      // RAW: Sx of chunk wave + (4 q + rq) W (the same for every tile)
    float up = 1.f;    // XCVT: 2^k, the power of two taken out of this wave's x pieces
    if constexpr (XCVT) {
        const int k = bf16_down_shift(wave_max_abs_bf16(xq, NQ));
        const float dn = __builtin_bit_cast(float, (uint32_t)(127 - k) << 23);
        up = __builtin_bit_cast(float, (uint32_t)(127 + k) << 23);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xq[q][j] = bf16pair_to_f16pair(xq[q][j], dn);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if constexpr (RAW) {
            const int li = 4 * q + rq, j = c >> 2;
            char* base = reinterpret_cast<char*>(xs) + li * 256 + (c & 3) * 64 + (j >> 1) * 32 + (j & 1) * 8;
            *reinterpret_cast<u2_t*>(base) = u2_t{xq[q][0], xq[q][2]};
            *reinterpret_cast<u2_t*>(base + 16) = u2_t{xq[q][1], xq[q][3]};
            const h2_t ones = as_h2(0x3C003C00u);
            float s8 = __builtin_amdgcn_fdot2(as_h2(xq[q][0]), ones, 0.f, false);
            s8 = __builtin_amdgcn_fdot2(as_h2(xq[q][1]), ones, s8, false);
            s8 = __builtin_amdgcn_fdot2(as_h2(xq[q][2]), ones, s8, false);
            s8 = __builtin_amdgcn_fdot2(as_h2(xq[q][3]), ones, s8, false);
            sx[q] = row16_sum(s8);
        } else {
            xs[q * 64 + lane] = xq[q];
        }
    }

    uint32_t magic_hi = 0x54005400u;
    asm volatile("" : "+v"(magic_hi));
    float acc = 0.f;
#if GPTQHIP_SK1P_XREG   // dev A/B (-DGPTQHIP_SK1P_XREG=1): the wave's x fragments (the same for every tile of the block) kept in registers instead of re-read
                        // from LDS per chunk: gate_up 12.58 -> 12.43 us (same box, two rounds), but 64 more registers put the instantiation at the
                        // 128-VGPR cap of a 1024-thread launch bound with 12 bytes of scratch per lane: not adopted for 0.3 % of a token
    u4_t xreg[D][4];
    if constexpr (RAW) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) xreg[d][i] = xs[d * 16 + rq * 4 + i];
    }
#endif
    auto compute = [&](const u4_t& wv, int li, const uint32_t* mcur) __attribute__((always_inline)) {
        if constexpr (RAW) {
            const float2 ab = reinterpret_cast<const float2*>(mcur)[li * 16 + c];
#if GPTQHIP_SK1P_XREG
            const u4_t* xr = xreg[li];
#else
            const u4_t* xr = xs + li * 16 + rq * 4;
#endif
            const f4_t zero4 = {0.f, 0.f, 0.f, 0.f};
            f4_t glo = zero4, ghi = zero4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t w0 = wv[2 * h], w1 = wv[2 * h + 1], w0s = w0 >> 8, w1s = w1 >> 8;
                const u4_t blo = {w0 & dk.lo, w0s & dk.lo, w1 & dk.lo, w1s & dk.lo};
                const u4_t bhi = {w0 & dk.hi, w0s & dk.hi, w1 & dk.hi, w1s & dk.hi};
                glo = mfma16<kFP16>(xr[2 * h], blo, glo);
                ghi = mfma16<kFP16>(xr[2 * h + 1], bhi, ghi);
            }
            acc = __builtin_fmaf(ab.x, glo[0], acc);
            acc = __builtin_fmaf(ab.y, ghi[0], acc);
            return;
        }
        const uint32_t mw = mcur[li * 16 + c];
        const u4_t* xa = xs + li * 16 + rq;
        if constexpr (ALG == 1 && ACT == kFP16 && SCL == kFP16) {
            const float sc = bits16_to_f32<SCL>((uint16_t)(mw & 0xffffu));
            const uint32_t zc = mw >> 16;
            const h2_t zlo = as_h2(zc | (zc << 16));
            const uint32_t zh = 0xD400u | ((zc & 0xFu) << 4);
            const h2_t zhi = as_h2(zh | (zh << 16));
            f4_t g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w = wv[j], w8 = w >> 8;
                u4_t b;
                b.x = as_u32(as_h2(and_or(w, dk.lo, dk.magic)) + zlo);
                b.y = as_u32(as_h2(and_or(w, dk.hi, magic_hi)) + zhi);
                b.z = as_u32(as_h2(and_or(w8, dk.lo, dk.magic)) + zlo);
                b.w = as_u32(as_h2(and_or(w8, dk.hi, magic_hi)) + zhi);
                if (j & 1) {
                    g1 = mfma16<ACT>(xa[4 * j], b, g1);
                } else {
                    g0 = mfma16<ACT>(xa[4 * j], b, g0);
                }
            }
            acc = __builtin_fmaf(sc, g0[0] + g1[0], acc);
        } else {
            const ColConst cc = expand_meta<4, SCL>(mw);
            f4_t g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) g = mfma16<ACT>(xa[4 * j], dequant_word4<ACT, SCL>(wv[j], cc, dk), g);
            acc += g[0];
        }
    };
    // One tile: MORE = another tile follows (its loads are issued UNCONDITIONALLY behind each stage, so hipcc's waits stay counted;
    // with `if (more)` around them it fell back to vmcnt(0) in the middle of the round).  The last tile runs the drain instantiation.
    int ti = 0;
    auto combine = [&](int t_idx, int tile_id) __attribute__((always_inline)) {
        while (__hip_atomic_load(cnt + t_idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < W) __builtin_amdgcn_s_sleep(1);
        const float v = sum_wave_rows(red[t_idx], W, lane);
        const int n = tile_id * kTileN + c;
        finish_outputs<ACT>(p, v, rq == 0 && n < p.N, 4 * rq, n, tile_id, 0, 0, lane, 0u, nullptr);
    };
    auto do_tile = [&](auto more_c) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_c)::value;
        uint32_t* mcur = ms + (ti & 1) * (D * 16) * (RAW ? 2 : 1);
        if constexpr (RAW) {
            float tsum = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float sc = bits16_to_f32<SCL>((uint16_t)(mq[q] & 0xffffu));
                const float z = (float)((mq[q] >> 16) & 0xFu);
                reinterpret_cast<float2*>(mcur)[q * 64 + lane] = float2{sc * 16777216.f * up, sc * 1048576.f * up};
                tsum = __builtin_fmaf(-(sc * z), sx[q], tsum);
            }
            tsum += __shfl_xor(tsum, 16, 64);
            tsum += __shfl_xor(tsum, 32, 64);
            acc = tsum * up;     // (the previous tile's sum left through `red` and acc was reset)
        } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) mcur[q * 64 + lane] = mq[q];     // this tile's constants (wave-private rows: no barrier)
        }
        if constexpr (MORE) {                                // the next tile's: one instruction per four chunks, a tile ahead
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                mrow[q] += tstep_m;
                mq[q] = *reinterpret_cast<const uint32_t*>(mrow[q]);
            }
            wsrc += tstep_w;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            compute(st[d], d, mcur);
            if constexpr (MORE) st[d] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(wsrc + d * wstep));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nb > 0) {
            red[ti][wave][lane] = acc;
            acc = 0.f;
            if (lane == 0) __hip_atomic_fetch_add(cnt + ti, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ti >= 1 && wave == (ti - 1) % W) combine(ti - 1, tile - (int)gridDim.x);
        } else {
        // in-block split-K reduction, rows double-buffered by tile parity: ONE barrier per tile; the waves take turns with the epilogue
        red[ti & 1][wave][lane] = acc;
        acc = 0.f;
        __syncthreads();
        if (wave == ti % W) {
            const float v = sum_wave_rows(red[ti & 1], W, lane);
            const int n = tile * kTileN + c;
            finish_outputs<ACT>(p, v, rq == 0 && n < p.N, 4 * rq, n, tile, 0, 0, lane, 0u, nullptr);
        }
        }
        tile += (int)gridDim.x;
        ++ti;
    };
    while (tile + (int)gridDim.x < tiles) do_tile(std::true_type{});
    do_tile(std::false_type{});
    if (nb > 0 && wave == (ti - 1) % W) combine(ti - 1, tile - (int)gridDim.x);
}

// tiles per block of the persistent variant; 0: not applicable
static int skinny1p_depth() {   // dev A/B switch: GPTQHIP_SK1P_D8=1 runs chunks / 8 waves with an 8-deep ring
    static const int d = [] { const char* v = getenv("GPTQHIP_SK1P_D8"); return (v && *v && *v != '0') ? 8 : 4; }();
    return d;
}
static int skinny1p_grid(const SkinnyParams& p, const SkinnyPlan& pl) {
    static const bool off = [] { const char* v = getenv("GPTQHIP_NO_PERSIST"); return v && *v && *v != '0'; }();
    const int tiles = ceil_div(p.N, kTileN), cus = 256;
    if (off || p.splits != 1 || p.residual != nullptr || p.bias != nullptr || p.out_f32) return 0;
    if (tiles < 2 * cus || tiles % cus != 0 || p.N % kTileN != 0) return 0;
    const int d = skinny1p_depth();
    // act-order: the staged x row shares the default dynamic LDS with the slots; RMSNorm ops need producer statistics there
    if (p.perm != nullptr && (d != 4 || (size_t)p.K * 2 > 24 * 1024 || (p.in_glue == kGlueRmsNorm && p.stats_in == nullptr))) return 0;
    if (p.chunks % d != 0 || p.chunks / d < 4 || p.chunks / d > 16) return 0;
    return cus;
}

constexpr int kPreloadMaxChunks = 16;   // chunks per wave the preload form parks (four 16-byte x instructions)

// 0 when the call is outside the preload form
template <int ACT, int SCL>
static int launch_skinny1(const SkinnyParams& p0, const SkinnyPlan& pl, int alg, hipStream_t stream, bool* served) {
    *served = false;
    if (p0.M != 1 || pl.mt != 1 || pl.gpc != 1 || !pl.regular || pl.nt > 1 || p0.n_mine > kPreloadMaxChunks) return 0;
    const bool perm = p0.perm != nullptr;
    if (perm && (pl.depth != 4 || p0.splits != 1 || (size_t)p0.K * 2 > 100 * 1024)) return 0;      // (the planner hands act-order calls the 4-deep ring)
    if (p0.in_glue != kGlueNone && p0.in_glue != kGlueRmsNorm) return 0;
    if (p0.in_glue == kGlueRmsNorm && p0.stats_in == nullptr && p0.splits > 1) return 0;
    if (p0.exact_bf16 || (pl.depth != 2 && pl.depth != 4)) return 0;
    SkinnyParams p = p0;
    const bool a1 = alg != 0 && ACT == kFP16 && SCL == kFP16;   // (bf16 scales: the reference rounds W to bf16 -- 2^-9 per weight -- keep its chain)
    const bool a2 = alg == 2 && (a1 || ACT == kBF16);           // raw codes as fp16 denormals (decode form 5; bf16 activations converted per wave)
    // (a wave-per-tile kernel for many-tile layers -- every wave owning whole tiles, no barrier / reduction per tile -- was built, parity-tested and measured
    // slower than skinny1p_kernel, with and without register spills: gate_up 13.5-13.6 vs 12.5-12.7 us; commits b4141bd..this one's parent, profiles/r06_decode_forms.txt)
    if (const int pg = skinny1p_grid(p0, pl)) {
        const int d = skinny1p_depth();
        const int waves = p.chunks / d;
        const dim3 grid(pg), block(64 * waves);
        // reduction rows per tile + arrival counters instead of a block barrier per tile, when they fit the default dynamic LDS (A/B: GPTQHIP_SK1P_BARRIER=1)
        // OPT-IN (GPTQHIP_SK1P_NOBARRIER=1): same-box A/B on the 8B gate_up 12.58-12.67 vs 12.68-12.81 us fp16 (inside the noise), 15.99 vs 15.62 bf16 (slower)
        static const bool force_barrier = [] { const char* v = getenv("GPTQHIP_SK1P_NOBARRIER"); return !(v && *v && *v != '0'); }();
        const int tpb = ceil_div(ceil_div(p.N, kTileN), pg);
        const size_t lds_nb = (size_t)waves * d * (a2 ? 512 : 384) + 96 + (size_t)tpb * 16 * 256 + 64;
        p.nb_tiles = (!force_barrier && tpb <= 16 && lds_nb + (perm ? (size_t)p.K * 2 : 0) <= 64 * 1024) ? tpb : 0;
        const size_t lds_bytes = (p.nb_tiles > 0 ? lds_nb : (size_t)waves * d * (a2 ? 512 : 384) + 96 + 2 * 16 * 256 + 64) + (perm ? (size_t)p.K * 2 : 0);
#define GPTQHIP_L1P(G_, A_, D_) hipLaunchKernelGGL((skinny1p_kernel<ACT, SCL, G_, A_, D_>), grid, block, lds_bytes, stream, p)
#define GPTQHIP_L1PP(G_, A_) hipLaunchKernelGGL((skinny1p_kernel<ACT, SCL, G_, A_, 4, true>), grid, block, lds_bytes, stream, p)
        if (perm) {      // (d == 4 here: skinny1p_grid)
            if (p.in_glue == kGlueRmsNorm) { if (a2) GPTQHIP_L1PP(kGlueRmsNorm, 2); else GPTQHIP_L1PP(kGlueRmsNorm, 0); }
            else { if (a2) GPTQHIP_L1PP(kGlueNone, 2); else GPTQHIP_L1PP(kGlueNone, 0); }
        } else if (a2 && d == 4) {
            if (p.in_glue == kGlueRmsNorm) GPTQHIP_L1P(kGlueRmsNorm, 2, 4); else GPTQHIP_L1P(kGlueNone, 2, 4);
        } else if (a2 && d == 8) {
            if (p.in_glue == kGlueRmsNorm) GPTQHIP_L1P(kGlueRmsNorm, 2, 8); else GPTQHIP_L1P(kGlueNone, 2, 8);
        } else if (d == 8) {
            if (p.in_glue == kGlueRmsNorm) { if (a1) GPTQHIP_L1P(kGlueRmsNorm, 1, 8); else GPTQHIP_L1P(kGlueRmsNorm, 0, 8); }
            else { if (a1) GPTQHIP_L1P(kGlueNone, 1, 8); else GPTQHIP_L1P(kGlueNone, 0, 8); }
        } else {
            if (p.in_glue == kGlueRmsNorm) { if (a1) GPTQHIP_L1P(kGlueRmsNorm, 1, 4); else GPTQHIP_L1P(kGlueRmsNorm, 0, 4); }
            else { if (a1) GPTQHIP_L1P(kGlueNone, 1, 4); else GPTQHIP_L1P(kGlueNone, 0, 4); }
        }
#undef GPTQHIP_L1P
#undef GPTQHIP_L1PP
        *served = true;
        return check_hip(hipGetLastError(), "skinny1p_kernel launch");
    }
    // Geometry: the planner's (waves, depth), or -- dev A/B switch GPTQHIP_SK1_D8=1 -- half the waves with an 8-deep ring (every wave's eight
    // chunks requested up front: the same bytes in flight per CU from half the waves to start)
    static const int d8_mode = [] { const char* v = getenv("GPTQHIP_SK1_D8"); return (v && *v) ? atoi(v) : 0; }();
    int waves = pl.waves, depth = pl.depth;
    if (d8_mode > 0 && !perm && p.splits == 1 && p.chunks % 8 == 0 && p.chunks / 8 >= 4 && p.chunks / 8 <= 16 && (d8_mode >= 2 || p.chunks / 8 <= 8)) {
        waves = p.chunks / 8;
        depth = 8;
        p.n_mine = 8;
    }
    const bool raw = a2;
    const int stride = ((p.n_mine + 3) >> 2) * (raw ? 1536 : 1280);     // x pieces + constants (>= the 1 KiB per wave the reduction rows need)
    p.slot_stride = stride;
    const dim3 grid(ceil_div(p.N, kTileN), p.splits), block(64 * waves);
    const size_t lds_bytes = (size_t)waves * stride + 96 + (size_t)waves * 256 + (perm ? (size_t)p.K * 2 : 0);   // slots | flag + scratch | reduction rows | x row (act-order)
#define GPTQHIP_L1(D_, G_, A_) hipLaunchKernelGGL((skinny1_kernel<ACT, SCL, D_, G_, A_>), grid, block, lds_bytes, stream, p)
    if (perm) {
        // (the staged x row can push the block past the default 64 KiB of dynamic LDS: raise the limit once per instantiation)
        auto go = [&](auto kern) {
            static bool attr_done = false;
            if (!attr_done && lds_bytes > 64 * 1024) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return check_hip(e, "skinny1_kernel: hipFuncSetAttribute");
                attr_done = true;
            }
            hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, p);
            return check_hip(hipGetLastError(), "skinny1_kernel (act-order) launch");
        };
        *served = true;
        if (raw) return p.in_glue == kGlueRmsNorm ? go(skinny1_kernel<ACT, SCL, 4, kGlueRmsNorm, 2, true>) : go(skinny1_kernel<ACT, SCL, 4, kGlueNone, 2, true>);
        if (p.in_glue == kGlueRmsNorm) return a1 ? go(skinny1_kernel<ACT, SCL, 4, kGlueRmsNorm, 1, true>) : go(skinny1_kernel<ACT, SCL, 4, kGlueRmsNorm, 0, true>);
        return a1 ? go(skinny1_kernel<ACT, SCL, 4, kGlueNone, 1, true>) : go(skinny1_kernel<ACT, SCL, 4, kGlueNone, 0, true>);
    } else if (depth == 8) {
        if (p.in_glue == kGlueRmsNorm) { if (raw) GPTQHIP_L1(8, kGlueRmsNorm, 2); else if (a1) GPTQHIP_L1(8, kGlueRmsNorm, 1); else GPTQHIP_L1(8, kGlueRmsNorm, 0); }
        else { if (raw) GPTQHIP_L1(8, kGlueNone, 2); else if (a1) GPTQHIP_L1(8, kGlueNone, 1); else GPTQHIP_L1(8, kGlueNone, 0); }
    } else if (depth == 4) {
        if (p.in_glue == kGlueRmsNorm) { if (raw) GPTQHIP_L1(4, kGlueRmsNorm, 2); else if (a1) GPTQHIP_L1(4, kGlueRmsNorm, 1); else GPTQHIP_L1(4, kGlueRmsNorm, 0); }
        else { if (raw) GPTQHIP_L1(4, kGlueNone, 2); else if (a1) GPTQHIP_L1(4, kGlueNone, 1); else GPTQHIP_L1(4, kGlueNone, 0); }
    } else {
        if (p.in_glue == kGlueRmsNorm) { if (raw) GPTQHIP_L1(2, kGlueRmsNorm, 2); else if (a1) GPTQHIP_L1(2, kGlueRmsNorm, 1); else GPTQHIP_L1(2, kGlueRmsNorm, 0); }
        else { if (raw) GPTQHIP_L1(2, kGlueNone, 2); else if (a1) GPTQHIP_L1(2, kGlueNone, 1); else GPTQHIP_L1(2, kGlueNone, 0); }
    }
#undef GPTQHIP_L1
    *served = true;
    return check_hip(hipGetLastError(), "skinny1_kernel launch");
}

// ------------------------------------------------------------------------------------------------
// Wide layers (N >= 8192 columns: fused gate_up, lm_head, the 70B projections) at 5..32 rows.  skinny_kernel gives every 16-column tile its own
// block, so each of the N/16 blocks stages the WHOLE activation tile (M x K) from L2 through LDS: 1792 x 256 KiB = 460 MB of
// on-chip traffic for 61 MB of weights on a Llama-3-8B gate_up at M = 32 (profiles/r03_mid_m_sweep.txt: 32 us where the weights
// stream in 14.6 us at M = 1).  Here a block owns NT = 4 (2) adjacent column tiles: per 128-row chunk a wave parks its activation
// rows in LDS ONCE, reads each K-step's A fragments ONCE and multiplies them with the four tiles' dequantised words (4 KiB of
// weights per ring stage and wave instead of 1).  Same ingredients as the split-ring pipeline above -- weight ring two deep,
// activation stage one deep through buffer loads, clamped (never conditional) loads so the waits stay counted, in-block split-K
// with an LDS reduction, fixed summation order -- specialised to what this regime needs: 4-bit weights, no glue, no cross-block
// split (N/64 >= 256 blocks fill the chip), the regular pipeline only.  Replaces nothing upstream beyond what skinny_kernel does
// (TorchLinear._forward_eager, torch.py:326-347).
// ------------------------------------------------------------------------------------------------
// 4 waves per SIMD (<= 128 VGPRs: two blocks per CU) wherever that does not push registers to scratch (ISA audit): not with
// per-K-step group constants, not for the bf16 / bf16 glue variant on 9..16 rows, not for 17..32 rows
template <int ACT, int SCL, int MT, int GPC, bool HALFQ, int GLUE, int NT = 4>
__host__ __device__ constexpr int wide_waves_per_simd() {
    // (17..32 rows: two tiles per block fit 128 VGPRs too, but two such blocks per CU measured equal to one four-tile block:
    // profiles/r03_wide_layers.txt)
    return (MT == 1 && GPC == 1 && !(GLUE != 0 && ACT == kBF16 && SCL == kBF16 && !HALFQ)) ? 4 : 2;
}

template <int GPC, int NT>
struct WideStage {
    u4_t w[NT];
    uint32_t meta[NT * GPC];
};

// GLUE (decode op on 5..16 rows, MT == 1 only): kGlueRmsNorm applies HF's RMSNorm to the rows while they are parked (per-row 1/rms
// from the producer's per-tile statistics or an in-kernel reduction, like skinny_kernel); the epilogue then also serves
// GPTQHIP_OUT_SILU_MUL_PAIRED (interleaved gate|up tiles).  Residual / stats_out epilogues stay with the one-tile kernel: the layers
// that use them (o_proj, down_proj) are not wide.
// Up to 16 rows (MT == 1) two 8-wave blocks share a CU (LDS: 2 x 35 KiB) as long as a wave stays within 128 VGPRs: the glue variant
// came out at 129-131 and ran ONE block per CU (gate_up with glue 23.5 us vs 16.7 without) -- hence the waves-per-SIMD hint.
template <int ACT, int SCL, int MT, int GPC, bool HALFQ, int NT, int GLUE = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(wide_waves_per_simd<ACT, SCL, MT, GPC, HALFQ, GLUE, NT>())))
void skinny_wide_kernel(SkinnyParams p) {
    static_assert(GLUE == 0 || MT == 1, "the decode op takes at most 16 rows");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int QUADS = HALFQ ? 4 * MT - 2 : 4 * MT;          // 4-row groups loaded per chunk (rows beyond them: never stored)
    constexpr int NR = MT * NT * 4;                               // accumulator registers per lane
    constexpr int kSlotA = 16 * MT * kRowsPitch * 16;
    constexpr int kSlot = kSlotA > NR * 256 ? kSlotA : NR * 256;  // bytes per wave: activation slot, later its reduction rows
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int c = lane & 15, rq = lane >> 4;
    const int tile0 = blockIdx.x * NT;
    const int c_end = p.chunks;
    const DequantConsts dk = make_dequant_consts<4>();

    f4_t acc[MT * NT];
#pragma unroll
    for (int i = 0; i < MT * NT; ++i) acc[i] = f4_t{0.f, 0.f, 0.f, 0.f};

    const char* wbase = reinterpret_cast<const char*>(p.qw) + (size_t)tile0 * p.chunks * 1024;
    const char* mbase = reinterpret_cast<const char*>(p.meta) + (size_t)tile0 * p.G * 64;
    const size_t wstride = (size_t)p.chunks * 1024, mstride = (size_t)p.G * 64;      // next column tile
    const uint32_t lane16 = (uint32_t)lane * 16u, c4 = (uint32_t)c * 4u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.M * p.K * 2, 0x00020000);
    const uint32_t voff = (uint32_t)rq * (uint32_t)p.K * 2u + (uint32_t)c * 16u;
    const uint32_t quad_stride = (uint32_t)p.K * 8u;
    u4_t* aslot = reinterpret_cast<u4_t*>(reinterpret_cast<char*>(lds) + wave * kSlot);

    WideStage<GPC, NT> st[2];
    u4_t xa[QUADS];
    u4_t xnw = {0u, 0u, 0u, 0u};       // GLUE: the chunk's norm-weight segment of this lane's 16-byte column piece
    GlueInv ginv;
    float* scratch = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + W * kSlot);   // GLUE: 16 x 1/rms
    float sv[GLUE != 0 ? 4 : 1][8];
    if constexpr (GLUE == kGlueRmsNorm) {
        // wave w (< 4) owns the statistics of rows w, w + 4, w + 8, w + 12: the producer's per-tile sums of squares are requested in
        // FRONT of the weight ring (eight clamped loads per lane and row)
        if (p.stats_in != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave + 4 * r;
                if (wave < 4 && row < p.M) {
                    const float* srow = p.stats_in + (size_t)row * p.stats_n;
                    const int last = p.stats_n - 1;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int idx = lane + 64 * i;
                        sv[r][i] = srow[idx < last ? idx : last];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    auto load_w = [&](WideStage<GPC, NT>& s, int chunk) __attribute__((always_inline)) {
        const int ck = chunk < c_end ? chunk : c_end - 1;       // padding chunks of the last ring round re-fetch the last real one
        const char* src = wbase + (size_t)ck * 1024;
#pragma unroll
        for (int t = 0; t < NT; ++t) s.w[t] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(src + t * wstride + lane16));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < GPC; ++j) {
                const int g = GPC == 1 ? (ck >> p.cpg_shift) : ((ck * 4 + j) >> p.cpg_shift);
                s.meta[t * GPC + j] = *reinterpret_cast<const uint32_t*>(mbase + t * mstride + ((size_t)g << 6) + c4);
            }
    };
    auto load_a = [&](int chunk) __attribute__((always_inline)) {
        const int ck = chunk < c_end ? chunk : c_end - 1;
#pragma unroll
        for (int i = 0; i < QUADS; ++i) xa[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, (uint32_t)ck * 256u + (uint32_t)i * quad_stride, 0);
        if constexpr (GLUE == kGlueRmsNorm)
            xnw = *reinterpret_cast<const u4_t*>(reinterpret_cast<const char*>(p.glue_b) + (size_t)ck * 256 + c * 16);
    };
    auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < QUADS; ++i) {
            if constexpr (GLUE == kGlueRmsNorm) {
                u4_t g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = glue_pair<ACT>(xa[i][j], xnw[j], ginv.v[i], GLUE);
                aslot[(4 * i + rq) * kRowsPitch + c] = g;
            } else {
                aslot[(4 * i + rq) * kRowsPitch + c] = xa[i];
            }
        }
    };
    auto multiply = [&](const WideStage<GPC, NT>& s) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (MT >= 4) {
                // 64 rows: the fragments are re-read per column tile (two LDS reads instead of one) rather than held in 16 registers
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const ColConst cc = expand_meta<4, SCL>(s.meta[t * GPC + (GPC == 4 ? j : 0)]);
                    const u4_t b = dequant_word4<ACT, SCL>(s.w[t][j], cc, dk);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt * NT + t] = mfma16<ACT>(aslot[(16 * mt + c) * kRowsPitch + 4 * j + rq], b, acc[mt * NT + t]);
                }
            } else {
                u4_t av[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = aslot[(16 * mt + c) * kRowsPitch + 4 * j + rq];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const ColConst cc = expand_meta<4, SCL>(s.meta[t * GPC + (GPC == 4 ? j : 0)]);
                    const u4_t b = dequant_word4<ACT, SCL>(s.w[t][j], cc, dk);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt * NT + t] = mfma16<ACT>(av[mt], b, acc[mt * NT + t]);
                }
            }
        }
    };

    int cur = wave;
    load_a(cur);                       // (waited for first)
    load_w(st[0], cur);
    load_w(st[1], cur + W);
    if constexpr (GLUE == kGlueRmsNorm) {
        if (p.stats_in != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave + 4 * r;
                if (wave < 4 && row < p.M) {
                    float ssum = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) ssum += (lane + 64 * i < p.stats_n) ? sv[r][i] : 0.f;
#pragma unroll
                    for (int mk = 32; mk >= 1; mk >>= 1) ssum += __shfl_xor(ssum, mk, 64);
                    if (lane == 0) scratch[row] = rsqrtf(ssum / (float)p.K + p.eps);
                }
            }
        } else {   // no producer statistics (the step's first op): a wave-local reduction of the row, fixed order
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave + 4 * r;
                if (wave < 4 && row < p.M) {
                    const u4_t* hs = reinterpret_cast<const u4_t*>(p.x) + (size_t)row * (p.K / 8);
                    float ss = 0.f;
                    for (int idx = lane; idx < p.K / 8; idx += 64) {
                        const u4_t h = hs[idx];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = bits16_to_f32<ACT>((uint16_t)(h[j] & 0xffffu)), b = bits16_to_f32<ACT>((uint16_t)(h[j] >> 16));
                            ss = __builtin_fmaf(a, a, ss);
                            ss = __builtin_fmaf(b, b, ss);
                        }
                    }
#pragma unroll
                    for (int mk = 32; mk >= 1; mk >>= 1) ss += __shfl_xor(ss, mk, 64);
                    if (lane == 0) scratch[row] = rsqrtf(ss / (float)p.K + p.eps);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < QUADS; ++i) ginv.v[i] = scratch[4 * i + rq < p.M ? 4 * i + rq : 0];
    }
    for (int it = 2; it < p.n_mine; it += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            park();
            load_a(cur + W);
            multiply(st[d]);
            load_w(st[d], cur + 2 * W);
            cur += W;
        }
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if (cur < c_end) {             // (only the last ring round can hold padding chunks: wave-uniform skip)
            park();
            if (d == 0) load_a(cur + W);
            multiply(st[d]);
        }
        cur += W;
    }

    // ---- in-block split-K reduction through LDS, then the reference's rounding chain (fixed summation order) ----
    __syncthreads();                   // the activation slots alias the reduction rows
    float(*red)[NR][64] = reinterpret_cast<float(*)[NR][64]>(lds);
#pragma unroll
    for (int r = 0; r < MT * NT; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][r * 4 + i][lane] = acc[r][i];
    __syncthreads();
    for (int r = wave; r < NR; r += W) {
        float v = 0.f;
        for (int w = 0; w < W; ++w) v += red[w][r][lane];
        const int a = r >> 2, i = r & 3, mt = a / NT, t = a - mt * NT;
        const int m = 16 * mt + 4 * rq + i, n = (tile0 + t) * kTileN + c;
        const bool live = m < p.M && n < p.N;
        if (GLUE != 0 && p.out_glue == kOutSiluMul) {
            // interleaved gate|up tile (fuse_gate_up_interleaved): lanes c < 8 hold gate columns, c >= 8 the matching up columns of
            // the SAME row (the reducer wave runs one (row tile, column tile, register) for all its lanes: the shuffle is legal)
            float y = round_through<ACT>(v);
            if (p.bias != nullptr && live) y = round_through<ACT>(y + load16_as_f32<ACT>(p.bias, (size_t)n));
            const float up = __shfl_down(y, 8, 64);
            const float a = round_through<ACT>(y / (1.0f + expf(-y))) * up;
            const int jn = (tile0 + t) * 8 + c;
            if (live && c < 8 && jn < p.N / 2) reinterpret_cast<uint16_t*>(p.out)[(size_t)m * (p.N / 2) + jn] = f32_to_16<ACT>(a);
        } else if (live) {
            if (p.out_f32) {
                reinterpret_cast<float*>(p.out)[(size_t)m * p.N + n] = v;
            } else {
                float y = round_through<ACT>(v);
                if (p.bias != nullptr) y = y + load16_as_f32<ACT>(p.bias, (size_t)n);
                reinterpret_cast<uint16_t*>(p.out)[(size_t)m * p.N + n] = f32_to_16<ACT>(y);
            }
        }
    }
}

// Column tiles per block: 4 where N / 64 >= 256 blocks still fill the chip (N >= 16384), else 2 (N >= 8192).  5..32 rows only: at
// 33..64 rows (measured with two tiles per block, profiles/r03_wide_layers.txt) the MFMA-tiled kernel is faster (28.7 vs 39.0 us on
// 4096x28672 at M = 48) and four tiles would not fit 256 VGPRs beside 64 accumulator + 64 activation registers.
constexpr int kWideMaxWaves = 8;    // (512-thread bound for every instantiation)
constexpr int kWideMaxM = 32;

template <int ACT, int SCL, int MT, bool HALFQ, int kWideNT>
static int launch_wide_gpc(const SkinnyParams& p, const SkinnyPlan& pl, hipStream_t stream) {
    const dim3 grid(ceil_div(p.N, kTileN) / kWideNT), block(64 * pl.waves);
    constexpr int NR = MT * kWideNT * 4;
    constexpr int kSlotA = 16 * MT * kRowsPitch * 16;
    constexpr int kSlot = kSlotA > NR * 256 ? kSlotA : NR * 256;
    const size_t lds_bytes = (size_t)pl.waves * kSlot + 64;
    if constexpr (MT == 1) {
        if (p.in_glue == kGlueRmsNorm) {     // decode op on 5..16 rows (one group constant per chunk: the ABI layer checks)
            hipLaunchKernelGGL((skinny_wide_kernel<ACT, SCL, MT, 1, HALFQ, kWideNT, kGlueRmsNorm>), grid, block, lds_bytes, stream, p);
            return check_hip(hipGetLastError(), "skinny_wide_kernel (decode glue) launch");
        }
    }
    if (pl.gpc == 1) {
        hipLaunchKernelGGL((skinny_wide_kernel<ACT, SCL, MT, 1, HALFQ, kWideNT>), grid, block, lds_bytes, stream, p);
    } else {
        hipLaunchKernelGGL((skinny_wide_kernel<ACT, SCL, MT, 4, HALFQ, kWideNT>), grid, block, lds_bytes, stream, p);
    }
    return check_hip(hipGetLastError(), "skinny_wide_kernel launch");
}

template <int ACT, int SCL, int NT>
static int launch_wide_nt(const SkinnyParams& p, const SkinnyPlan& pl, hipStream_t stream) {
    if (pl.mt == 1) return p.M <= 8 ? launch_wide_gpc<ACT, SCL, 1, true, NT>(p, pl, stream) : launch_wide_gpc<ACT, SCL, 1, false, NT>(p, pl, stream);
    if (pl.mt == 2) return p.M <= 24 ? launch_wide_gpc<ACT, SCL, 2, true, NT>(p, pl, stream) : launch_wide_gpc<ACT, SCL, 2, false, NT>(p, pl, stream);
    set_error("skinny_wide_kernel: at most %d rows", kWideMaxM);
    return -22;
}
template <int ACT, int SCL>
static int launch_wide(const SkinnyParams& p, const SkinnyPlan& pl, hipStream_t stream) {
    return pl.nt == 4 ? launch_wide_nt<ACT, SCL, 4>(p, pl, stream) : launch_wide_nt<ACT, SCL, 2>(p, pl, stream);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int BITS, int ACT, int SCL, int MT, int AM, int D, int LB = skinny_launch_bound<BITS, MT>()>
static int launch_skinny_gpc(const SkinnyParams& p, const SkinnyPlan& pl, hipStream_t stream) {
    if (64 * pl.waves > LB) {
        set_error("skinny_kernel: %d waves per block exceed the instantiation's %d-thread bound", pl.waves, LB);
        return -22;
    }
    const dim3 grid(ceil_div(p.N, kTileN), p.splits);
    const dim3 block(64 * pl.waves);
    constexpr int kSlot = slot_bytes<AM, MT>() > MT * 1024 ? slot_bytes<AM, MT>() : MT * 1024;
    const size_t lds_bytes = (size_t)pl.waves * kSlot + 16 + (AM == AM_ROW1P ? (size_t)p.K * 2 + 80 : 64);
    if constexpr ((AM == AM_ROW1 && MT == 1 && (D == 4 || D == 2)) || (AM == AM_ROW1P && MT == 1 && D == 4) || (AM == AM_ROW4 && MT == 1 && D == 4) ||
                  (is_rows<AM>() && MT == 1 && D == 2)) {
        if (p.in_glue != kGlueNone) {  // decode op with input glue: regular plans only (ABI checks)
            if (pl.gpc != 1 && AM == AM_ROW1P) {
                set_error("skinny_kernel: the in-kernel act-order variant needs group_size %% 128 == 0");
                return -22;
            }
            if (p.in_glue == kGlueRmsNorm) {
                if (pl.gpc == 1) {
                    hipLaunchKernelGGL((skinny_kernel<BITS, ACT, SCL, MT, 1, AM, D, kGlueRmsNorm>), grid, block, lds_bytes, stream, p);
                } else if constexpr (AM != AM_ROW1P) {   // group_size 32 / 64: a group constant per K-step
                    hipLaunchKernelGGL((skinny_kernel<BITS, ACT, SCL, MT, 4, AM, D, kGlueRmsNorm>), grid, block, lds_bytes, stream, p);
                }
            } else if constexpr (AM != AM_ROW4 && !is_rows<AM>()) {
                if (pl.gpc == 1) {
                    hipLaunchKernelGGL((skinny_kernel<BITS, ACT, SCL, MT, 1, AM, D, kGlueSiluMul>), grid, block, lds_bytes, stream, p);
                } else if constexpr (AM != AM_ROW1P) {
                    hipLaunchKernelGGL((skinny_kernel<BITS, ACT, SCL, MT, 4, AM, D, kGlueSiluMul>), grid, block, lds_bytes, stream, p);
                }
            } else {
                set_error("decode op: SiLU*mul INPUT glue exists for one row only (use the paired gate_up epilogue)");
                return -22;   // GPTQHIP_EINVAL (the ABI layer rejects this combination before it gets here)
            }
            return check_hip(hipGetLastError(), "skinny_kernel (decode glue) launch");
        }
    }
    if (pl.gpc == 1) {
        hipLaunchKernelGGL((skinny_kernel<BITS, ACT, SCL, MT, 1, AM, D, 0, LB>), grid, block, lds_bytes, stream, p);
    } else if constexpr (AM != AM_ROW1P) {
        hipLaunchKernelGGL((skinny_kernel<BITS, ACT, SCL, MT, 4, AM, D, 0, LB>), grid, block, lds_bytes, stream, p);
    } else {
        // (gptqhip_gemm / gptqhip_decode_linear only pick the in-kernel permutation for one group constant per chunk)
        set_error("skinny_kernel: the in-kernel act-order variant needs group_size %% 128 == 0");
        return -22;
    }
    return check_hip(hipGetLastError(), "skinny_kernel launch");
}

template <int BITS, int ACT, int SCL>
static int launch_skinny_mt(const SkinnyParams& p, const SkinnyPlan& pl, hipStream_t stream) {
    if (pl.mt == 1 && p.M == 1 && p.perm != nullptr) return launch_skinny_gpc<BITS, ACT, SCL, 1, AM_ROW1P, 4>(p, pl, stream);
    if (pl.mt == 1 && p.M == 1 && pl.depth == 2) return launch_skinny_gpc<BITS, ACT, SCL, 1, AM_ROW1, 2>(p, pl, stream);
    if (pl.mt == 1 && p.M == 1) return launch_skinny_gpc<BITS, ACT, SCL, 1, AM_ROW1, 4>(p, pl, stream);
    if (pl.mt == 1 && p.M <= 4 && pl.depth == 4) return launch_skinny_gpc<BITS, ACT, SCL, 1, AM_ROW4, 4>(p, pl, stream);
    if (pl.mt == 1 && p.M <= 8) return launch_skinny_gpc<BITS, ACT, SCL, 1, AM_ROWSH, 2>(p, pl, stream);
    if (pl.mt == 1) return launch_skinny_gpc<BITS, ACT, SCL, 1, AM_ROWS, 2>(p, pl, stream);
    if (pl.mt == 2 && p.M <= 24) return launch_skinny_gpc<BITS, ACT, SCL, 2, AM_ROWSH, 2>(p, pl, stream);
    if (pl.mt == 2) return launch_skinny_gpc<BITS, ACT, SCL, 2, AM_ROWS, 2>(p, pl, stream);
    if constexpr (BITS == 4) {   // 33..64 rows in one launch (4-bit only: the 8-bit register stage does not fit beside 64 activation registers)
        if (p.M <= 56) return launch_skinny_gpc<BITS, ACT, SCL, 4, AM_ROWSH, 2>(p, pl, stream);
        return launch_skinny_gpc<BITS, ACT, SCL, 4, AM_ROWS, 2>(p, pl, stream);   // (callers chunk M to <= 64 rows)
    } else {
        set_error("skinny_kernel: more than 32 rows per launch need 4-bit weights");
        return -22;
    }
}

SkinnyPlan plan_skinny(int M, int K, int N, int group_size, int force_split, int force_waves, bool in_kernel_perm, int bits, int allow_wide, int prefer_deep) {
    static const bool allow_depth2 = [] { const char* v = getenv("GPTQHIP_NO_DEPTH2"); return !(v && *v && *v != '0'); }();
    static const bool allow_pad = [] { const char* v = getenv("GPTQHIP_NO_PAD"); return !(v && *v && *v != '0'); }();   // A/B switch
    SkinnyPlan pl;
    const int mtiles = ceil_div(M, 16);
    pl.mt = mtiles <= 1 ? 1 : (mtiles <= 2 ? 2 : 4);  // gptqhip_gemm feeds at most 32 rows per launch (64 for 4-bit narrow layers)
    pl.gpc = (group_size % kChunkK == 0) ? 1 : 4;
    pl.chunks = ceil_div(K, kChunkK);
    const int tiles = ceil_div(N, kTileN);
    // waves per block: ~4 chunks per wave (= ring depth, all of a wave's loads in flight at once), at least
    // the 4*MT waves the LDS reduce mapping needs
    int waves = pl.chunks >= 64 ? 16 : (pl.chunks >= 32 ? 8 : 4);
    if (tiles <= 256 && pl.chunks >= 16) waves = 16;  // few tiles: one block per CU, go wide on K
    // prefer a wave count that gives every wave a multiple of the ring depth (regular pipeline, counted waits):
    // e.g. K=14336 -> 112 chunks -> 14 waves x 8 chunks
    pl.depth = (pl.mt == 1 && M <= 4) ? 4 : 2;
    bool chosen = false;
    if (pl.mt == 1) {
        // (measured in round 2, profiles/r02_waves_sweep.txt, r02_batch_decode_sweep.txt): what matters is the number of waves on the chip, not
        // per block -- about 16 per CU (4096 in flight), each with a long K run (>= 2 ring rounds): fewer, longer-lived blocks
        // amortise the ramp, the statistics / reduce prologue and the tail.  Llama-3-70B gate_up (3584 tiles, K=8192): 16 waves
        // per block 61 us, 4 waves 47 us; qkv 15.5 -> 12.8 (8 waves); down (K=28672) 29.5 -> 27.0 (8 waves).
        // (the preload form: A/B switch GPTQHIP_SK1_WAVE_TARGET = waves on the chip the plan aims for)
        static const int deep_target = [] { const char* v = getenv("GPTQHIP_SK1_WAVE_TARGET"); return (v && *v) ? atoi(v) : 4096; }();
        int target = (prefer_deep ? deep_target : 4096) / (tiles > 0 ? tiles : 1);
        // between one and two blocks per CU (Llama-3-8B q|k|v: 384 tiles) the raw-code form (20 VALU per chunk) does better with half the waves
        // per block and two ring rounds each -- 4 x 8 chunks 5.28-5.29 us vs 8 x 4 chunks 5.45-5.49 (profiles/r06_decode_forms.txt); a chip-wide
        // target of 2048 waves instead loses on the long-K layers (down_proj 7.63 -> 7.82, 70B 22.99 -> 24.79), and the bit-faithful bf16 form
        // (more VALU per chunk) loses on q|k|v too (6.25 -> 6.42), so only this band of this form takes it
        if (prefer_deep == 2 && tiles > 256 && tiles < 512 && deep_target == 4096) target = 2048 / tiles;
        target = target < 4 ? 4 : (target > 16 ? 16 : target);
        int best = 0, best_depth = pl.depth;
        // batch 1 may also run a 2-deep ring (16 waves x 2 chunks on K = 4096: twice the waves dequantise the same bytes
        // once they have landed -- narrow layers like o_proj are latency-, not stream-bound)
        const int force_s = force_split > 0 ? (force_split < pl.chunks ? force_split : pl.chunks) : 0;
        const int depth_hi = pl.depth;                                        // the kernel variants: D = 4 up to 4 rows, else 2
        // 2..4 rows on a SHORT K (fewer than 16 chunks, e.g. the o_proj row shard of Llama-3-70B at TP = 8: K = 1024): no wave count
        // gives whole rounds of the 4-deep ring, so those rows may ride the 2-deep pipeline of the 5..8-row variant (round 4: before,
        // the shape fell off the regular pipeline and gptqhip_decode_supported said no -- found by tests/test_gpu_tp8_shapes.py)
        const bool short_k_rows = M >= 2 && M <= 4 && pl.chunks < 16 && !in_kernel_perm;
        // (prefer_deep: the preload form, skinny1_kernel -- 8 waves x 4 chunks beat 16 x 2 on 4096^2: 4.39 vs 4.68 us, profiles/r06_decode_forms.txt)
        const int depth_lo = ((M == 1 && allow_depth2 && !in_kernel_perm && !(prefer_deep && pl.chunks >= 32)) || short_k_rows) ? 2 : pl.depth;
        // pass 0: exact plans (every wave's chunks are whole ring rounds); pass 1: plans whose LAST round carries padding chunks
        // (clamped loads, skipped compute -- see Cursor) for chunk counts with awkward factors (Llama-2 down_proj: 86 = 2 * 43,
        // Qwen2-7B: 148 = 4 * 37), accepted up to 1/8 of wasted loads, least waste first
        float best_score = 0.f;
        for (int pass = 0; pass < (allow_pad ? 2 : 1) && best == 0; ++pass) {
            for (int depth = depth_hi; depth >= depth_lo; depth -= 2) {
                for (int w = 4; w <= 16; ++w) {
                    // the candidate must stay on the regular pipeline AFTER the cross-block split-K decision below (narrow layers)
                    int sp = 1;
                    if (tiles < 48 && pl.chunks >= 4 * w) sp = ceil_div(48, tiles);
                    const int max_sp = pl.chunks / w < 1 ? 1 : pl.chunks / w;
                    if (sp > max_sp) sp = max_sp;
                    if (force_s > 0) sp = force_s;
                    const int cps = ceil_div(pl.chunks, sp);
                    const int nsp = ceil_div(pl.chunks, cps);
                    const int virt = ceil_div(cps, w * depth) * w * depth * nsp;
                    const int waste = virt - pl.chunks;
                    // (a SHORT K -- fewer than 16 chunks, e.g. the 512-row o_proj shard of an 8B model at TP = 8 -- cannot fill whole
                    // rounds of four or more waves: there any padding is accepted, it is at most one round of L2-hit loads)
                    if (pass == 0 ? waste != 0 : (waste * 8 > pl.chunks && (pl.chunks >= 16 || in_kernel_perm))) continue;
                    // pass 1 trades wasted loads against the distance from the wave target: one wave off ~ 1 % of extra loads
                    const float score = (float)waste / (float)pl.chunks + 0.01f * (float)abs(w - target);
                    bool better;
                    if (best == 0) {
                        better = true;
                    } else if (pass == 1) {
                        better = score < best_score;
                    } else {
                        better = abs(w - target) < abs(best - target) || (abs(w - target) == abs(best - target) && depth == best_depth && w > best);
                    }
                    if (better) {
                        best = w;
                        best_depth = depth;
                        best_score = score;
                    }
                }
            }
        }
        if (best > 0) {
            waves = best;
            pl.depth = best_depth;
            chosen = true;
        }
    }
    if (!chosen) {
        int best = 0;
        for (int w = 16; w >= 4; --w) {
            if (pl.chunks % (w * pl.depth) == 0 && w >= waves / 2) { best = w; break; }
        }
        if (best > 0 && (best >= waves || pl.chunks / (waves * pl.depth) * (waves * pl.depth) != pl.chunks)) waves = best;
    }
    if (pl.mt >= 2) {
        // 17..64 rows (split-ring pipeline: the activation stage is one deep).  Round 2 ran 16 waves per block with two activation
        // stages in registers (1024-thread bound = 128 VGPRs: every instantiation spilled 24-60 registers); the waves per block are
        // now bounded by the instantiation's launch bound and the plan is the largest count that gives whole ring rounds
        const int cap = skinny_max_waves(bits, pl.mt);
        waves = cap < 8 ? cap : 8;
        for (int w = cap; w >= 4; --w) {
            if (pl.chunks % (w * pl.depth) == 0) { waves = w; break; }
        }
    }
    const int min_waves = pl.mt == 4 ? 4 : 4 * pl.mt;   // (MT == 4: reducer waves loop over the accumulator registers)
    if (waves < min_waves) waves = min_waves;
    if (force_waves > 0) waves = force_waves < min_waves ? min_waves : force_waves;
    if (pl.mt >= 2 && waves > skinny_max_waves(bits, pl.mt)) waves = skinny_max_waves(bits, pl.mt);
    pl.waves = waves;
    // cross-block split-K costs a publish + ticket + re-read round trip (~1.5-2 us measured): only worth it
    // when the tiles alone leave most of the chip idle AND there is a long K range to share
    const int target_blocks = 48;
    int s = 1;
    if (tiles < target_blocks && pl.chunks >= 4 * waves) s = ceil_div(target_blocks, tiles);
    int max_s = pl.chunks / waves;  // keep >= 1 chunk per wave
    if (max_s < 1) max_s = 1;
    if (s > max_s) s = max_s;
    if (force_split > 0) s = force_split < pl.chunks ? force_split : pl.chunks;
    if (pl.mt == 4) s = 1;   // no cross-block split-K for the 33..64-row instantiations (their epilogue is the plain one)
    pl.chunks_per_split = ceil_div(pl.chunks, s);
    pl.splits = ceil_div(pl.chunks, pl.chunks_per_split);
    pl.slab_floats = pl.splits > 1 ? (size_t)pl.splits * M * N : 0;
    pl.rounds = ceil_div(pl.chunks_per_split, pl.waves * pl.depth);
    const int virt_chunks = pl.rounds * pl.waves * pl.depth * pl.splits;   // incl. the padding of every block's last ring round
    pl.regular = (allow_pad ? ((virt_chunks - pl.chunks) * 8 <= pl.chunks || (pl.chunks < 16 && pl.mt == 1 && pl.splits == 1 && !in_kernel_perm)) : virt_chunks == pl.chunks) &&
                         K % kChunkK == 0 &&
                         (pl.gpc == 1 ? (group_size >= K || ((group_size / kChunkK) & (group_size / kChunkK - 1)) == 0)
                                      : (group_size == 32 || group_size == 64))
                     ? 1
                     : 0;
    // wide layers at 5..64 rows (skinny_wide_kernel: four column tiles per block share one staging of the activation tile).  Needs
    // the regular pipeline with a two-deep weight ring, 4-bit weights, whole groups of four tiles and enough of them to fill the chip
    // without a cross-block split; the caller says whether its epilogue is the plain one (allow_wide: gptqhip_gemm yes, decode op no)
    static const bool wide_off = [] { const char* v = getenv("GPTQHIP_NO_WIDE"); return v && *v && *v != '0'; }();   // A/B switch
    static const int wide_min_blocks = [] { const char* v = getenv("GPTQHIP_WIDE_MIN_BLOCKS"); return (v && *v) ? atoi(v) : 192; }();   // A/B switch; 192: measured, profiles/r03_wide_layers.txt
    const int kWideNT = (tiles % 4 == 0 && tiles / 4 >= wide_min_blocks) ? 4 : 2;
    // from 5 rows; the decode op WITH glue (allow_wide == 2) on four-tile layers from 2 rows (gate_up with RMSNorm + paired SiLU at
    // 2 / 4 rows: 18.8 -> 17.8 / 20.2 -> 18.3 us; the plain product and two-tile layers lose below 5: profiles/r03_wide_layers.txt)
    static const int wide_min_m_env = [] { const char* v = getenv("GPTQHIP_WIDE_MIN_M"); return (v && *v) ? atoi(v) : 0; }();   // A/B switch
    const int wide_min_m = wide_min_m_env > 0 ? wide_min_m_env : ((allow_wide == 2 && kWideNT == 4) ? 2 : 5);
    if (allow_wide && !wide_off && bits == 4 && M >= wide_min_m && M <= kWideMaxM && !in_kernel_perm && tiles % kWideNT == 0 &&
        tiles / kWideNT >= wide_min_blocks && force_split <= 1 && K % kChunkK == 0 &&
        (pl.gpc == 1 ? (group_size >= K || ((group_size / kChunkK) & (group_size / kChunkK - 1)) == 0) : (group_size == 32 || group_size == 64))) {
        const int cap = force_waves > 0 && force_waves < kWideMaxWaves ? force_waves : kWideMaxWaves;
        int best = 0;
        for (int w = cap; w >= 4 && best == 0; --w) {
            const int virt = ceil_div(pl.chunks, w * 2) * w * 2;
            if ((virt - pl.chunks) * 8 <= pl.chunks) best = w;      // whole ring rounds, at most 1/8 padding chunks
        }
        if (best > 0) {
            pl.nt = kWideNT;
            pl.waves = best;
            pl.depth = 2;
            pl.splits = 1;
            pl.chunks_per_split = pl.chunks;
            pl.slab_floats = 0;
            pl.rounds = ceil_div(pl.chunks, best * 2);
            pl.regular = 1;
        }
    }
    return pl;
}

int launch_skinny(const GemmArgs& a, const SkinnyPlan& pl, float* slabs, int* counters, hipStream_t stream) {
    SkinnyParams p;
    p.x = a.x;
    p.perm = a.perm;
    p.qw = a.qweight;
    p.meta = a.meta;
    p.bias = a.bias;
    p.out = a.out;
    p.slabs = slabs;
    p.counters = counters;
    p.M = a.M;
    p.K = a.K;
    p.N = a.N;
    p.G = a.K / a.group_size;
    p.group_size = a.group_size;
    p.chunks = pl.chunks;
    p.chunks_per_split = pl.chunks_per_split;
    p.splits = pl.splits;
    p.out_f32 = a.out_f32;
    p.regular = pl.regular;
    p.n_mine = pl.rounds * pl.depth;
    p.cpg_shift = -1;
    if (a.group_size >= a.K) {
        p.cpg_shift = 20;   // one group for the whole K (group_size = -1 checkpoints): every chunk index >> 20 is group 0
    } else if (a.group_size % kChunkK == 0) {
        const int cpg = a.group_size / kChunkK;
        if ((cpg & (cpg - 1)) == 0) {
            int sh = 0;
            while ((1 << sh) < cpg) ++sh;
            p.cpg_shift = sh;
        }
    } else if (pl.regular) {
        p.cpg_shift = a.group_size == 64 ? 1 : 0;   // GPC == 4 on the regular pipeline: K-steps (32 rows) per group, log2
    }
    p.exact_bf16 = a.exact_bf16;
    p.alg_fp16 = a.alg_fp16;
    p.glue_b = a.glue_b;
    p.residual = a.residual;
    p.eps = a.eps;
    p.in_glue = a.in_glue;
    p.stats_in = a.stats_in;
    p.stats_out = a.stats_out;
    p.out_glue = a.out_glue;
    p.stats_n = a.stats_n;
    if (pl.nt > 1) {
        const bool glue_ok = a.in_glue == kGlueNone ? a.out_glue == kOutNone
                                                    : (a.in_glue == kGlueRmsNorm && pl.mt == 1 && pl.gpc == 1 && (a.out_glue == kOutNone || a.out_glue == kOutSiluMul));
        if (a.bits != 4 || !glue_ok || a.residual != nullptr || a.stats_out != nullptr || a.perm != nullptr || (a.out_glue == kOutSiluMul && a.out_f32)) {
            set_error("skinny_wide_kernel: planned for a call it cannot serve (glue / permutation / 8-bit)");
            return -22;
        }
        if (a.group_size % kChunkK != 0) p.cpg_shift = a.group_size == 64 ? 1 : 0;      // K-steps (32 rows) per group, log2
        if (a.act_dtype == kFP16 && a.scale_dtype == kFP16) return launch_wide<kFP16, kFP16>(p, pl, stream);
        if (a.act_dtype == kBF16 && a.scale_dtype == kFP16) return launch_wide<kBF16, kFP16>(p, pl, stream);
        if (a.act_dtype == kFP16 && a.scale_dtype == kBF16) return launch_wide<kFP16, kBF16>(p, pl, stream);
        return launch_wide<kBF16, kBF16>(p, pl, stream);
    }
    if (a.preload && a.bits == 4) {
        bool served = false;
        int rc = 0;
        if (a.act_dtype == kFP16 && a.scale_dtype == kFP16) rc = launch_skinny1<kFP16, kFP16>(p, pl, a.alg_fp16, stream, &served);
        else if (a.act_dtype == kBF16 && a.scale_dtype == kFP16) rc = launch_skinny1<kBF16, kFP16>(p, pl, a.alg_fp16, stream, &served);
        else if (a.act_dtype == kFP16 && a.scale_dtype == kBF16) rc = launch_skinny1<kFP16, kBF16>(p, pl, a.alg_fp16, stream, &served);
        else rc = launch_skinny1<kBF16, kBF16>(p, pl, a.alg_fp16, stream, &served);
        if (served) return rc;
    }
#define GPTQHIP_DISPATCH(B, A_, S_) return launch_skinny_mt<B, A_, S_>(p, pl, stream)
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
