// Prefill kernel, host side: planner, launcher, split-K reduce, and the 4-bit / 16-bit-output instantiations of the kernel
// template (gptqhip_tiled_kernel.h; fp32-output ones in gptqhip_tiled_f32.hip, 8-bit ones in gptqhip_tiled8.hip).
#include "gptqhip_tiled_kernel.h"

namespace gptqhip {

int launch_tiled_w4(const TiledParams& p, int act_dtype, int scale_dtype, int gpc, int bm, hipStream_t stream) {
    return launch_tiled_bits<4, 0>(p, act_dtype, scale_dtype, gpc, bm, stream);
}

// Sum the split-K slabs in a fixed order (deterministic), then the reference's rounding chain.  One thread per 4
// consecutive columns (16-byte slab loads).  SP > 0: the split count is a compile-time constant and all slab loads (and the
// bias) are requested before the first addition; SP == 0: any split count, eight loads in flight.  The additions stay in slab
// order either way.  The kernel takes 4.5-4.8 us for 2..8 slabs of a 128 x 4096 output whatever the load order (measured both
// ways, profiles/r03_midm_trace.txt): it is bound by reading slabs that the producing blocks on OTHER XCDs wrote (the L2s are
// per XCD and written back at the kernel boundary), i.e. 17 MB through the memory side at ~4 TB/s, not by latency.
// Slab loads: non-temporal for up to four slabs (round 5 A/B, profiles/r05_slab_store_policy.txt: 4096^2 at M = 128 with 3 slabs 14.0 -> 13.7 us per
// call, 4096x6144 at M = 72 with 4 slabs 14.1 -> 12.7, the other 3- / 4-slab cases +-1 %), plain above (8 slabs: 0..+3 % slower with nt).
template <bool NT>
__device__ __forceinline__ f4_t slab_load(const float* p) {
    if constexpr (NT) {
        return __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p));
    } else {
        return *reinterpret_cast<const f4_t*>(p);
    }
}
#define SLAB_LOAD(p_) slab_load<(SP > 0 && SP <= 4)>(p_)
template <int ACT, int SP>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, const void* __restrict__ bias,
                                                            void* __restrict__ out, int M, int N, int ldo, int splits, int out_f32) {
    const size_t quads = (size_t)M * N / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= quads) return;
    const size_t stride = (size_t)M * N;
    const float* src = slabs + 4 * i;
    const int row = (int)((4 * i) / (size_t)N);
    const int n = (int)(4 * i - (size_t)row * N);
    const size_t o = (size_t)row * (size_t)ldo + n;  // N % 4 == 0: the quad stays inside one row
    u2_t bq = {0u, 0u};
    if (bias != nullptr && !out_f32) bq = *reinterpret_cast<const u2_t*>(reinterpret_cast<const uint16_t*>(bias) + n);
    f4_t s;
    if constexpr (SP > 0) {
        f4_t v[SP];
#pragma unroll
        for (int sp = 0; sp < SP; ++sp) v[sp] = SLAB_LOAD(src + (size_t)sp * stride);
        s = v[0];
#pragma unroll
        for (int sp = 1; sp < SP; ++sp) s += v[sp];
    } else {
        s = SLAB_LOAD(src);
        int sp = 1;
        for (; sp + 8 <= splits; sp += 8) {
            f4_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = SLAB_LOAD(src + (size_t)(sp + j) * stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; sp < splits; ++sp) s += SLAB_LOAD(src + (size_t)sp * stride);
    }
    if (out_f32) {
        *reinterpret_cast<f4_t*>(reinterpret_cast<float*>(out) + o) = s;
        return;
    }
    uint16_t r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float y = round_through<ACT>(s[j]);
        if (bias != nullptr) y = y + bits16_to_f32<ACT>((uint16_t)((j < 2 ? bq.x : bq.y) >> (16 * (j & 1))));
        r[j] = f32_to_16<ACT>(y);
    }
    u2_t ov;
    ov.x = (uint32_t)r[0] | ((uint32_t)r[1] << 16);
    ov.y = (uint32_t)r[2] | ((uint32_t)r[3] << 16);
    *reinterpret_cast<u2_t*>(reinterpret_cast<uint16_t*>(out) + o) = ov;
}

// Launch model of the prefill kernel for grids of at most one round of 256-row tiles, microseconds (main kernel + split-K reduce):
//   main = a + b f + r cps (c + d f)      f = share of the 256 CUs with a block, cps = 128-row chunks per block, r = rounds of the
//                                          persistent tile loop (a partial last round counts 0.75 + 0.25 rem/256: it clocks higher)
//   reduce = max(4.6, 1.5 + slab bytes / 6.2 TB/s)
// a..d per tile height, least squares over 926 timed (shape, M, tile height, split) points on one MI355X (tests/dev/tiled_plan_check.py,
// tiled_model_check.py; 10 layer shapes of Llama-2/3 7B..70B, M = 96..2048): mean error 3 %, 90th percentile 6 %.  It only has to RANK
// the candidates; measured against the round-2 rules on the same box it wins 5-32 % on 50 of 160 points and loses > 4 % on 3.
// Round 5: refitted per tile height (now 32 .. 128 rows in steps of 16, and 256) on graph-replayed launches over rotating cold weights
// (tests/dev/midm_heights.py -> midm_fit.py, profiles/r05_midm_heights_sweep*.txt), with one shared term for grids that leave more
// than a quarter of the CUs idle: + 10.5 us x max(0, 0.75 - f); and, in the same fit, a second coefficient set for the 128-column-block
// form of the kernel (one column tile per wave, gptqhip_tiled_n128_r<rows>.hip).  2703 averaged points of six runs (M = 64..2048): relative
// fit error 5.1 % mean / 11 % p90 (the run-to-run spread of a point is +-5 %); replaying the search on them the model's pick is 1.5 % (mean)
// behind the best measured point, where the round-4 planner's pick (256-column blocks, 64 / 128 / 256 rows) is 8.7 % behind.
static double tiled_cost_us(int M, int K, int N, int bm, int s, int bn = kTiledBN) {
    // rows 0..7: 256-column blocks, tile heights 32 .. 128, 256; rows 8..14: 128-column blocks (one column tile per wave), heights 32 .. 128
    static const int kHeights[8] = {32, 48, 64, 80, 96, 112, 128, 256};
    static const double kCoef[15][4] = {{-9.980, 14.492, 1.077, -0.152}, {-10.387, 16.491, 1.187, -0.159}, {-7.283, 14.738, 1.154, -0.029},
                                        {-5.225, 13.447, 1.250, -0.035}, {-5.039, 13.967, 1.425, -0.035},  {-5.134, 14.638, 1.611, -0.083},
                                        {-4.449, 15.450, 1.657, -0.030}, {-3.566, 19.111, 2.788, 0.015},
                                        {-2.380, 3.950, 0.474, 0.030},   {-3.416, 5.773, 0.540, 0.036},    {-4.522, 8.763, 0.577, 0.054},
                                        {-5.342, 10.944, 0.696, -0.013}, {-5.464, 11.440, 0.795, 0.001},   {-5.543, 12.336, 0.908, -0.084},
                                        {-4.921, 11.782, 0.962, -0.008}};
    const double kIdle = 10.819;
    int hi = 7;
    for (int i = 0; i < 8; ++i)
        if (kHeights[i] == bm) hi = i;
    const double* co = kCoef[bn == 128 ? 8 + (hi < 7 ? hi : 6) : hi];
    const int chunks = ceil_div(K, kChunkK), cps = ceil_div(chunks, s), s_eff = ceil_div(chunks, cps);
    const long tiles = (long)ceil_div(N, bn) * ceil_div(M, bm), blocks = tiles * s_eff;
    double r, f;
    if (s_eff == 1) {
        const long full = tiles / 256, rem = tiles % 256;
        r = (double)full + (rem ? 0.75 + 0.25 * (double)rem / 256.0 : 0.0);
        f = full >= 1 ? 1.0 : (double)rem / 256.0;
    } else {
        r = blocks <= 256 ? 1.0 : (double)blocks / 256.0;
        f = blocks <= 256 ? (double)blocks / 256.0 : 1.0;
    }
    // (the fit has negative intercepts: clamp at the cost of one chunk pass so that short-K plans -- outside the 4-bit / g128 / M = 64 .. 2048 sweeps
    // that calibrated it -- cannot come out free or negative)
    double main_us = co[0] + co[1] * f + r * cps * (co[2] + co[3] * f) + kIdle * (f < 0.75 ? 0.75 - f : 0.0);
    if (main_us < 1.0 + 0.5 * r * cps) main_us = 1.0 + 0.5 * r * cps;
    const double slab_mb = (double)s_eff * M * N * 4.0 / 1e6;
    const double reduce_us = s_eff > 1 ? (slab_mb / 6.2 + 1.5 > 4.6 ? slab_mb / 6.2 + 1.5 : 4.6) : 0.0;
    return main_us + reduce_us;
}

constexpr int kN128MaxRows = 1024;  // 128-column blocks and the in-between tile heights up to this many rows (the calibration sweeps' range)

TiledPlan plan_tiled(int M, int K, int N, int group_size, int bits, int force_variant, int force_split) {
    TiledPlan pl;
    pl.gpc = (group_size % kChunkK == 0) ? 1 : 4;
    // Tile height from a small measured cost model (unit = one full round of 256 blocks with 256-row tiles on 256 CUs).
    // The chip is power-bound in this kernel: a partly filled round runs at higher clocks, partial(b) ~ 0.45 + 0.55 b/256
    // (K=4096: 128 / 192 / 256 blocks take 81 / 99 / 113 us), and a round of 128-row tiles costs ~0.56.  Candidates:
    // all 256-row, all 128-row, or 256-row tiles for the full rounds plus ONE launch of 128-row tiles for the block
    // columns of a last round that would be at most half full (M=4096, N=6144: 207 / 198 / 186 us).
    const int cus = 256;
    const double kHalf = 0.56;  // one round of 128-row tiles relative to one round of 256-row tiles
    const int nbx = ceil_div(N, kTiledBN), nby256 = ceil_div(M, 256), nby128 = ceil_div(M, 128);
    auto cost = [&](long blocks, double unit) {
        const long full = blocks / cus, rem = blocks % cus;
        return unit * ((double)full + (rem ? 0.45 + 0.55 * (double)rem / cus : 0.0));
    };
    const long blocks256 = (long)nbx * nby256, blocks128 = (long)nbx * nby128;
    const double t256 = cost(blocks256, 1.0), t128 = cost(blocks128, kHalf);
    pl.bm = (blocks256 > 128 && t256 <= t128) ? 256 : 128;
    pl.tail_cols = 0;
    if (blocks256 > cus && force_variant == 0 && force_split == 0) {
        const int rem = (int)(blocks256 % cus);
        const int nbx_b = rem > 0 && rem <= cus / 2 ? ceil_div(rem, nby256) : 0;
        if (nbx_b > 0 && nbx_b < nbx) {
            const double ttail = cost((long)(nbx - nbx_b) * nby256, 1.0) + cost((long)nbx_b * nby128, kHalf) + 0.03;
            if (ttail < t256 && ttail < t128) {
                pl.bm = 256;
                pl.tail_cols = nbx_b;
            }
        }
    }
    if (force_variant == 1) pl.bm = 256;
    if (force_variant == 2) pl.bm = 128;
    // (round 1 confined 8-bit weights and group sizes 32/64 to 128-row tiles because their register stages spill beside 128
    // accumulators; measured in round 2 they are 19-36 % faster on 256-row tiles anyway -- gptqhip_tiled_kernel.h)
    // few rows: 64-row tiles (half the per-chunk work of a 128-row tile, twice the blocks ahead of the split-K decision).
    // Round-2 sweep (profiles/r02_tiled_plan_sweep.txt): M=128 4096^2 23.7 -> 17.0 us, M=256 29.4 -> 21.0 us; beyond 128 rows only
    // while the column count keeps the grid small (M=256, N=28672: 59 us on 128-row tiles vs 77 us)
    if ((M <= 128 || (M <= 256 && N <= 4608)) && force_variant == 0) {
        pl.bm = 64;
        pl.tail_cols = 0;
    }
    if (force_variant == 3) pl.bm = 64;
    // the extra heights exist for 4-bit weights with one group constant per chunk (gptqhip_tiled_r<rows>.hip)
    const bool any16 = bits == 4 && pl.gpc == 1;
    if (force_variant >= 32 && any16 && force_variant <= 128 && force_variant % 16 == 0) pl.bm = force_variant;
    // (dev / tests: 1000 + rows = 128-column blocks with that tile height)
    if (force_variant >= 1032 && force_variant <= 1128 && force_variant % 16 == 8 && any16) {
        pl.bm = force_variant - 1000;
        pl.bn = 128;
    }
    // split K across blocks when the (M, N) grid alone leaves most CUs idle (mid-size M, or K-heavy layers):
    // fp32 partial slabs + a tiny reduce kernel (a kernel boundary is cheaper than re-reading 100s of KB of slabs
    // through a last-arriver block -- MI355X_MICROARCH.md "handoff-payload")
    const int chunks = ceil_div(K, kChunkK);
    const long blocks = (long)ceil_div(M, pl.bm) * ceil_div(N, pl.bn);
    int s = 1;
    if (blocks <= 128 && chunks >= 8) {
        s = (int)(256 / blocks);  // measured: splitting grids that already have > 128 blocks loses to the reduce pass
        if (s > chunks / 4) s = chunks / 4;           // at least 4 chunks (512 rows of K) per block
        const size_t cap_floats = (size_t)16 << 20;  // 64 MiB of slabs at most
        while (s > 1 && (size_t)s * M * N > cap_floats) --s;
        if (s < 1) s = 1;
    }
    if (force_split > 0) s = force_split < chunks ? force_split : chunks;
    if (force_variant == 0 && force_split == 0 && blocks256 <= cus) {
        // at most ONE round of 256-row tiles (serving batches, short prefills, K-heavy layers): tile height and split factor together
        // from the measured launch model below instead of the two rules above -- those picked 128-row tiles + 5 splits where 64-row
        // tiles + 3 are 15-20 % faster (4096x6144 at M=160..320), never combined 256-row tiles with split-K (14336x4096 at M=1280:
        // 174 -> 137 us, 8192x10240 at M=448: 101 -> 89) and kept 64-row tiles on very wide layers (8192x57344 at M<=128: 148 -> 113 us).
        double best = 1e30;
        for (int bn : {256, 128}) {
            // (128-column blocks: 4-bit / one constant per chunk, up to 1024 rows, like the in-between heights)
            if (bn == 128 && (!any16 || M > kN128MaxRows)) continue;
            const int nbx_c = ceil_div(N, bn);
            for (int bm : {32, 48, 64, 80, 96, 112, 128, 256}) {
                if (bm <= 64 && M > 1024) continue;
                if (bn == 128 && bm == 256) continue;
                // (the in-between heights: 4-bit / one constant per chunk, and up to 1024 rows -- beyond, whole 64- / 128- / 256-row tiles
                // waste little and the sweeps that calibrated the model end; 128-column blocks above 512 rows were swept at 64 / 96 / 128 rows)
                if ((!any16 || M > kN128MaxRows) && bm != 64 && bm != 128 && bm != 256) continue;
                if (bn == 128 && M > 512 && bm != 64 && bm != 96 && bm != 128) continue;
                const long tiles = (long)nbx_c * ceil_div(M, bm);
                for (int sc = 1; sc <= 16; ++sc) {
                    if (sc > 1 && (sc > chunks / 4 || tiles * sc > cus || (size_t)sc * M * N > ((size_t)16 << 20))) break;
                    // (blocks of four chunks run ~5 % behind the model -- its residual by chunks per block, tests/dev/midm_fit.py -- and lose to
                    // five-chunk blocks on every shape of the sweep: not a candidate while K allows five)
                    if (sc > 1 && chunks >= 20 && ceil_div(chunks, sc) < 5) continue;
                    const double t = tiled_cost_us(M, K, N, bm, sc, bn);
                    if (t < best) {
                        best = t;
                        pl.bm = bm;
                        pl.bn = bn;
                        s = sc;
                    }
                }
            }
        }
        pl.tail_cols = 0;
    }
    pl.chunks_per_split = ceil_div(chunks, s);
    pl.splits = ceil_div(chunks, pl.chunks_per_split);
    pl.slab_floats = pl.splits > 1 ? (size_t)pl.splits * M * N : 0;
    return pl;
}

int launch_tiled(const GemmArgs& a, const TiledPlan& pl, float* slabs, hipStream_t stream) {
    TiledParams p;
    p.x = a.x;
    p.qw = a.qweight;
    p.meta = a.meta;
    p.bias = a.bias;
    p.out = a.out;
    p.M = a.M;
    p.K = a.K;
    p.N = a.N;
    p.ldo = a.ldo > 0 ? a.ldo : a.N;
    p.G = a.K / a.group_size;
    p.group_size = a.group_size;
    p.chunks = ceil_div(a.K, kChunkK);
    p.tiles = ceil_div(a.N, kTileN);
    p.out_f32 = a.out_f32;
    p.splits = pl.splits;
    p.chunks_per_split = pl.chunks_per_split;
    p.slabs = slabs;
    p.cpg_shift = -1;
    if (a.group_size % kChunkK == 0) {
        const int cpg = a.group_size / kChunkK;
        if ((cpg & (cpg - 1)) == 0) {
            int sh = 0;
            while ((1 << sh) < cpg) ++sh;
            p.cpg_shift = sh;
        }
    }
    // fp32 epilogue (split-K slabs, tensor-parallel partial sums) or the 16-bit rounding epilogue
    const bool f32 = p.splits > 1 || p.out_f32;
    int rc_main;
    if (pl.bn == 128) {
        switch (pl.bm) {
            case 32: rc_main = launch_tiled_w4_n128_r32(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 48: rc_main = launch_tiled_w4_n128_r48(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 64: rc_main = launch_tiled_w4_n128_r64(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 80: rc_main = launch_tiled_w4_n128_r80(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 96: rc_main = launch_tiled_w4_n128_r96(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 112: rc_main = launch_tiled_w4_n128_r112(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 128: rc_main = launch_tiled_w4_n128_r128(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            default: set_error("tiled kernel: no %d-row tile with 128-column blocks", pl.bm); return -22;
        }
    } else if (a.bits == 4 && pl.gpc == 1 && pl.bm != 64 && pl.bm != 128 && pl.bm != 256) {
        switch (pl.bm) {
            case 32: rc_main = launch_tiled_w4_r32(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 48: rc_main = launch_tiled_w4_r48(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 80: rc_main = launch_tiled_w4_r80(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 96: rc_main = launch_tiled_w4_r96(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            case 112: rc_main = launch_tiled_w4_r112(p, a.act_dtype, a.scale_dtype, f32 ? 1 : 0, stream); break;
            default: set_error("tiled kernel: no %d-row tile", pl.bm); return -22;
        }
    } else {
        rc_main = a.bits != 4 ? launch_tiled_w8(p, a.act_dtype, a.scale_dtype, pl.gpc, pl.bm, stream)
                  : f32   ? launch_tiled_w4_f32(p, a.act_dtype, a.scale_dtype, pl.gpc, pl.bm, stream)
                          : launch_tiled_w4(p, a.act_dtype, a.scale_dtype, pl.gpc, pl.bm, stream);
    }
    if (rc_main != 0 || pl.splits <= 1) return rc_main;
    const size_t quads = (size_t)a.M * a.N / 4;
    const dim3 grid((unsigned)((quads + 255) / 256));
#define GPTQHIP_REDUCE(SP_)                                                                                                          \
    do {                                                                                                                              \
        if (a.act_dtype == kFP16)                                                                                                     \
            hipLaunchKernelGGL((splitk_reduce_kernel<kFP16, SP_>), grid, dim3(256), 0, stream, slabs, a.bias, a.out, a.M, a.N, p.ldo, \
                               pl.splits, a.out_f32);                                                                                 \
        else                                                                                                                          \
            hipLaunchKernelGGL((splitk_reduce_kernel<kBF16, SP_>), grid, dim3(256), 0, stream, slabs, a.bias, a.out, a.M, a.N, p.ldo, \
                               pl.splits, a.out_f32);                                                                                 \
    } while (0)
    switch (pl.splits) {
        case 2: GPTQHIP_REDUCE(2); break;
        case 3: GPTQHIP_REDUCE(3); break;
        case 4: GPTQHIP_REDUCE(4); break;
        case 5: GPTQHIP_REDUCE(5); break;
        case 6: GPTQHIP_REDUCE(6); break;
        case 7: GPTQHIP_REDUCE(7); break;
        case 8: GPTQHIP_REDUCE(8); break;
        default: GPTQHIP_REDUCE(0); break;
    }
#undef GPTQHIP_REDUCE
    return check_hip(hipGetLastError(), "splitk_reduce_kernel launch");
}

}  // namespace gptqhip
