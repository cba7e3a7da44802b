// Host-side glue shared by the translation units of libgptqhip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace gptqhip {

constexpr int kTiledBN = 256;  // columns per block of the prefill kernel = WAVES x TPW x 16

struct GemmArgs {
    const void* x;
    const uint32_t* qweight;  // tile-major words (gptqhip_device.h)
    const uint32_t* meta;     // [tiles][G][16] scale|zero constants
    const void* bias;
    void* out;
    int M, K, N, group_size, bits, act_dtype, scale_dtype;
    int out_f32;  // write unrounded fp32 accumulators (tensor-parallel partial sums)
    const int32_t* perm = nullptr;  // act-order permutation the kernel applies to x itself (decode, M == 1), else nullptr
    int exact_bf16 = 0;  // GPTQHIP_GEMM_EXACT_BF16 (decode kernel, bf16 activations)
    int preload = 0;     // decode form 3: skinny1_kernel (x / constants of a wave's chunks parked in LDS up front: one VMEM instruction per chunk)
    int alg_fp16 = 0;    // decode form 2: algebraic dequant in the skinny kernel (fp16 activations, M <= 4, 4-bit, one constant per chunk)
    int ldo = 0;  // output row stride in elements (0: N) -- lets a launch cover a column sub-range of a wider output
    // batch-1 decode op (gptqhip_decode_linear): fused decoder-layer glue of the skinny kernel's M == 1 variant
    int in_glue = 0;                 // GPTQHIP_GLUE_*
    const void* glue_b = nullptr;    // RMSNORM: norm weight [K]
    const void* residual = nullptr;  // [N]: out = act(residual + y)
    float eps = 0.f;
    const float* stats_in = nullptr; // RMSNORM: producer's per-tile sums of h^2
    int stats_n = 0;
    float* stats_out = nullptr;      // per-tile sums of out^2 for the next op's RMSNorm
    int out_glue = 0;                // GPTQHIP_OUT_*
};

struct SkinnyPlan {
    int mt;                // 16-row activation tiles per block (1, 2, 4)
    int gpc;               // meta words per 128-row chunk (1: group_size % 128 == 0, else 4 = per K-step)
    int chunks;            // ceil(K / 128)
    int waves;             // waves per block (in-block split-K)
    int depth;             // register-ring depth of the kernel variant the launcher will pick
    int regular;           // straight-line pipeline: every wave runs `rounds` whole ring rounds (the last may hold padding chunks)
    int rounds;            // ring rounds per wave on the regular pipeline: rounds * depth * waves >= chunks_per_split
    int chunks_per_split;  // chunks handled by one block
    int splits;            // grid.y (cross-block split-K)
    size_t slab_floats;    // fp32 partial slabs, 0 when splits == 1
    int nt = 1;            // column tiles per block: 4 (2 for 33..64 rows) = the wide-layer kernel (skinny_wide_kernel), else 1
};

struct TiledPlan {
    int gpc;  // meta words per chunk (1 or 4)
    int bm;   // rows per block tile (256 or 128)
    int splits;            // grid.z (split-K through fp32 slabs + reduce kernel)
    int chunks_per_split;
    size_t slab_floats;
    int tail_cols = 0;  // trailing block columns (256 wide) handed to a second launch with 128-row tiles; 0: none
    int bn = 256;       // columns per block: 256, or 128 (one column tile per wave: 4-bit, one constant per chunk, bm 64 / 128)
};

// decode_stream_kernel (gptqhip_stream.hip): batch-1 decode op with the weights streamed HBM -> LDS by LDS-DMA
struct StreamPlan {
    int ok;                // the shape is served (4-bit, K % 128 == 0, one group constant per chunk, LDS map fits 160 KiB)
    int chunks, tiles;
    int waves;             // waves per block (in-block split-K)
    int grid;              // blocks; block b works on tiles b, b + grid, ...
    int tiles_per_block;
    int x_rounds;          // 1 KiB pieces of the x row staged per wave
    int ring_slots;        // 1 KiB slots of a wave's LDS ring
    int off_nw, off_csum, off_stats, off_ring, off_red, off_epi, off_scr, off_meta, lds_bytes;   // LDS map
    int meta_pieces;       // 1 KiB pieces of a tile's constant block
};
// with_norm: the call carries the RMSNorm input glue (norm weight + statistics staged in LDS); with_epi: a residual or bias
StreamPlan plan_stream(int K, int N, int group_size, int bits, int force_waves, bool with_norm, bool with_epi);
int launch_stream(const GemmArgs& a, const StreamPlan& pl, hipStream_t stream);

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// in_kernel_perm: the plan is for the batch-1 act-order variant (AM_ROW1P), which only exists with the 4-deep ring
SkinnyPlan plan_skinny(int M, int K, int N, int group_size, int force_split, int force_waves, bool in_kernel_perm = false, int bits = 4,
                       int allow_wide = 0,    // 0: one column tile per block; 1: wide form from 5 rows; 2: decode op with glue (from 2 rows)
                       int prefer_deep = 0);   // batch 1: 1 = no 2-deep ring plans on K >= 4096 (the preload form of the decode kernel), 2 = ... with the raw-code dequant (form 5)
int launch_skinny(const GemmArgs& a, const SkinnyPlan& pl, float* slabs, int* counters, hipStream_t stream);

TiledPlan plan_tiled(int M, int K, int N, int group_size, int bits, int force_variant, int force_split);
int launch_tiled(const GemmArgs& a, const TiledPlan& pl, float* slabs, hipStream_t stream);

int launch_dequant(const int32_t* qweight, const int32_t* qzeros, const void* scales, const int32_t* g_idx, void* out,
                   int K, int N, int group_size, int bits, int scale_dtype, int out_dtype, hipStream_t stream);
int launch_dequant_tiled(const uint32_t* qweight_t, const uint32_t* meta, const int32_t* perm, void* out, int K, int N,
                         int group_size, int bits, int scale_dtype, int out_dtype, hipStream_t stream);
int launch_repack_awq(const int32_t* qw_awq, const int32_t* qz_awq, int32_t* qw_out, int32_t* qz_out, int K, int N,
                      int G, hipStream_t stream);
int launch_widen_codes(const int32_t* qweight, const int32_t* qzeros, int32_t* qweight_out, int32_t* qzeros_out, int K, int N, int G,
                       int bits, int planar, hipStream_t stream);
int launch_repack_tiled(const int32_t* qweight, const int32_t* qzeros, const void* scales, const int32_t* perm,
                        uint32_t* qweight_t, uint32_t* meta, int K, int N, int group_size, int bits,
                        hipStream_t stream);
int launch_embedding(const int64_t* ids, const uint32_t* qw, const uint32_t* meta, const int32_t* inv_perm, void* out,
                     int32_t* status, int T, int K, int N, int group_size, int bits, int scale_dtype,
                     hipStream_t stream);
int launch_pack_gptq(const float* weight, const float* scales, const int32_t* zeros, const int32_t* g_idx,
                     int32_t* qweight, int32_t* qzeros, int K, int N, int G, int bits, int planar, hipStream_t stream);
int launch_gather_cols(const void* x, const int32_t* perm, void* out, int M, int K, hipStream_t stream);
int launch_rmsnorm_gather(const void* h, const void* weight, const int32_t* perm, void* out, int M, int K, float eps, int act_dtype,
                          hipStream_t stream);

}  // namespace gptqhip
