// Host-side glue shared by the translation units of libgptqhip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace gptqhip {

struct GemmArgs {
    const void* x;
    const int32_t* qweight;
    const int32_t* qzeros;
    const void* scales;
    const void* bias;
    void* out;
    int M, K, N, group_size, bits, act_dtype, scale_dtype;
};

struct SkinnyPlan {
    int spg;               // K-steps (32 k) that share one group's scale/zero per loop iteration
    int chunks_total;      // K / (32*spg)
    int chunks_per_split;  // chunks handled by one block (its 4 waves split them again)
    int splits;            // grid.y
    size_t slab_floats;    // fp32 partial slabs, 0 when splits == 1
    int counters;          // one arrival counter per 64-column strip
};

struct TiledPlan {
    int splits;
    size_t slab_floats;
    int counters;
};

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

SkinnyPlan plan_skinny(int M, int K, int N, int group_size, int force_split);
int launch_skinny(const GemmArgs& a, const SkinnyPlan& pl, float* slabs, int* counters, hipStream_t stream);

TiledPlan plan_tiled(int M, int K, int N, int group_size);
int launch_tiled(const GemmArgs& a, const TiledPlan& pl, float* slabs, int* counters, hipStream_t stream);

int launch_dequant(const int32_t* qweight, const int32_t* qzeros, const void* scales, const int32_t* g_idx, void* out,
                   int K, int N, int group_size, int bits, int scale_dtype, int out_dtype, hipStream_t stream);
int launch_repack_awq(const int32_t* qw_awq, const int32_t* qz_awq, int32_t* qw_out, int32_t* qz_out, int K, int N,
                      int G, hipStream_t stream);
int launch_repack_rows(const int32_t* qweight, const int32_t* perm, int32_t* out, int K, int N, int bits,
                       hipStream_t stream);
int launch_gather_cols(const void* x, const int32_t* perm, void* out, int M, int K, hipStream_t stream);

}  // namespace gptqhip
