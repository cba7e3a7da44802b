// 4-bit prefill kernel with the fp32 epilogue (split-K slabs, tensor-parallel partial sums): its own translation unit.
#include "gptqhip_tiled_kernel.h"

namespace gptqhip {

int launch_tiled_w4_f32(const TiledParams& p, int act_dtype, int scale_dtype, int gpc, int bm, hipStream_t stream) {
    return launch_tiled_bits<4, 1>(p, act_dtype, scale_dtype, gpc, bm, stream);
}

}  // namespace gptqhip
