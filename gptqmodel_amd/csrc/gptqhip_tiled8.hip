// 8-bit instantiations of the prefill kernel template (its own translation unit: compiles in parallel with the 4-bit half).
#include "gptqhip_tiled_kernel.h"

namespace gptqhip {

int launch_tiled_w8(const TiledParams& p, int act_dtype, int scale_dtype, int gpc, int bm, hipStream_t stream) {
    if (p.splits > 1 || p.out_f32) return launch_tiled_bits<8, 1>(p, act_dtype, scale_dtype, gpc, bm, stream);
    return launch_tiled_bits<8, 0>(p, act_dtype, scale_dtype, gpc, bm, stream);
}

}  // namespace gptqhip
