// 8-bit instantiations of the prefill kernel template (its own translation unit: compiles in parallel with the 4-bit half).
#include "gptqhip_tiled_kernel.h"

namespace gptqhip {

int launch_tiled_w8(const TiledParams& p, int act_dtype, int scale_dtype, int gpc, int bm, hipStream_t stream) {
    return launch_tiled_bits<8>(p, act_dtype, scale_dtype, gpc, bm, stream);
}

}  // namespace gptqhip
