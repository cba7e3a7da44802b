// Word layouts of one group of 32 codes for every bit width the reference packs (host + device; quantisation- / load-time helpers,
// not on the hot path).  Continuous: code i sits at bit `bits * i` of the little-endian stream of `bits` int32 words (2 / 4 / 8 bits
// tile a word; 3-bit codes 10 and 21 straddle one: gptqmodel/nn_modules/qlinear/__init__.py:982-991).  Planar ("split-plane",
// gptqmodel/utils/planar_packing.py:7-24; always for 5 / 6 / 7 bits, for 3 bits under FORMAT.GPTQ_P): the code is cut into planes of
// width 4 / 2 / 1 from the low bits up -- 3: (2)(1), 5: (4)(1), 6: (4)(2), 7: (4)(2)(1) -- each plane stored as `width` words in
// which word j holds the plane's field of codes [j * 32 / width, (j + 1) * 32 / width) at shifts width * (i mod 32 / width).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GPTQHIP_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define GPTQHIP_HD inline
#endif

namespace gptqhip {

GPTQHIP_HD int plane_width(int left) { return left >= 8 ? 8 : (left >= 4 ? 4 : (left >= 2 ? 2 : 1)); }

// codes c[0..31] (each < 2^bits) -> out[0..bits-1]
GPTQHIP_HD void encode_group32(const uint32_t* c, int bits, int planar, uint32_t* out) {
    for (int t = 0; t < bits; ++t) out[t] = 0u;
    if (!planar) {
        for (int i = 0; i < 32; ++i) {
            const int pos = bits * i, w = pos >> 5, sh = pos & 31;
            out[w] |= c[i] << sh;
            if (sh + bits > 32) out[w + 1] |= c[i] >> (32 - sh);
        }
        return;
    }
    int row = 0, off = 0, left = bits;
    while (left > 0) {
        const int width = plane_width(left), pf = 32 / width;
        for (int i = 0; i < 32; ++i) out[row + i / pf] |= ((c[i] >> off) & ((1u << width) - 1u)) << (width * (i % pf));
        row += width;
        off += width;
        left -= width;
    }
}

}  // namespace gptqhip
