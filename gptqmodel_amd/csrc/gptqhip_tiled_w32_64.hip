#define GPTQHIP_W32_ROWS 64
#include "gptqhip_tiled_w32.inc"
