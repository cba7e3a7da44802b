#define GPTQHIP_W32_ROWS 128
#include "gptqhip_tiled_w32.inc"
