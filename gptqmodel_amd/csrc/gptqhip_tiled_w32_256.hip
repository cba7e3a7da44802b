#define GPTQHIP_W32_ROWS 256
#include "gptqhip_tiled_w32.inc"
