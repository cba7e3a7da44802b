#define GPTQHIP_ROWS 64
#include "gptqhip_tiled_n128.inc"
