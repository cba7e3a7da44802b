#define GPTQHIP_ROWS 32
#include "gptqhip_tiled_n128.inc"
