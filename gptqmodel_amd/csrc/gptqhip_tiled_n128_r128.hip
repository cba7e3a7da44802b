#define GPTQHIP_ROWS 128
#include "gptqhip_tiled_n128.inc"
