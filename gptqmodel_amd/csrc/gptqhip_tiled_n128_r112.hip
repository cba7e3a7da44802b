#define GPTQHIP_ROWS 112
#include "gptqhip_tiled_n128.inc"
