#define GPTQHIP_ROWS 96
#include "gptqhip_tiled_n128.inc"
