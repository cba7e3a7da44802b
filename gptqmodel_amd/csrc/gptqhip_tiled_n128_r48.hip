#define GPTQHIP_ROWS 48
#include "gptqhip_tiled_n128.inc"
