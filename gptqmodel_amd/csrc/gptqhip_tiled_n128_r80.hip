#define GPTQHIP_ROWS 80
#include "gptqhip_tiled_n128.inc"
