#define GPTQHIP_ROWS 32
#include "gptqhip_tiled_rows.inc"
