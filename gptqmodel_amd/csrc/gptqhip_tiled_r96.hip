#define GPTQHIP_ROWS 96
#include "gptqhip_tiled_rows.inc"
