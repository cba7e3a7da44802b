#define GPTQHIP_ROWS 48
#include "gptqhip_tiled_rows.inc"
