#define TILE_U int32_t
#include "launch_tile_f64.inc"
