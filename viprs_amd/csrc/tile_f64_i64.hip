#define TILE_U int64_t
#include "launch_tile_f64.inc"
