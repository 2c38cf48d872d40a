#define TILE_U int16_t
#include "launch_tile_f64.inc"
