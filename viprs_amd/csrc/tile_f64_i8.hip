#define TILE_U int8_t
#include "launch_tile_f64.inc"
