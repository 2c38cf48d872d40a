#define TILE_U float
#include "launch_tile_f64.inc"
