#define TILE_U double
#include "launch_tile_f64.inc"
