#define GRID_U int16_t
#include "launch_grid.inc"
