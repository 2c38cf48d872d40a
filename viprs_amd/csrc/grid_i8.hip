#define GRID_U int8_t
#include "launch_grid.inc"
