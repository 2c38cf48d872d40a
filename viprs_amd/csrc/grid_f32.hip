#define GRID_U float
#include "launch_grid.inc"
