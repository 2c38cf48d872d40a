#define BAND_U int8_t
#include "launch_band.inc"
