#define BAND_U int16_t
#include "launch_band.inc"
