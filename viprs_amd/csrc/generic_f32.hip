#define GENERIC_T float
#include "launch_generic.inc"
