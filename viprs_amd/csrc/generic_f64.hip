#define GENERIC_T double
#include "launch_generic.inc"
