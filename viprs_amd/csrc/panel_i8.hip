#define PANEL_U int8_t
#include "launch_panel.inc"
