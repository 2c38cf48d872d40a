#define PANEL_U int16_t
#include "launch_panel.inc"
