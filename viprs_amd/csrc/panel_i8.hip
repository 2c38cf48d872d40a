#define PANEL_U int8_t
#define PANEL_TAG i8
#include "launch_panel.inc"
