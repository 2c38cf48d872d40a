#define PANEL_U int16_t
#define PANEL_TAG i16
#include "launch_panel.inc"
