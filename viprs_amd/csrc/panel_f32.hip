#define PANEL_U float
#define PANEL_TAG f32
#include "launch_panel.inc"
