#define PANEL_U float
#include "launch_panel.inc"
