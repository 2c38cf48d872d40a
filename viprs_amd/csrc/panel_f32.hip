#define PANEL_U float
#define PANEL_TRACE_EXPORT
#include "launch_panel.inc"
