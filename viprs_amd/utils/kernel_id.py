"""Identity of the kernel sources behind a measurement (profiles/pmc_traffic.json): a PMC traffic figure is a constant of
the kernel it was collected on -- `bench.py` quotes it only while the sources of that kernel family are unchanged."""
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
_COMMON = ("kernels_common.h", "device_math.h")
FAMILIES = {
    "panel": ("estep_panel.h", "launch_panel.inc"),
    "grid": ("estep_grid_mfma.h", "launch_grid.inc"),
    "tile_f64": ("estep_tile.h", "launch_tile_f64.inc"),
}


def family_of(traffic_key):
    """'cfg3_float32_sym' -> 'panel', '..._grid32' -> 'grid', '..._f64' -> 'tile_f64'."""
    if "grid" in traffic_key:
        return "grid"
    if traffic_key.endswith("_f64"):
        return "tile_f64"
    return "panel"


def source_hash(family):
    h = hashlib.sha1()
    for fn in FAMILIES[family] + _COMMON:
        with open(os.path.join(_CSRC, fn), "rb") as f:
            h.update(fn.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:12]
