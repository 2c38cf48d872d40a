"""On-disk LD stores (zarr v2 directory stores, blosc-compressed) without zarr / numcodecs / magenpy."""
from .zarr_ld import ZarrArray, ZarrLDMatrix, blosc_compress, blosc_decompress, find_ld_stores, write_ld_store  # noqa: F401
