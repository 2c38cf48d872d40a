"""viprs_amd -- MI355X-native coordinate-ascent E-step for VIPRS (shz9/viprs) over LD blocks.

Hand-written HIP kernels for gfx950 behind a C ABI (``include/viprs_hip.h``) and a ctypes shim
that keeps the reference's ``cpp_e_step`` / ``VIPRS.e_step()`` / ``VIPRS.fit()`` surface.
"""
__version__ = "0.1.0"
