"""svision_amd -- MI355X-native hot path of SVision (signature collection,
similarity-image encoding, CNN classification) behind SVision's own surfaces.

The compute kernels live in ``libsvx.so`` (hand-written HIP for gfx950, C ABI in
``include/svx.h``), bound with ctypes in :mod:`svision_amd._lib`.  There is no
CPU fallback: importing the device ops without the built library raises.
"""
__version__ = "0.1.0"
