"""svision_amd -- MI355X-native hot path of SVision (signature collection,
similarity-image encoding, CNN classification) behind SVision's own surfaces.

The compute kernels live in ``libsvx.so`` (hand-written HIP for gfx950, C ABI in
``include/svx.h``), bound with ctypes in :mod:`svision_amd._lib`.  There is no
CPU fallback: importing the device ops without the built library raises.
"""
__version__ = "0.1.0"


def _check_host_modules():
    import sys
    if "svision_amd.build_host" in sys.modules or (sys.argv and sys.argv[0].endswith("build_host.py")):
        return
    # read-only: a compiled host module older than its .py source is bypassed (meta-path finder), never deleted here
    from .build_host import guard_imports
    guard_imports(log=lambda msg: print(msg, file=sys.stderr))


_check_host_modules()
del _check_host_modules
