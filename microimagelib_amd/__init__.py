"""microimagelib_amd -- MI355X-native Richardson-Lucy deconvolution and 3-D affine
registration behind the ``libapi.h`` C API of eguomin/microImageLib.

``microimagelib_amd.api``     host-pointer (NumPy) mirror of libapi.h
``microimagelib_amd.device``  device-resident front end over include/mil_device.h (needs torch)
``microimagelib_amd._lib``    ctypes loader; raises if libapi.so is not built -- no fallback.
"""
from . import _lib  # noqa: F401
from . import api  # noqa: F401

__all__ = ["api", "_lib"]
