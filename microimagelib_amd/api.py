"""Host-pointer API: thin NumPy front end over the ``libapi.h`` C entry points.

Function names, argument meaning and return conventions are the reference's
(``include/libapi.h`` of eguomin/microImageLib); arrays are C-ordered ``[z][y][x]``
float32 views of the TIFF-ordered buffers the C API takes (``imSize = {sx, sy, sz}``).
Every call goes through ``libapi.so``; nothing is computed in Python.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

c_float_p = _lib.c_float_p
c_uint_p = _lib.c_uint_p


def _f32(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 3:
        raise ValueError("expected a 3-D volume [z][y][x]")
    return a


def _fp(a: np.ndarray):
    return a.ctypes.data_as(c_float_p)


def _size(shape):
    """ndarray shape (sz, sy, sx) -> C ``unsigned int[3] = {sx, sy, sz}``."""
    return (C.c_uint * 3)(int(shape[2]), int(shape[1]), int(shape[0]))


class LibApiError(RuntimeError):
    def __init__(self, fn, code):
        super().__init__(f"{fn} returned {code}")
        self.code = code


def snap_transform_size(n: int) -> int:
    return _lib.load().mil_snap_transform_size(int(n))


# ------------------------------------------------------------------ deconvolution
def _out_like(img, out):
    """The C API writes into a caller-owned buffer; `out` lets a caller reuse one (a fresh 268 MB ndarray costs ~10 ms of
    page faults on its first touch)."""
    if out is None:
        return np.empty_like(img)
    if out.dtype != np.float32 or out.shape != img.shape or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float32 array of the image's shape")
    return out


def release_cached_memory() -> int:
    """Give the device memory the library keeps between calls (buffer pool, engine, OTFs) back to the driver."""
    return int(_lib.load().mil_release_cached_memory())


def decon_singleview(img, psf, iters, const_init=False, device=0, gpu_mem_mode=-1, verbose=False,
                     unmatch=False, psf_bp=None, out=None):
    """``decon_singleview`` (reference include/libapi.h:42-43).  Returns (decon, records[10])."""
    lib = _lib.load()
    img, psf = _f32(img), _f32(psf)
    bp = _f32(psf_bp) if (unmatch and psf_bp is not None) else np.zeros(1, dtype=np.float32)
    if unmatch and psf_bp is None:
        raise ValueError("unmatch=True needs psf_bp")
    out = _out_like(img, out)
    rec = np.zeros(10, dtype=np.float32)
    rc = lib.decon_singleview(_fp(out), _fp(img), _size(img.shape), _fp(psf), _size(psf.shape), bool(const_init),
                              int(iters), int(device), int(gpu_mem_mode), bool(verbose), _fp(rec), bool(unmatch),
                              _fp(bp))
    if rc != 0:
        raise LibApiError("decon_singleview", rc)
    return out, rec


def decon_dualview(img1, img2, psf1, psf2, iters, const_init=False, device=0, gpu_mem_mode=-1, verbose=False,
                   unmatch=False, psf_bp1=None, psf_bp2=None, out=None):
    """``decon_dualview`` (reference include/libapi.h:45-46).  Returns (decon, records[10])."""
    lib = _lib.load()
    img1, img2, psf1, psf2 = _f32(img1), _f32(img2), _f32(psf1), _f32(psf2)
    if img1.shape != img2.shape or psf1.shape != psf2.shape:
        raise ValueError("both views (and both PSFs) must have the same size")
    if unmatch and (psf_bp1 is None or psf_bp2 is None):
        raise ValueError("unmatch=True needs psf_bp1 and psf_bp2")
    bp1 = _f32(psf_bp1) if unmatch else np.zeros(1, dtype=np.float32)
    bp2 = _f32(psf_bp2) if unmatch else np.zeros(1, dtype=np.float32)
    out = _out_like(img1, out)
    rec = np.zeros(10, dtype=np.float32)
    rc = lib.decon_dualview(_fp(out), _fp(img1), _fp(img2), _size(img1.shape), _fp(psf1), _fp(psf2),
                            _size(psf1.shape), bool(const_init), int(iters), int(device), int(gpu_mem_mode),
                            bool(verbose), _fp(rec), bool(unmatch), _fp(bp1), _fp(bp2))
    if rc != 0:
        raise LibApiError("decon_dualview", rc)
    return out, rec


# ------------------------------------------------------------------ registration
def reg3d(target, source, i_tmx=None, reg_choice=2, aff_method=6, input_tmx=False, ftol=1e-4, it_limit=3000,
          device=0, gpu_mem_mode=-1, verbose=False):
    """``reg3d`` (reference include/libapi.h:35-36).  Returns (registered, tmx[12], records[11])."""
    lib = _lib.load()
    target, source = _f32(target), _f32(source)
    tmx = np.zeros(12, dtype=np.float32)
    if i_tmx is not None:
        tmx[:] = np.asarray(i_tmx, dtype=np.float32).reshape(12)
    out = np.empty_like(target)
    rec = np.zeros(11, dtype=np.float32)
    rc = lib.reg3d(_fp(out), _fp(tmx), _fp(target), _fp(source), _size(target.shape), _size(source.shape),
                   int(reg_choice), int(aff_method), bool(input_tmx), float(ftol), int(it_limit), int(device),
                   int(gpu_mem_mode), bool(verbose), _fp(rec))
    if rc != 0:
        raise LibApiError("reg3d", rc)
    return out, tmx, rec


def reg2d(target, source, i_tmx=None, reg_choice=2, input_tmx=False, ftol=1e-4, it_limit=3000, device=0,
          gpu_mem_mode=1, verbose=False):
    """``reg2d`` (reference include/libapi.h:24-25) on [h][w] images.  Returns (registered, tmx[6], records[11])."""
    lib = _lib.load()
    target = np.ascontiguousarray(target, dtype=np.float32)
    source = np.ascontiguousarray(source, dtype=np.float32)
    if target.ndim != 2 or source.ndim != 2:
        raise ValueError("expected 2-D images [y][x]")
    tmx = np.zeros(6, dtype=np.float32)
    if i_tmx is not None:
        tmx[:] = np.asarray(i_tmx, dtype=np.float32).reshape(6)
    out = np.zeros_like(target)
    rec = np.zeros(11, dtype=np.float32)
    s1 = (C.c_uint * 3)(target.shape[1], target.shape[0], 1)
    s2 = (C.c_uint * 3)(source.shape[1], source.shape[0], 1)
    rc = lib.reg2d(_fp(out), _fp(tmx), _fp(target), _fp(source), s1, s2, int(reg_choice), bool(input_tmx), float(ftol),
                   int(it_limit), int(device), int(gpu_mem_mode), bool(verbose), _fp(rec))
    if rc != 0:
        raise LibApiError("reg2d", rc)
    return out, tmx, rec


def atrans3dgpu(source, tmx, out_shape, device=0):
    """``atrans3dgpu`` (reference include/libapi.h:30)."""
    lib = _lib.load()
    source = _f32(source)
    m = np.ascontiguousarray(tmx, dtype=np.float32).reshape(12)
    out = np.empty(tuple(int(s) for s in out_shape), dtype=np.float32)
    rc = lib.atrans3dgpu(_fp(out), _fp(m), _fp(source), _size(out.shape), _size(source.shape), int(device))
    if rc != 0:
        raise LibApiError("atrans3dgpu", rc)
    return out


def checkmatrix(tmx, sx, sy, sz) -> bool:
    m = np.ascontiguousarray(tmx, dtype=np.float32).reshape(12)
    return bool(_lib.load().checkmatrix(_fp(m), int(sx), int(sy), int(sz)))


# ------------------------------------------------------------------ projections / resize / rotate
def mp2dgpu(img, z_proj=True, x_proj=True, y_proj=True):
    """``mp2dgpu`` (include/libapi.h:54).  Returns (xy[sy][sx], yz[sz][sy], zx[sx][sz])."""
    lib = _lib.load()
    img = _f32(img)
    sz, sy, sx = img.shape
    mp = np.zeros(sx * sy + sy * sz + sz * sx, dtype=np.float32)
    size_mp = (C.c_uint * 6)()
    rc = lib.mp2dgpu(_fp(mp), size_mp, _fp(img), _size(img.shape), bool(z_proj), bool(x_proj), bool(y_proj))
    if rc != 0:
        raise LibApiError("mp2dgpu", rc)
    a = mp[: sx * sy].reshape(sy, sx)
    b = mp[sx * sy: sx * sy + sy * sz].reshape(sz, sy)
    c = mp[sx * sy + sy * sz:].reshape(sx, sz)
    return a, b, c


def mip3dgpu(img, r_axis, project_num):
    """``mip3dgpu`` (include/libapi.h:59-60).  Returns stack [projectNum][h][w]."""
    lib = _lib.load()
    img = _f32(img)
    sz, sy, sx = img.shape
    if r_axis == 1:
        rot = int(round(np.sqrt(float(sy * sy + sz * sz))))
        w, h = sx, rot
    elif r_axis == 2:
        rot = int(round(np.sqrt(float(sx * sx + sz * sz))))
        w, h = rot, sy
    else:
        raise ValueError("r_axis must be 1 or 2")
    out = np.zeros((int(project_num), h, w), dtype=np.float32)
    size_mp = (C.c_uint * 6)()
    rc = lib.mip3dgpu(_fp(out), size_mp, _fp(img), _size(img.shape), int(r_axis), int(project_num))
    if rc != 0:
        raise LibApiError("mip3dgpu", rc)
    assert (size_mp[0], size_mp[1], size_mp[2]) == (w, h, int(project_num))
    return out


def imresize3d(img, out_shape, device=0):
    """``imresize3d`` (include/libapi.h:65-66): resample to ``out_shape`` (sz, sy, sx)."""
    lib = _lib.load()
    img = _f32(img)
    out = np.empty(tuple(int(s) for s in out_shape), dtype=np.float32)
    rc = lib.imresize3d(_fp(out), _fp(img), out.shape[2], out.shape[1], out.shape[0], img.shape[2], img.shape[1],
                        img.shape[0], int(device))
    if rc != 0:
        raise LibApiError("imresize3d", rc)
    return out


def imoperation3D(img, op_choice, device=0):
    """``imoperation3D`` (include/libapi.h:68): 1 = +90 deg about Y, 2 = -90 deg, 0 = nothing."""
    lib = _lib.load()
    img = _f32(img)
    if op_choice == 0:
        return img.copy()
    out = np.empty((img.shape[2], img.shape[1], img.shape[0]), dtype=np.float32)
    size_out = (C.c_uint * 3)()
    rc = lib.imoperation3D(_fp(out), size_out, _fp(img), _size(img.shape), int(op_choice), int(device))
    if rc != 0:
        raise LibApiError("imoperation3D", rc)
    return out


def alignsize3d(img, out_shape, gpu_mem_mode=1):
    """``alignsize3d`` (include/libapi.h:62-63): centred crop / zero-pad to ``out_shape``."""
    lib = _lib.load()
    img = _f32(img)
    out = np.empty(tuple(int(s) for s in out_shape), dtype=np.float32)
    # the C entry indexes [x*sy*sz + y*sz + z]: its first extent is the slowest axis
    rc = lib.alignsize3d(_fp(out), _fp(img), out.shape[0], out.shape[1], out.shape[2], img.shape[0], img.shape[1],
                         img.shape[2], int(gpu_mem_mode))
    if rc != 0:
        raise LibApiError("alignsize3d", rc)
    return out


# ------------------------------------------------------------------ TIFF
def gettifinfo(path: str):
    size = (C.c_uint * 3)()
    bits = _lib.load().gettifinfo(path.encode(), size)
    return int(bits), (int(size[0]), int(size[1]), int(size[2]))


def readtifstack(path: str, out=None) -> np.ndarray:
    """``readtifstack`` (reference include/libapi.h); `out`: a caller-owned float32 [z][y][x] buffer of the stack's size."""
    _, (sx, sy, sz) = gettifinfo(path)
    if out is None:
        out = np.empty((sz, sy, sx), dtype=np.float32)
    elif out.dtype != np.float32 or out.shape != (sz, sy, sx) or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float32 array of the stack's shape")
    size = (C.c_uint * 3)()
    _lib.load().readtifstack(_fp(out), path.encode(), size)
    return out


def writetifstack(path: str, img, bits: int = 32) -> None:
    img = _f32(img)
    _lib.load().writetifstack(path.encode(), _fp(img), _size(img.shape), int(bits))
