"""Device-resident front end over ``include/mil_device.h`` for hosts that keep volumes
in HBM (the batch driver, bench.py, the parity tests).  PyTorch is used only as the
owner of device memory / streams; all compute is in ``libapi.so``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _ptr(t: torch.Tensor) -> int:
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("expected a contiguous device tensor")
    return t.data_ptr()


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc}")


def _stream_ptr(stream) -> int:
    if stream is None:
        stream = torch.cuda.current_stream()
    return int(stream.cuda_stream)


# ---- raw kernels -----------------------------------------------------------------
def rl_ratio_(T: torch.Tensor, A: torch.Tensor, stream=None):
    _check(_lib.load().mil_rl_ratio(_ptr(T), _ptr(A), T.numel(), _stream_ptr(stream)), "mil_rl_ratio")
    return T


def rl_update_(E: torch.Tensor, T: torch.Tensor, eps: float = 0.01, stream=None):
    _check(_lib.load().mil_rl_update(_ptr(E), _ptr(T), E.numel(), float(eps), _stream_ptr(stream)), "mil_rl_update")
    return E


def cmul_(S: torch.Tensor, otf: torch.Tensor, stream=None):
    assert S.dtype == torch.complex64 and otf.dtype == torch.complex64
    _check(_lib.load().mil_cmul(_ptr(S), _ptr(otf), S.numel(), _stream_ptr(stream)), "mil_cmul")
    return S


def sum_f64(a: torch.Tensor, stream=None) -> float:
    out = C.c_double(0)
    _check(_lib.load().mil_sum_f64(_ptr(a), a.numel(), C.byref(out), _stream_ptr(stream)), "mil_sum_f64")
    return out.value


def _dims(t):  # tensor [z][y][x] -> (nx, ny, nz)
    return int(t.shape[2]), int(t.shape[1]), int(t.shape[0])


def pad_psf(psf: torch.Tensor, box_shape, scale=1.0, flip=False, stream=None):
    out = torch.empty(tuple(box_shape), dtype=torch.float32, device=psf.device)
    _check(_lib.load().mil_pad_psf(_ptr(out), _ptr(psf), *_dims(out), *_dims(psf), float(scale), int(flip),
                                   _stream_ptr(stream)), "mil_pad_psf")
    return out


def pad_stack(a: torch.Tensor, box_shape, stream=None):
    out = torch.empty(tuple(box_shape), dtype=torch.float32, device=a.device)
    _check(_lib.load().mil_pad_stack(_ptr(out), _ptr(a), *_dims(out), *_dims(a), _stream_ptr(stream)), "mil_pad_stack")
    return out


def crop_stack(a: torch.Tensor, shape, stream=None):
    out = torch.empty(tuple(shape), dtype=torch.float32, device=a.device)
    _check(_lib.load().mil_crop_stack(_ptr(out), _ptr(a), *_dims(out), *_dims(a), _stream_ptr(stream)),
           "mil_crop_stack")
    return out


def align_size(a: torch.Tensor, shape, stream=None):
    out = torch.empty(tuple(shape), dtype=torch.float32, device=a.device)
    _check(_lib.load().mil_align_size(_ptr(out), _ptr(a), *_dims(out), *_dims(a), _stream_ptr(stream)),
           "mil_align_size")
    return out


def flip3(a: torch.Tensor, stream=None):
    out = torch.empty_like(a)
    _check(_lib.load().mil_flip3(_ptr(out), _ptr(a), *_dims(a), _stream_ptr(stream)), "mil_flip3")
    return out


def rot_y90(a: torch.Tensor, direction: int, stream=None):
    out = torch.empty((a.shape[2], a.shape[1], a.shape[0]), dtype=torch.float32, device=a.device)
    _check(_lib.load().mil_rot_y90(_ptr(out), _ptr(a), *_dims(a), int(direction), _stream_ptr(stream)), "mil_rot_y90")
    return out


def max_projection(a: torch.Tensor, direction: int, stream=None):
    sz, sy, sx = a.shape
    shape = {1: (sy, sx), 2: (sx, sz), 3: (sz, sy)}[direction]
    out = torch.empty(shape, dtype=torch.float32, device=a.device)
    _check(_lib.load().mil_max_projection(_ptr(out), _ptr(a), *_dims(a), int(direction), _stream_ptr(stream)),
           "mil_max_projection")
    return out


def mip3d(a: torch.Tensor, r_axis: int, project_num: int, stream=None):
    """``mip3dgpu`` on a device volume: stack [projectNum][h][w] in HBM (rotations about x: r_axis 1, about y: 2)."""
    import ctypes as C
    lib = _lib.load()
    size3 = (C.c_int * 3)()
    _check(lib.mil_mip3d(None, _ptr(a), *_dims(a), int(r_axis), int(project_num), size3, _stream_ptr(stream)), "mil_mip3d")
    out = torch.empty((int(project_num), size3[1], size3[0]), dtype=torch.float32, device=a.device)
    _check(lib.mil_mip3d(_ptr(out), _ptr(a), *_dims(a), int(r_axis), int(project_num), size3, _stream_ptr(stream)), "mil_mip3d")
    return out


def resample_mip_z(src: torch.Tensor, m12, box_shape, stream=None):
    """One angle of the 3-D MIP series with the caller's matrix: resample into a box (nz, ny, nx) and project along its z, fused;
    returns the [ny][nx] plane."""
    m = np.ascontiguousarray(m12, dtype=np.float32).reshape(12)
    bz, by, bx = (int(v) for v in box_shape)
    out = torch.empty((by, bx), dtype=torch.float32, device=src.device)
    _check(_lib.load().mil_resample_mip_z(_ptr(out), _ptr(src), bx, by, bz, *_dims(src), m.ctypes.data_as(_lib.c_float_p),
                                          _stream_ptr(stream)), "mil_resample_mip_z")
    return out


def affine_resample(src: torch.Tensor, m12, out_shape, stream=None):
    out = torch.empty(tuple(out_shape), dtype=torch.float32, device=src.device)
    m = np.ascontiguousarray(m12, dtype=np.float32).reshape(12)
    _check(_lib.load().mil_affine_resample(_ptr(out), _ptr(src), *_dims(out), *_dims(src),
                                           m.ctypes.data_as(_lib.c_float_p), _stream_ptr(stream)),
           "mil_affine_resample")
    return out


class PaddedSource:
    """Source volume copied with a replicated one-voxel border -- the form the registration cost kernel reads
    (csrc/corr_padded.hip).  Pass it to corr_sums in place of the source tensor."""

    def __init__(self, src: torch.Tensor, stream=None):
        if src.dtype != torch.float32 or src.dim() != 3:
            raise ValueError("expected a float32 volume [z][y][x]")
        self.shape = tuple(src.shape)
        self._h = C.c_void_p()
        _check(_lib.load().mil_padded_source_create(C.byref(self._h), _ptr(src), *_dims(src), _stream_ptr(stream)),
               "mil_padded_source_create")

    @staticmethod
    def eligible(target_shape, source_shape) -> bool:
        t, s = tuple(target_shape), tuple(source_shape)
        return bool(_lib.load().mil_padded_source_eligible(t[2], t[1], t[0], s[2], s[1], s[0]))

    def close(self):
        if self._h:
            _lib.load().mil_padded_source_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_texture_weight_bits(bits: int = 0) -> int:
    """0: exact float32 interpolation weights (default); 8: weights rounded to 8 fractional bits (CUDA texture-unit
    emulation, comparison aid).  Process-wide; returns the setting in force."""
    return int(_lib.load().mil_set_texture_weight_bits(int(bits)))


def corr_padded_force_kernel(kernel: int = -1):
    """Test / profiling aid: pin the padded-source cost kernel (0 plain + plane reuse, 1 row pairs, 2 row pairs + plane
    reuse, 3-5 the same with two voxels per lane); -1 restores the per-matrix choice."""
    _lib.load().mil_corr_padded_force_kernel(int(kernel))


def corr_set_fences(on: int = -1):
    """Diagnostic: 1 adds release / acquire fences around the cost kernels' completion counters, 0 removes them, -1 returns
    control to MIL_CORR_FENCES (read once per process)."""
    _lib.load().mil_corr_set_fences(int(on))


def corr_set_chunk_wgs_per_cu(n: int = 0):
    """Experiments: workgroups per compute unit of the chunked cost kernels (0 = the library's choice)."""
    _lib.load().mil_corr_set_chunk_wgs_per_cu(int(n))


def corr_sums(target: torch.Tensor, src, m12, stream=None):
    m = np.ascontiguousarray(m12, dtype=np.float32).reshape(12)
    out = (C.c_double * 2)()
    if isinstance(src, PaddedSource):
        _check(_lib.load().mil_corr_sums_padded(_ptr(target), src._h, *_dims(target), m.ctypes.data_as(_lib.c_float_p),
                                                out, _stream_ptr(stream)), "mil_corr_sums_padded")
        return out[0], out[1]
    _check(_lib.load().mil_corr_sums(_ptr(target), _ptr(src), *_dims(target), *_dims(src),
                                     m.ctypes.data_as(_lib.c_float_p), out, _stream_ptr(stream)), "mil_corr_sums")
    return out[0], out[1]


def reg3d_last_cost_launches() -> int:
    """Cost-kernel launches of the last reg3d_affine on this thread (records[5] counts evaluations; repeats of a point
    just visited cost no launch)."""
    return int(_lib.load().mil_reg3d_last_cost_launches())


def reg3d_trace_begin(capacity: int) -> None:
    """Record the (matrix, cost) of every evaluation this thread's next registrations make (0 stops)."""
    _lib.load().mil_reg3d_trace_begin(int(capacity))


def reg3d_trace_read(max_entries: int):
    """-> (float32 [n][13]: 12 matrix floats + cost per stored evaluation, number of evaluations seen)."""
    buf = np.zeros((int(max_entries), 13), dtype=np.float32)
    seen = int(_lib.load().mil_reg3d_trace_read(buf.ctypes.data_as(_lib.c_float_p), int(max_entries)))
    return buf[:min(seen, int(max_entries))], seen


def reg3d_affine(target: torch.Tensor, source: torch.Tensor, aff_method=6, i_tmx=None, flag_tmx=False, ftol=1e-4,
                 it_limit=3000, verbose=False, stream=None):
    """reg3d_affine1 on device-resident volumes.  Returns (registered, tmx[12], records[11])."""
    reg = torch.empty_like(target)
    tmx = np.zeros(12, dtype=np.float32)
    if i_tmx is not None:
        tmx[:] = np.asarray(i_tmx, dtype=np.float32).reshape(12)
    rec = np.zeros(11, dtype=np.float32)
    _check(_lib.load().mil_reg3d_affine(_ptr(reg), tmx.ctypes.data_as(_lib.c_float_p), _ptr(target), _ptr(source),
                                        *_dims(target), int(aff_method), int(flag_tmx), float(ftol), int(it_limit),
                                        int(verbose), rec.ctypes.data_as(_lib.c_float_p), _stream_ptr(stream)),
           "mil_reg3d_affine")
    return reg, tmx, rec


def reg3d_mip_shift(target: torch.Tensor, source: torch.Tensor, stream=None):
    """regChoice 4 pre-registration: (shift_x, shift_y, shift_z) from 2-D MIP grid searches."""
    shift = np.zeros(3, dtype=np.float32)
    _check(_lib.load().mil_reg3d_mip_shift(shift.ctypes.data_as(_lib.c_float_p), _ptr(target), _ptr(source),
                                           *_dims(target), _stream_ptr(stream)), "mil_reg3d_mip_shift")
    return shift


def reg3d_phasor_shift(target: torch.Tensor, source: torch.Tensor, stream=None):
    """regChoice 1 / 3 pre-registration: integer (shift_x, shift_y, shift_z), source(p) ~ target(p - shift)."""
    shift = (C.c_longlong * 3)()
    _check(_lib.load().mil_reg3d_phasor_shift(shift, _ptr(target), _ptr(source), *_dims(target), _stream_ptr(stream)),
           "mil_reg3d_phasor_shift")
    return np.array([shift[0], shift[1], shift[2]], dtype=np.int64)


def shift_volume(a: torch.Tensor, dx: int, dy: int, dz: int, wrap=False, stream=None):
    out = torch.empty_like(a)
    _check(_lib.load().mil_shift_volume(_ptr(out), _ptr(a), *_dims(a), int(dx), int(dy), int(dz), int(wrap),
                                        _stream_ptr(stream)), "mil_shift_volume")
    return out


# ---- deconvolution engine ----------------------------------------------------------
class DeconEngine:
    """RL engine for one FFT box (shape ``(nz, ny, nx)``) on the current device."""

    TIMER_UPDATE, TIMER_RATIO, TIMER_CMUL, TIMER_FFT = 0, 1, 2, 3
    AUTO, ROCFFT, FUSED = 0, 1, 2

    def __init__(self, box_shape, stream=None, backend=0):
        self._lib = _lib.load()
        self.box_shape = tuple(int(s) for s in box_shape)
        nz, ny, nx = self.box_shape
        h = C.c_void_p()
        # The engine runs on its own (non-default) HIP stream; `_enter` orders it after whatever
        # the caller enqueued on torch's current stream, `sync` makes results visible to the host.
        self.stream = stream if stream is not None else torch.cuda.Stream()
        _check(self._lib.mil_decon_create_backend(C.byref(h), nx, ny, nz, int(self.stream.cuda_stream), int(backend)),
               "mil_decon_create_backend")
        self._h = h
        self.backend = int(self._lib.mil_decon_backend(h))
        self.two_pass = int(self._lib.mil_decon_two_pass(h))  # fused backend: 3 = two passes per convolution (fft2p.hip)
        self.natural_spectrum_shape = (nz, ny, nx // 2 + 1)
        self.spectrum_count = int(self._lib.mil_decon_spectrum_count(h))  # opaque, backend-private layout

    def _enter(self, *tensors):
        """Order the engine's stream after torch's current stream (the producer of the operands) and tell the caching
        allocator that `tensors` are in use on the engine's stream, so that a temporary freed right after the call is
        not handed out again while the engine still reads or writes it."""
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)
            for t in tensors:
                t.record_stream(self.stream)

    def _leave(self):
        """Order torch's current stream after the engine's: results can be consumed by torch ops without `sync()`."""
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mil_decon_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gen_otf(self, psf: torch.Tensor, flip=False) -> torch.Tensor:
        otf = torch.empty(self.spectrum_count, dtype=torch.complex64, device=psf.device)
        px, py, pz = _dims(psf)
        self._enter(otf, psf)
        _check(self._lib.mil_decon_gen_otf(self._h, _ptr(otf), _ptr(psf), px, py, pz, int(flip)), "mil_decon_gen_otf")
        self._leave()
        return otf

    def otf_to_natural(self, otf: torch.Tensor) -> torch.Tensor:
        """Engine-private OTF -> [nz][ny][nx/2+1] complex64 (numpy rfftn order)."""
        nat = torch.empty(self.natural_spectrum_shape, dtype=torch.complex64, device=otf.device)
        self._enter(nat, otf)
        _check(self._lib.mil_decon_otf_to_natural(self._h, _ptr(nat), _ptr(otf)), "mil_decon_otf_to_natural")
        self._leave()
        return nat

    def convolve(self, vol: torch.Tensor, otf: torch.Tensor) -> torch.Tensor:
        """Unnormalised circular convolution C2R(R2C(vol) * otf)."""
        out = torch.empty_like(vol)
        self._enter(out, vol, otf)
        _check(self._lib.mil_decon_convolve(self._h, _ptr(out), _ptr(vol), _ptr(otf)), "mil_decon_convolve")
        self._leave()
        return out

    def init_single(self, A, E, const_init=False):
        self._enter(A, E)
        _check(self._lib.mil_decon_init_single(self._h, _ptr(A), _ptr(E), int(const_init)), "mil_decon_init_single")
        self._leave()

    def init_dual(self, A, B, E, const_init=False):
        self._enter(A, B, E)
        _check(self._lib.mil_decon_init_dual(self._h, _ptr(A), _ptr(B), _ptr(E), int(const_init)),
               "mil_decon_init_dual")
        self._leave()

    def run_single(self, E, A, otf, otf_bp, iters):
        self._enter(E, A, otf, otf_bp)
        _check(self._lib.mil_decon_run_single(self._h, _ptr(E), _ptr(A), _ptr(otf), _ptr(otf_bp), int(iters)),
               "mil_decon_run_single")
        self._leave()

    def run_dual(self, E, A, B, otf1, otf2, otf_bp1, otf_bp2, iters):
        self._enter(E, A, B, otf1, otf2, otf_bp1, otf_bp2)
        _check(self._lib.mil_decon_run_dual(self._h, _ptr(E), _ptr(A), _ptr(B), _ptr(otf1), _ptr(otf2),
                                            _ptr(otf_bp1), _ptr(otf_bp2), int(iters)), "mil_decon_run_dual")
        self._leave()

    def sync(self):
        _check(self._lib.mil_decon_sync(self._h), "mil_decon_sync")

    def timer_enable(self, which, enable=True):
        _check(self._lib.mil_decon_timer_enable(self._h, int(which), int(enable)), "mil_decon_timer_enable")

    def timer_read(self, which):
        ms = C.c_double(0)
        n = C.c_size_t(0)
        _check(self._lib.mil_decon_timer_read(self._h, int(which), C.byref(ms), C.byref(n)), "mil_decon_timer_read")
        return ms.value, n.value
