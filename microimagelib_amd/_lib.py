"""ctypes loader for the in-tree ``libapi.so`` (C ABIs of include/libapi.h + include/mil_device.h).

There is no Python or CPU fallback anywhere in this package: if the shared library is
missing or fails to load, importing the bindings raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIL_LIBAPI_PATH: load another build of the library (A/B timing of experimental builds; tools/ only)
LIB_PATH = os.environ.get("MIL_LIBAPI_PATH") or os.path.join(_HERE, "libapi.so")

c_float_p = C.POINTER(C.c_float)
c_uint_p = C.POINTER(C.c_uint)
c_double_p = C.POINTER(C.c_double)
COST_FN = C.CFUNCTYPE(C.c_float, c_float_p, C.c_void_p)

# name -> (restype, argtypes); covers every symbol declared in include/*.h
SIGNATURES = {
    # ---- include/libapi.h
    "concat": (C.c_void_p, None),  # variadic
    "fexists": (C.c_bool, [C.c_char_p]),
    "gettifinfo": (C.c_ushort, [C.c_char_p, c_uint_p]),
    "readtifstack": (None, [c_float_p, C.c_char_p, c_uint_p]),
    "writetifstack": (None, [C.c_char_p, c_float_p, c_uint_p, C.c_ushort]),
    "readtifstack_16to16": (None, [C.POINTER(C.c_ushort), C.c_char_p, c_uint_p]),
    "writetifstack_16to16": (None, [C.c_char_p, C.POINTER(C.c_ushort), c_uint_p]),
    "queryDevice": (None, []),
    "checkmatrix": (C.c_bool, [c_float_p, C.c_longlong, C.c_longlong, C.c_longlong]),
    "atrans3dgpu": (C.c_int, [c_float_p, c_float_p, c_float_p, c_uint_p, c_uint_p, C.c_int]),
    "atrans3dgpu_16bit": (C.c_int, [C.POINTER(C.c_ushort), c_float_p, C.POINTER(C.c_ushort), c_uint_p, c_uint_p,
                                    C.c_int]),
    "reg3d": (C.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_uint_p, c_uint_p, C.c_int, C.c_int, C.c_bool,
                        C.c_float, C.c_int, C.c_int, C.c_int, C.c_bool, c_float_p]),
    "reg_3dgpu": (C.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_uint_p, c_uint_p, C.c_int, C.c_int,
                            C.c_float, C.c_int, C.c_int, C.c_int, c_float_p]),
    "reg2d": (C.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_uint_p, c_uint_p, C.c_int, C.c_bool, C.c_float,
                        C.c_int, C.c_int, C.c_int, C.c_bool, c_float_p]),
    "decon_singleview": (C.c_int, [c_float_p, c_float_p, c_uint_p, c_float_p, c_uint_p, C.c_bool, C.c_int, C.c_int,
                                   C.c_int, C.c_bool, c_float_p, C.c_bool, c_float_p]),
    "decon_dualview": (C.c_int, [c_float_p, c_float_p, c_float_p, c_uint_p, c_float_p, c_float_p, c_uint_p, C.c_bool,
                                 C.c_int, C.c_int, C.c_int, C.c_bool, c_float_p, C.c_bool, c_float_p, c_float_p]),
    "fusion_dualview": (C.c_int, None),
    "mp2dgpu": (C.c_int, [c_float_p, c_uint_p, c_float_p, c_uint_p, C.c_bool, C.c_bool, C.c_bool]),
    "mp3dgpu": (C.c_int, [c_float_p, c_uint_p, c_float_p, c_uint_p, C.c_bool, C.c_bool, C.c_int]),
    "mip3dgpu": (C.c_int, [c_float_p, c_uint_p, c_float_p, c_uint_p, C.c_int, C.c_longlong]),
    "alignsize3d": (C.c_int, [c_float_p, c_float_p] + [C.c_longlong] * 6 + [C.c_int]),
    "imresize3d": (C.c_int, [c_float_p, c_float_p] + [C.c_longlong] * 6 + [C.c_int]),
    "imoperation3D": (C.c_int, [c_float_p, c_uint_p, c_float_p, c_uint_p, C.c_int, C.c_int]),
    # ---- include/mil_device.h
    "mil_version": (C.c_char_p, []),
    "mil_device_count": (C.c_int, []),
    "mil_set_device": (C.c_int, [C.c_int]),
    "mil_dev_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "mil_snap_transform_size": (C.c_int, [C.c_int]),
    "mil_release_cached_memory": (C.c_size_t, []),
    "mil_cached_memory_bytes": (C.c_size_t, []),
    "mil_decon_last_call_split": (C.c_int, [c_double_p]),
    "mil_set_texture_weight_bits": (C.c_int, [C.c_int]),
    "mil_rl_ratio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mil_rl_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]),
    "mil_cmul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mil_sum_f64": (C.c_int, [C.c_void_p, C.c_size_t, c_double_p, C.c_void_p]),
    "mil_pad_psf": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.c_int, C.c_void_p]),
    "mil_pad_stack": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "mil_crop_stack": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "mil_align_size": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "mil_flip3": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p]),
    "mil_rot_y90": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "mil_max_projection": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "mil_mip3d": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_longlong, C.POINTER(C.c_int), C.c_void_p]),
    "mil_decon_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mil_decon_create_backend": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "mil_decon_destroy": (None, [C.c_void_p]),
    "mil_decon_backend": (C.c_int, [C.c_void_p]),
    "mil_decon_two_pass": (C.c_int, [C.c_void_p]),
    "mil_fx_two_pass_box": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "mil_decon_convolve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mil_decon_otf_to_natural": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mil_fft_digit_reversal": (C.c_int, [C.c_int, C.POINTER(C.c_ushort)]),
    "mil_decon_real_count": (C.c_size_t, [C.c_void_p]),
    "mil_decon_spectrum_count": (C.c_size_t, [C.c_void_p]),
    "mil_decon_gen_otf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mil_decon_init_single": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mil_decon_init_dual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mil_decon_run_single": (C.c_int, [C.c_void_p] * 5 + [C.c_int]),
    "mil_decon_run_dual": (C.c_int, [C.c_void_p] * 8 + [C.c_int]),
    "mil_decon_sync": (C.c_int, [C.c_void_p]),
    "mil_decon_timer_enable": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mil_decon_timer_read": (C.c_int, [C.c_void_p, C.c_int, c_double_p, C.POINTER(C.c_size_t)]),
    "mil_affine_resample": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [c_float_p, C.c_void_p]),
    "mil_affine_resample_path": (C.c_int, [C.c_int] * 6),
    "mil_dev_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "mil_dev_free": (None, [C.c_void_p]),
    "mil_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "mil_host_free": (None, [C.c_void_p]),
    "mil_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "mil_stream_destroy": (None, [C.c_void_p]),
    "mil_stream_sync": (C.c_int, [C.c_void_p]),
    "mil_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mil_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mil_u16_to_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mil_f32_to_u16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mil_imresize3d": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_longlong] * 6 + [C.c_void_p]),
    "mil_reg3d": (C.c_int, [C.c_void_p, c_float_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_int, C.c_int,
                            C.c_int, C.c_float, C.c_int, C.c_int, c_float_p, C.c_void_p]),
    "mil_decon_singleview": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint), c_float_p, C.POINTER(C.c_uint), C.c_int, C.c_int,
                                       C.c_int, c_float_p, C.c_int, c_float_p, C.c_void_p]),
    "mil_decon_dualview": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint), c_float_p, c_float_p,
                                     C.POINTER(C.c_uint), C.c_int, C.c_int, C.c_int, c_float_p, C.c_int, c_float_p, c_float_p,
                                     C.c_void_p]),
    "mil_resample_mip_z": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [c_float_p, C.c_void_p]),
    "mil_corr_sums": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [c_float_p, c_double_p, C.c_void_p]),
    "mil_padded_source_eligible": (C.c_int, [C.c_int] * 6),
    "mil_reg3d_last_cost_launches": (C.c_int, []),
    "mil_reg3d_trace_begin": (C.c_int, [C.c_int]),
    "mil_reg3d_trace_read": (C.c_int, [c_float_p, C.c_int]),
    "mil_corr_padded_force_kernel": (C.c_int, [C.c_int]),
    "mil_corr_set_fences": (C.c_int, [C.c_int]),
    "mil_corr_set_chunk_wgs_per_cu": (C.c_int, [C.c_int]),
    "mil_corr_padded_pick_kernel": (C.c_int, [c_float_p]),
    "mil_corr_padded_pick_kernel2": (C.c_int, [c_float_p, C.c_int]),
    "mil_padded_source_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p] + [C.c_int] * 3 + [C.c_void_p]),
    "mil_padded_source_destroy": (None, [C.c_void_p]),
    "mil_corr_sums_padded": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [c_float_p, c_double_p, C.c_void_p]),
    "mil_reg3d_affine": (C.c_int, [C.c_void_p, c_float_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_int, C.c_int, c_float_p, C.c_void_p]),
    "mil_reg3d_mip_shift": (C.c_int, [c_float_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mil_reg3d_phasor_shift": (C.c_int, [C.POINTER(C.c_longlong), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p]),
    "mil_shift_volume": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong,
                                   C.c_longlong, C.c_int, C.c_void_p]),
    "mil_powell": (C.c_int, [c_float_p, c_float_p, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), c_float_p,
                             COST_FN, C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "mil_p2matrix": (None, [c_float_p, c_float_p]),
    "mil_matrix2p": (None, [c_float_p, c_float_p]),
    "mil_dof9tomatrix": (None, [c_float_p, c_float_p, C.c_int]),
    "mil_matrixmultiply": (None, [c_float_p, c_float_p, c_float_p]),
    "mil_cpu_rfft3d": (C.c_int, [c_float_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mil_rot2matrix": (None, [c_float_p, C.c_float, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int]),
}

_lib = None


def load() -> C.CDLL:
    """Load libapi.so (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C microimagelib_amd/csrc`. There is no fallback path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        if argtypes is not None:
            fn.argtypes = argtypes
    _lib = lib
    return lib
