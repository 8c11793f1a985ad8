"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/libapi.h and include/mil_device.h declare (no compute is attempted here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = text.replace('extern "C" {', " ")
    names = []
    for stmt in text.split(";"):
        stmt = " ".join(stmt.split())
        if stmt.startswith("typedef") or "(" not in stmt:
            continue
        m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*\(", stmt)
        if m:
            names.append(m.group(1))
    return names


@pytest.mark.parametrize("header", ["libapi.h", "mil_device.h"])
def test_every_declared_symbol_is_exported(lib, header):
    names = declared_functions(header)
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"{header}: not exported: {missing}"


def test_python_signature_table_covers_headers():
    from microimagelib_amd import _lib
    declared = set(declared_functions("libapi.h")) | set(declared_functions("mil_device.h"))
    assert declared <= set(_lib.SIGNATURES), sorted(declared - set(_lib.SIGNATURES))


def test_reference_cli_symbols_present(lib):
    """Symbols the reference CLIs reference."""
    for name in ("gettifinfo", "readtifstack", "writetifstack", "fexists", "concat", "decon_singleview",
                 "decon_dualview", "reg3d", "imresize3d", "imoperation3D", "queryDevice", "checkmatrix", "mp2dgpu",
                 "mip3dgpu"):
        assert hasattr(lib, name)


def test_library_reports_no_fallback_when_missing(tmp_path, monkeypatch):
    from microimagelib_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libapi.so"))
    with pytest.raises(ImportError):
        _lib.load()


def test_snap_transform_size_table(lib):
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "tables.npz"))
    for n, want in zip(g["snap_in"], g["snap_out"]):
        assert lib.mil_snap_transform_size(int(n)) == int(want)
    # SURVEY.md section 4.3
    for n, want in ((20, 32), (36, 64), (50, 64), (100, 128), (128, 128), (130, 192), (256, 256), (300, 320),
                    (500, 512), (512, 512)):
        assert lib.mil_snap_transform_size(n) == want


def test_two_pass_box_table(lib):
    """Which FFT boxes run two passes per convolution (csrc/fft2p*.hip) and how wide their yz tiles / OTF column groups are:
    host-side rule, no device.  x = 192 .. 1024 and y = 128 .. 1024 in multiples of 64, z a multiple of 64 with
    (ny / 64) * nz <= 4096; the LDS yz kernels take 16 columns below (ny / 64) * nz = 512, then the widest of 8 / 4 / 2 that leaves
    two workgroups per CU."""
    import ctypes

    def q(nx, ny, nz):
        c = ctypes.c_int(-1)
        ok = lib.mil_fx_two_pass_box(nx, ny, nz, ctypes.byref(c))
        return ok, c.value

    assert q(512, 512, 256) == (1, 4)                      # the benchmark box: LDS yz kernel, T x nz = 2048 (MIL_FX_YZ_REGISTER=1: 8)
    assert q(384, 512, 256) == (1, 4)                      # same (y, z), x not a power of two
    assert q(384, 384, 256) == (1, 4) and q(448, 320, 128) == (1, 8)   # the boxes of the round-4 review
    assert q(192, 192, 64) == (1, 16) and q(256, 128, 64) == (1, 16)   # small tiles: whole 128-byte row segments
    assert q(384, 384, 384) == (1, 4) and q(512, 512, 512) == (1, 2) and q(1024, 1024, 256) == (1, 2)
    assert q(1024, 1024, 512) == (0, 16) and q(512, 1024, 1024) == (0, 16)   # (ny / 64) * nz beyond 4096: classic sequence
    assert q(128, 128, 64) == (0, 16) and q(512, 512, 32) == (0, 16) and q(512, 64, 64) == (0, 16)
    assert q(200, 256, 64) == (0, 16) and q(0, 0, 0) == (0, 16)
    n = 0
    for nx in range(192, 1025, 64):
        for ny in range(128, 1025, 64):
            for nz in range(64, 1025, 64):
                ok, cols = q(nx, ny, nz)
                assert ok == int((ny // 64) * nz <= 4096), (nx, ny, nz)
                if ok:
                    n += 1
                    t = ny // 64
                    assert cols in (2, 4, 8, 16) and (cols == 16) == (t * nz < 512)
                    assert t * nz * cols <= 16384  # the tile fits the LDS
    assert n == 1764  # 14 x extents times the 126 (ny, nz) pairs with (ny / 64) * nz <= 4096


def test_concat_and_fexists(lib, tmp_path):
    lib.concat.restype = ctypes.c_void_p
    p = lib.concat(3, ctypes.c_char_p(b"ab"), ctypes.c_char_p(b""), ctypes.c_char_p(b"cde"))
    assert ctypes.string_at(p) == b"abcde"
    ctypes.CDLL(None).free(ctypes.c_void_p(p))
    f = tmp_path / "x.txt"
    assert not lib.fexists(str(f).encode())
    f.write_text("1")
    assert lib.fexists(str(f).encode())


def _exported_symbols():
    import subprocess
    from microimagelib_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    return [line.split()[-1] for line in out.splitlines() if line.strip()]


def test_only_the_c_abi_is_exported():
    """A drop-in library loaded next to torch / ImageJ's JNI glue exports libapi.h + mil_* and nothing else: no C++
    internals (_ZN3mil...), no HIP fat-binary bookkeeping (csrc/libapi.map)."""
    syms = _exported_symbols()
    assert syms, "nm listed no dynamic symbols"
    mangled = [s for s in syms if s.startswith("_Z")]
    assert not mangled, mangled[:5]
    declared = set(declared_functions("libapi.h")) | set(declared_functions("mil_device.h"))
    extra = [s for s in syms if s not in declared and not s.startswith("mil_")]
    assert not extra, extra
    assert declared <= set(syms)


def test_profiler_library_is_not_a_link_dependency():
    """roctx ranges are resolved with dlopen at run time (csrc/trace.h): a ROCm install without rocprofiler-sdk still
    loads the library."""
    import subprocess
    from microimagelib_amd import _lib
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    needed = [line for line in out.splitlines() if "NEEDED" in line]
    assert needed and not any("roctx" in n or "rocprofiler" in n for n in needed), needed
