"""TIFF stack codec of libapi.h (gettifinfo / readtifstack* / writetifstack*): host logic, no GPU.
On-disk format of the reference (src/apifunc.cpp:140-326): baseline TIFF, one IFD per slice, one sample per
pixel, uncompressed, one strip per slice; 16-bit unsigned (written by C truncation, :255) or 32-bit IEEE float."""
import ctypes as C

import numpy as np
import pytest

from microimagelib_amd import api, _lib


def test_float32_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    vol = rng.normal(100, 50, (5, 7, 11)).astype(np.float32)
    p = str(tmp_path / "f32.tif")
    api.writetifstack(p, vol, 32)
    bits, size = api.gettifinfo(p)
    assert bits == 32 and tuple(size) == (11, 7, 5)
    assert np.array_equal(api.readtifstack(p), vol)


def test_uint16_truncates_like_a_c_cast(tmp_path):
    vol = np.array([[[0.0, 0.99, 1.5, 65535.0, 1234.999]]], dtype=np.float32)
    p = str(tmp_path / "u16.tif")
    api.writetifstack(p, vol, 16)
    bits, size = api.gettifinfo(p)
    assert bits == 16 and tuple(size) == (5, 1, 1)
    assert np.array_equal(api.readtifstack(p), np.array([[[0, 0, 1, 65535, 1234]]], dtype=np.float32))


def test_16to16_round_trip(tmp_path):
    lib = _lib.load()
    rng = np.random.default_rng(1)
    vol = rng.integers(0, 65536, (3, 4, 6), dtype=np.uint16)
    p = str(tmp_path / "u16raw.tif").encode()
    size = (C.c_uint * 3)(6, 4, 3)
    lib.writetifstack_16to16(p, vol.ctypes.data_as(C.POINTER(C.c_ushort)), size)
    back = np.zeros_like(vol)
    got = (C.c_uint * 3)()
    lib.readtifstack_16to16(back.ctypes.data_as(C.POINTER(C.c_ushort)), p, got)
    assert tuple(got) == (6, 4, 3) and np.array_equal(back, vol)


def test_files_are_plain_multi_page_tiff_readable_by_pillow(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(2)
    vol = rng.uniform(0, 1000, (4, 6, 9)).astype(np.float32)
    p = str(tmp_path / "pil32.tif")
    api.writetifstack(p, vol, 32)
    im = PIL.open(p)
    assert im.n_frames == 4 and im.size == (9, 6)
    for z in range(4):
        im.seek(z)
        page = np.array(im)
        assert page.dtype == np.float32 and np.array_equal(page, vol[z])
    # 16-bit: same tag set as the reference writer (src/apifunc.cpp:261-270), incl. its PLANARCONFIG_SEPARATE,
    # which Pillow's I;16 decoder refuses -- so check the tags and the raw strips instead of decoding through Pillow
    p = str(tmp_path / "pil16.tif")
    api.writetifstack(p, vol, 16)
    im = PIL.open(p)
    assert im.n_frames == 4 and im.size == (9, 6)
    raw = open(p, "rb").read()
    for z in range(4):
        im.seek(z)
        t = im.tag_v2
        assert t[258] == (16,) and t[259] == 1 and t[262] == 1 and t[277] == 1 and t[278] == 6 and t[284] == 2
        off, n = t[273][0], t[279][0]
        assert n == 9 * 6 * 2
        page = np.frombuffer(raw[off:off + n], dtype="<u2").reshape(6, 9)
        assert np.array_equal(page, np.floor(vol[z]).astype(np.uint16))


def test_reads_tiff_written_by_pillow(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 4000, (3, 5, 8), dtype=np.uint16)
    pages = [PIL.fromarray(vol[z]) for z in range(3)]
    p = str(tmp_path / "from_pil.tif")
    pages[0].save(p, save_all=True, append_images=pages[1:], compression=None)
    bits, size = api.gettifinfo(p)
    assert bits == 16 and tuple(size) == (8, 5, 3)
    assert np.array_equal(api.readtifstack(p), vol.astype(np.float32))
