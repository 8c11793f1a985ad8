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


# ---------------------------------------------------------------------------------------------------------------------
# What the reference reads through libtiff (TIFFOpen + TIFFReadScanline, src/apifunc.cpp:121-190) and microscopes / Fiji /
# Python write: BigTIFF, several strips per slice, PackBits / LZW / Deflate strips, predictors 2 and 3.  The files are
# built here byte by byte (independent encoders, below), and with Pillow where its libtiff offers the codec.
# ---------------------------------------------------------------------------------------------------------------------
import struct
import subprocess
import sys
import zlib


def _packbits(data: bytes) -> bytes:
    out, i, n = bytearray(), 0, len(data)
    while i < n:
        run = 1
        while i + run < n and run < 128 and data[i + run] == data[i]:
            run += 1
        if run >= 3:
            out += bytes([(257 - run) & 0xFF, data[i]])
            i += run
            continue
        j = i
        while j < n and j - i < 128:
            if j + 2 < n and data[j] == data[j + 1] == data[j + 2]:
                break
            j += 1
        out += bytes([j - i - 1]) + data[i:j]
        i = j
    return bytes(out)


def _lzw(data: bytes) -> bytes:
    """TIFF LZW encoder: MSB-first codes, 9..12 bits, early change, Clear (256) first and when the table fills, EOI (257)."""
    out, acc, nbits = bytearray(), 0, 0

    def emit(code, width):
        nonlocal acc, nbits
        acc = (acc << width) | code
        nbits += width
        while nbits >= 8:
            out.append((acc >> (nbits - 8)) & 0xFF)
            nbits -= 8

    table = {bytes([i]): i for i in range(256)}
    nxt, width = 258, 9
    emit(256, width)
    w = b""
    for b in data:
        wb = w + bytes([b])
        if wb in table:
            w = wb
            continue
        emit(table[w], width)
        table[wb] = nxt
        nxt += 1
        # the encoder's table runs one entry ahead of the decoder's, which widens when ITS next free code reaches 2^width - 1
        if nxt == (1 << width) and width < 12:
            width += 1
        if nxt >= 4094:
            emit(256, width)
            table = {bytes([i]): i for i in range(256)}
            nxt, width = 258, 9
        w = bytes([b])
    if w:
        emit(table[w], width)
        # the decoder adds an entry for this code too (and may widen) before it reads EOI
        nxt += 1
        if nxt == (1 << width) and width < 12:
            width += 1
    emit(257, width)
    if nbits:
        out.append((acc << (8 - nbits)) & 0xFF)
    return bytes(out)


def _predict_horizontal(rows: np.ndarray) -> np.ndarray:
    d = rows.copy()
    d[:, 1:] = rows[:, 1:] - rows[:, :-1]  # unsigned wrap-around
    return d


def _predict_float(rows: np.ndarray, big_endian: bool) -> bytes:
    """TIFF Technical Note 3: per row, byte planes (most significant first), then byte-wise differencing."""
    out = bytearray()
    for r in rows:
        be = r.astype(">f4").tobytes()
        w = len(r)
        planar = bytes(be[4 * x + b] for b in range(4) for x in range(w))
        diff = bytes([planar[0]] + [(planar[k] - planar[k - 1]) & 0xFF for k in range(1, len(planar))])
        out += diff
    return bytes(out)


def _write_stack(path, vol, compression=1, predictor=1, rows_per_strip=None, big=False, big_endian=False):
    """Multi-page TIFF / BigTIFF, one IFD per slice, strips of `rows_per_strip` rows, codec per strip."""
    E = ">" if big_endian else "<"
    nz, ny, nx = vol.shape
    bits = vol.dtype.itemsize * 8
    is_float = vol.dtype.kind == "f"
    rps = rows_per_strip or ny
    codec = {1: lambda b: b, 32773: _packbits, 5: _lzw, 8: lambda b: zlib.compress(b, 6), 32946: lambda b: zlib.compress(b, 6)}[compression]
    f = bytearray()
    f += (b"MM" if big_endian else b"II")
    if big:
        f += struct.pack(E + "HHHQ", 43, 8, 0, 0)
        link = 8
    else:
        f += struct.pack(E + "HI", 42, 0)
        link = 4
    for z in range(nz):
        offs, cnts = [], []
        for y0 in range(0, ny, rps):
            rows = vol[z, y0:y0 + rps]
            if predictor == 2:
                raw = _predict_horizontal(rows.view(f"u{vol.dtype.itemsize}")).astype(f"{E}u{vol.dtype.itemsize}").tobytes()
            elif predictor == 3:
                raw = _predict_float(rows, big_endian)
            else:
                raw = rows.astype(E + vol.dtype.str[1:]).tobytes()
            enc = codec(raw)
            offs.append(len(f))
            cnts.append(len(enc))
            f += enc
            if len(f) & 1:
                f += b"\0"
        ns = len(offs)
        OT, osz = (16, 8) if big else (4, 4)
        extra = {}

        def array_tag(values):
            if ns * osz <= (8 if big else 4):
                return values
            pos = len(f)
            f.extend(struct.pack(E + ("Q" if big else "I") * ns, *values))
            return pos
        o_val, c_val = array_tag(offs), array_tag(cnts)
        tags = [(256, 4, 1, nx), (257, 4, 1, ny), (258, 3, 1, bits), (259, 3, 1, compression), (262, 3, 1, 1),
                (273, OT, ns, o_val), (277, 3, 1, 1), (278, 4, 1, rps), (279, OT, ns, c_val)]
        if predictor != 1:
            tags.append((317, 3, 1, predictor))
        tags.append((339, 3, 1, 3 if is_float else 1))
        ifd = len(f)
        f[link:link + osz] = struct.pack(E + ("Q" if big else "I"), ifd)
        f += struct.pack(E + ("Q" if big else "H"), len(tags))
        for tag, typ, cnt, val in tags:
            f += struct.pack(E + "HH", tag, typ)
            f += struct.pack(E + ("Q" if big else "I"), cnt)
            width = 8 if big else 4
            if isinstance(val, list):  # inline array
                body = struct.pack(E + ("Q" if typ == 16 else "I") * cnt, *val)
            elif cnt > 1 or typ in (4, 16):
                body = struct.pack(E + ("Q" if (typ == 16 or big and cnt > 1) else "I"), val) if not (cnt > 1 and not big) else struct.pack(E + "I", val)
            else:
                body = struct.pack(E + "H", val)
            f += body.ljust(width, b"\0")
        link = len(f)
        f += struct.pack(E + ("Q" if big else "I"), 0)
    with open(path, "wb") as fh:
        fh.write(bytes(f))


_CODECS = [(1, "none"), (32773, "packbits"), (5, "lzw"), (8, "deflate"), (32946, "old-deflate")]


@pytest.mark.parametrize("compression,name", _CODECS)
@pytest.mark.parametrize("big", [False, True])
def test_reads_compressed_strips_and_bigtiff(tmp_path, compression, name, big):
    rng = np.random.default_rng(5)
    # low-entropy 16-bit data (runs and repeats: every codec path incl. LZW table growth) and a float stack
    u16 = (rng.integers(0, 40, (3, 37, 53)) * (rng.integers(0, 4, (3, 37, 53)) == 0)).astype(np.uint16) + 100
    f32 = rng.normal(500, 100, (2, 19, 31)).astype(np.float32)
    f32[:, 5:9] = 7.25  # runs
    for vol, rps in ((u16, 8), (u16, None), (f32, 5)):
        for big_endian in (False, True):
            p = str(tmp_path / f"{name}_{vol.dtype}_{rps}_{big}_{big_endian}.tif")
            _write_stack(p, vol, compression=compression, rows_per_strip=rps, big=big, big_endian=big_endian)
            bits, size = api.gettifinfo(p)
            assert bits == vol.dtype.itemsize * 8 and tuple(size) == vol.shape[::-1]
            assert np.array_equal(api.readtifstack(p), vol.astype(np.float32)), p


@pytest.mark.parametrize("compression", [5, 8])
def test_predictors(tmp_path, compression):
    rng = np.random.default_rng(6)
    u16 = np.cumsum(rng.integers(0, 5, (2, 23, 41)), axis=2).astype(np.uint16)  # smooth rows: what predictor 2 is for
    f32 = np.cumsum(rng.normal(0, 1, (2, 11, 29)), axis=2).astype(np.float32)
    for big_endian in (False, True):
        p = str(tmp_path / f"pred2_u16_{big_endian}.tif")
        _write_stack(p, u16, compression=compression, predictor=2, rows_per_strip=7, big_endian=big_endian)
        assert np.array_equal(api.readtifstack(p), u16.astype(np.float32))
        p = str(tmp_path / f"pred2_f32_{big_endian}.tif")
        _write_stack(p, f32, compression=compression, predictor=2, rows_per_strip=4, big_endian=big_endian)
        assert np.array_equal(api.readtifstack(p), f32)
        p = str(tmp_path / f"pred3_f32_{big_endian}.tif")
        _write_stack(p, f32, compression=compression, predictor=3, rows_per_strip=4, big_endian=big_endian)
        assert np.array_equal(api.readtifstack(p), f32)


def test_reads_what_pillow_compresses(tmp_path):
    """The same codecs produced by an independent implementation (Pillow's libtiff), where this Pillow build has them."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(8)
    vol = (rng.integers(0, 30, (3, 40, 64)) * 17).astype(np.uint16)
    pages = [PIL.fromarray(vol[z]) for z in range(3)]
    done = 0
    for comp in ("tiff_lzw", "tiff_adobe_deflate", "packbits"):
        p = str(tmp_path / f"pil_{comp}.tif")
        try:
            pages[0].save(p, save_all=True, append_images=pages[1:], compression=comp)
        except Exception:  # noqa: BLE001 -- codec not compiled into this Pillow
            continue
        assert np.array_equal(api.readtifstack(p), vol.astype(np.float32)), comp
        done += 1
    try:
        p = str(tmp_path / "pil_big.tif")
        pages[0].save(p, save_all=True, append_images=pages[1:], big_tiff=True)
        if open(p, "rb").read(4)[2] == 43:
            assert np.array_equal(api.readtifstack(p), vol.astype(np.float32))
            done += 1
    except Exception:  # noqa: BLE001
        pass
    if done == 0:
        pytest.skip("this Pillow writes none of the compressed / BigTIFF forms")


def test_writer_switches_to_bigtiff(tmp_path, monkeypatch):
    """Stacks beyond 4 GiB need BigTIFF (libtiff's classic writer fails there); forced here on a small stack."""
    rng = np.random.default_rng(9)
    vol = rng.uniform(0, 3000, (4, 9, 13)).astype(np.float32)
    monkeypatch.setenv("MIL_TIFF_BIG", "1")
    for bits in (16, 32):
        p = str(tmp_path / f"big{bits}.tif")
        api.writetifstack(p, vol, bits)
        head = open(p, "rb").read(16)
        assert head[:4] == b"II+\x00" and struct.unpack("<HHQ", head[4:16])[:2] == (8, 0)
        want = vol if bits == 32 else np.floor(vol).astype(np.float32)
        assert np.array_equal(api.readtifstack(p), want)
    PIL = pytest.importorskip("PIL.Image")
    try:
        im = PIL.open(str(tmp_path / "big32.tif"))
        assert im.n_frames == 4
        im.seek(3)
        assert np.array_equal(np.array(im), vol[3])
    except Exception:  # noqa: BLE001 -- a Pillow without BigTIFF support
        pass


_MALFORMED_PROBE = r"""
import sys
sys.path.insert(0, sys.argv[2])
from microimagelib_amd import api
api.readtifstack(sys.argv[1])
print("READ-OK")
"""


def test_malformed_compressed_files_are_refused_not_overrun(tmp_path):
    """Truncated / corrupt compressed strips, tiles and absurd directory counts end in the reader's error exit (the
    reference's I/O layer exits on unreadable input too, src/apifunc.cpp:117-120) -- never in a crash or an overrun."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(10)
    vol = (rng.integers(0, 9, (2, 16, 24)) * 11).astype(np.uint16)
    cases = {}
    for comp, name in ((5, "lzw"), (8, "deflate"), (32773, "packbits")):
        p = str(tmp_path / f"ok_{name}.tif")
        _write_stack(p, vol, compression=comp, rows_per_strip=8)
        raw = bytearray(open(p, "rb").read())
        cut = bytes(raw[:len(raw) - 40])
        cases[f"truncated_{name}"] = cut
        junk = bytearray(raw)
        for i in range(8, 60):
            junk[i] ^= 0x5A  # strip payload
        cases[f"corrupt_{name}"] = bytes(junk)
    p = str(tmp_path / "ok_big.tif")
    _write_stack(p, vol, big=True)
    raw = bytearray(open(p, "rb").read())
    ifd = struct.unpack("<Q", raw[8:16])[0]
    raw[ifd:ifd + 8] = struct.pack("<Q", 1 << 40)  # directory entry count
    cases["bigtiff_absurd_count"] = bytes(raw)
    for name, blob in cases.items():
        q = str(tmp_path / f"{name}.tif")
        open(q, "wb").write(blob)
        r = subprocess.run([sys.executable, "-c", _MALFORMED_PROBE, q, root], capture_output=True, text=True, timeout=120)
        if name.startswith("corrupt_packbits"):
            continue  # any byte string is a valid PackBits stream; wrong data, right length, no error possible
        assert r.returncode == 1 and "TIFF read error" in r.stderr and "READ-OK" not in r.stdout, (name, r.returncode, r.stderr[-300:])
