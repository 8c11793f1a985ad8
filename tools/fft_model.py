"""NumPy model of the device FFT pipeline (development aid, CPU only).

Mirrors csrc/fft.hip stage by stage: in-place mixed-radix DIF (natural in, digit-reversed out)
and its transpose DIT (digit-reversed in, natural out), the packed real<->complex x transform,
and the 3-pass convolution with a privately ordered OTF.
"""
import numpy as np


def pick_radix(n):
    for r in (8, 4, 2, 3, 5, 7, 11, 13):
        if n % r == 0:
            return r
    raise ValueError(n)


def plan(n):
    r = []
    while n > 1:
        r.append(pick_radix(n)); n //= r[-1]
    return r


def freq_map(n, radices):
    """freq[p] = frequency index held at position p after the DIF stages."""
    if not radices:
        return np.zeros(1, dtype=np.int64)
    r = radices[0]; m = n // r
    sub = freq_map(m, radices[1:])
    p = np.arange(n)
    return r * sub[p % m] + p // m


def dft_small(v, sign):
    r = v.shape[0]
    k = np.arange(r)
    w = np.exp(sign * 2j * np.pi * np.outer(k, k) / r)
    return np.tensordot(w, v, axes=(1, 0))


def stages(n, radices):
    out = []; L = n
    for r in radices:
        out.append((L, r)); L //= r
    return out


def fft_dif(x, sign):
    """x: [n, ...] complex, transforms axis 0 in place order; returns digit-reversed output."""
    x = x.astype(np.complex128).copy(); n = x.shape[0]
    for (L, r) in stages(n, plan(n)):
        m = L // r
        for blk in range(n // L):
            for i in range(m):
                idx = blk * L + i + m * np.arange(r)
                v = dft_small(x[idx], sign)
                tw = np.exp(sign * 2j * np.pi * i * np.arange(r) / L)
                x[idx] = v * tw.reshape((r,) + (1,) * (x.ndim - 1))
    return x


def fft_dit(x, sign):
    """x: digit-reversed-position input; returns natural-order transform."""
    x = x.astype(np.complex128).copy(); n = x.shape[0]
    for (L, r) in reversed(stages(n, plan(n))):
        m = L // r
        for blk in range(n // L):
            for i in range(m):
                idx = blk * L + i + m * np.arange(r)
                tw = np.exp(sign * 2j * np.pi * i * np.arange(r) / L)
                x[idx] = dft_small(x[idx] * tw.reshape((r,) + (1,) * (x.ndim - 1)), sign)
    return x


def x_forward(rows):
    """rows [nrows, nx] real -> [nrows, nx/2+1] complex (natural order), via packed complex FFT."""
    nx = rows.shape[1]; h = nx // 2
    z = rows[:, 0::2] + 1j * rows[:, 1::2]
    fm = freq_map(h, plan(h)); pos = np.argsort(fm)  # pos[n]: position of element n
    tile = np.zeros((h, rows.shape[0]), complex)
    tile[pos[np.arange(h)]] = z.T                     # element n at position pos(n)
    Z = fft_dit(tile, -1)                             # natural order
    k = np.arange(h + 1)
    Zk = Z[k % h]; Zc = np.conj(Z[(h - k) % h])
    w = np.exp(-2j * np.pi * k / nx)[:, None]
    X = 0.5 * (Zk + Zc) - 0.5j * w * (Zk - Zc)
    return X.T


def x_inverse(spec):
    """spec [nrows, nx/2+1] -> [nrows, nx] real, unnormalised (= nx * x)."""
    h = spec.shape[1] - 1; nx = 2 * h
    X = spec.T
    k = np.arange(h)
    Xk = X[k]; Xc = np.conj(X[h - k])
    w = np.exp(2j * np.pi * k / nx)[:, None]
    Zp = (Xk + Xc) + 1j * w * (Xk - Xc)
    out = fft_dif(Zp, +1)                             # position p holds z[freq(p)]
    fm = freq_map(h, plan(h)); pos = np.argsort(fm)
    z = out[pos]                                      # z[n] = out[pos(n)]
    rows = np.empty((spec.shape[0], nx))
    rows[:, 0::2] = z.real.T; rows[:, 1::2] = z.imag.T
    return rows


def rfft3_private(vol):
    """vol [nz, ny, nx] -> spectrum [kz_s, ky_s, kx] (y, z digit-scrambled, x natural)."""
    nz, ny, nx = vol.shape
    s = x_forward(vol.reshape(nz * ny, nx)).reshape(nz, ny, nx // 2 + 1)
    s = np.moveaxis(fft_dif(np.moveaxis(s, 1, 0), -1), 0, 1)
    s = fft_dif(s, -1)
    return s


def irfft3_private(s):
    nz, ny, nxh1 = s.shape
    s = fft_dit(s, +1)
    s = np.moveaxis(fft_dit(np.moveaxis(s, 1, 0), +1), 0, 1)
    return x_inverse(s.reshape(nz * ny, nxh1)).reshape(nz, ny, 2 * (nxh1 - 1))


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (8, 16, 32, 64, 128, 256, 512, 1024, 96, 160, 192, 224, 320, 352, 416, 448, 576, 960):
        x = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
        fm = freq_map(n, plan(n))
        want = np.fft.fft(x, axis=0)
        got = fft_dif(x, -1)
        assert np.allclose(got, want[fm]), n
        pos = np.argsort(fm)
        v = np.zeros_like(x); v[pos] = x
        assert np.allclose(fft_dit(v, -1), want), n
        assert np.allclose(fft_dit(fft_dif(x, -1), +1), n * x), n
    for shape in ((16, 16, 16), (16, 32, 64), (32, 16, 128), (16, 192, 320)):
        vol = rng.standard_normal(shape)
        nz, ny, nx = shape
        S = rfft3_private(vol)
        fz = freq_map(nz, plan(nz)); fy = freq_map(ny, plan(ny))
        want = np.fft.rfftn(vol)
        assert np.allclose(S, want[fz][:, fy]), shape
        back = irfft3_private(S)
        assert np.allclose(back, vol * vol.size), shape
    print("fft model ok")
