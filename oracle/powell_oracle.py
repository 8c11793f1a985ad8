"""ORACLE (test infrastructure, not product): restatement of the reference's Powell
direction-set optimiser, src/api_powell.c:119-360 (brent / mnbrak / f1dim /
linmin / powell), reproducing the C arithmetic exactly: ``float`` variables,
``double`` wherever C's usual arithmetic conversions promote (double literals,
``fabs``), x86-64 SSE evaluation (FLT_EVAL_METHOD 0).

PARITY STATUS: **pinned** -- tests/test_powell.py checks this restatement and the
product's csrc/powell.cpp bit-for-bit (parameters, directions, fret, evaluation
count and the evaluation trace) against ``oracle/_ref/libpowell_ref.so``, which
oracle/Makefile compiles from the reference's own src/api_powell.c.

Vectors and matrices are 1-based float32 arrays like the reference's NR
allocators (index 0 unused).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

f32 = np.float32
f64 = np.float64

# brent  (src/api_powell.c:113-116)
ITMAX_BRENT = 100
CGOLD = 0.3819660
ZEPS = 1.0e-10
# mnbrak (:189-191)
GOLD = 1.618034
GLIMIT = 100.0
TINY = 1.0e-20
# linmin (:254)
TOL = 0.01
# powell (:304)
ITMAX_POWELL = 100
COST_LIMIT = 1.001


def _sign(a, b):
    """#define SIGN(a,b) ((b) >= 0.0 ? fabs(a) : -fabs(a))  -- double result."""
    return abs(f64(a)) if f64(b) >= 0.0 else -abs(f64(a))


def brent(ax, bx, cx, f, tol):
    """src/api_powell.c:119-184.  Returns (fx, xmin)."""
    ax, bx, cx, tol = f32(ax), f32(bx), f32(cx), f32(tol)
    e = f32(0.0)
    d = f32(0.0)
    a = ax if ax < cx else cx
    b = ax if ax > cx else cx
    x = w = v = bx
    fw = fv = fx = f32(f(x))
    for _ in range(1, ITMAX_BRENT + 1):
        xm = f32(0.5 * f64(f32(a + b)))
        tol1 = f32(f64(tol) * abs(f64(x)) + ZEPS)
        tol2 = f32(2.0 * f64(tol1))
        if abs(f64(f32(x - xm))) <= (f64(tol2) - 0.5 * f64(f32(b - a))):
            return fx, x
        if abs(f64(e)) > f64(tol1):
            r = f32(f32(x - w) * f32(fx - fv))
            q = f32(f32(x - v) * f32(fx - fw))
            p = f32(f32(f32(x - v) * q) - f32(f32(x - w) * r))
            q = f32(2.0 * f64(f32(q - r)))
            if f64(q) > 0.0:
                p = f32(-p)
            q = f32(abs(f64(q)))
            etemp = e
            e = d
            if (abs(f64(p)) >= abs(0.5 * f64(q) * f64(etemp))
                    or p <= f32(q * f32(a - x)) or p >= f32(q * f32(b - x))):
                e = f32(a - x) if x >= xm else f32(b - x)
                d = f32(CGOLD * f64(e))
            else:
                if q == 0:
                    return fx, None  # "added by min": returns without setting *xmin
                d = f32(p / q)
                u = f32(x + d)
                if f32(u - a) < tol2 or f32(b - u) < tol2:
                    d = f32(_sign(tol1, f32(xm - x)))
        else:
            e = f32(a - x) if x >= xm else f32(b - x)
            d = f32(CGOLD * f64(e))
        if abs(f64(d)) >= f64(tol1):
            u = f32(f64(f32(x + d)))
        else:
            u = f32(f64(x) + _sign(tol1, d))
        fu = f32(f(u))
        if fu <= fx:
            if u >= x:
                a = x
            else:
                b = x
            v, w, x = w, x, u
            fv, fw, fx = fw, fx, fu
        else:
            if u < x:
                a = u
            else:
                b = u
            if fu <= fw or w == x:
                v, w, fv, fw = w, u, fw, fu
            elif fu <= fv or v == x or v == w:
                v, fv = u, fu
    return fx, x


def mnbrak(ax, bx, f):
    """src/api_powell.c:194-247.  Returns (ax, bx, cx, fa, fb, fc)."""
    ax, bx = f32(ax), f32(bx)
    fa = f32(f(ax))
    fb = f32(f(bx))
    if fb > fa:
        ax, bx = bx, ax
        fa, fb = fb, fa
    cx = f32(f64(bx) + GOLD * f64(f32(bx - ax)))
    fc = f32(f(cx))
    while fb > fc:
        r = f32(f32(bx - ax) * f32(fb - fc))
        q = f32(f32(bx - cx) * f32(fb - fa))
        qr = f32(q - r)
        fmax = max(f32(abs(f64(qr))), f32(TINY))  # FMAX on float statics
        num = f32(f32(f32(bx - cx) * q) - f32(f32(bx - ax) * r))
        u = f32(f64(bx) - f64(num) / (2.0 * _sign(fmax, qr)))
        ulim = f32(f64(bx) + GLIMIT * f64(f32(cx - bx)))
        if f64(f32(f32(bx - u) * f32(u - cx))) > 0.0:
            fu = f32(f(u))
            if fu < fc:
                return u_ret(bx, u, cx, fb, fu, fc)
            elif fu > fb:
                return ax, bx, u, fa, fb, fu
            u = f32(f64(cx) + GOLD * f64(f32(cx - bx)))
            fu = f32(f(u))
        elif f64(f32(f32(cx - u) * f32(u - ulim))) > 0.0:
            fu = f32(f(u))
            if fu < fc:
                bx, cx = cx, u
                u = f32(f64(cx) + GOLD * f64(f32(cx - bx)))
                fb, fc = fc, fu
                fu = f32(f(u))
        elif f64(f32(f32(u - ulim) * f32(ulim - cx))) >= 0.0:
            u = ulim
            fu = f32(f(u))
        else:
            u = f32(f64(cx) + GOLD * f64(f32(cx - bx)))
            fu = f32(f(u))
        ax, bx, cx = bx, cx, u
        fa, fb, fc = fb, fc, fu
    return ax, bx, cx, fa, fb, fc


def u_ret(bx, u, cx, fb, fu, fc):
    """mnbrak early exit `*ax=*bx; *bx=u; *fa=*fb; *fb=fu; return` (:213-218)."""
    return bx, u, cx, fb, fu, fc


def linmin(p, xi, n, func):
    """src/api_powell.c:273-301.  Modifies p and xi (1-based) in place; returns fret."""
    pcom = p.copy()
    xicom = xi.copy()

    def f1dim(x):  # :260-270
        xt = np.zeros(n + 1, dtype=f32)
        x = f32(x)
        for j in range(1, n + 1):
            xt[j] = f32(pcom[j] + f32(x * xicom[j]))
        return func(xt)

    ax, xx, bx, _fa, _fx, _fb = mnbrak(0.0, 1.0, f1dim)
    fret, xmin = brent(ax, xx, bx, f1dim, TOL)
    if xmin is None:
        raise RuntimeError("brent q==0 early exit leaves xmin unset in the reference")
    for j in range(1, n + 1):
        xi[j] = f32(xi[j] * xmin)
        p[j] = f32(p[j] + xi[j])
    return fret


def powell(p, xi, n, ftol, func, counter, it_limit):
    """src/api_powell.c:305-360.  ``p``: float32[n+1...] 1-based, ``xi``: float32 2-D
    1-based direction matrix (columns are directions); both modified in place.
    ``counter`` is the one-element list the cost function increments (the reference's
    global ``itNumStatic`` passed as ``*totalIt``).  Returns (iter, fret)."""
    ftol = f32(ftol)
    fret = f32(func(p))
    if f64(fret) >= COST_LIMIT:
        return 0, fret
    pt = p.copy()
    ptt = np.zeros_like(p)
    xit = np.zeros_like(p)
    it = 1
    while True:
        fp = fret
        ibig = 0
        dele = f32(0.0)
        for i in range(1, n + 1):
            for j in range(1, n + 1):
                xit[j] = xi[j, i]
            fptt = fret
            fret = f32(linmin(p, xit, n, func))
            if abs(f64(f32(fptt - fret))) > f64(dele):
                dele = f32(abs(f64(f32(fptt - fret))))
                ibig = i
            if counter[0] >= it_limit:
                return it, fret
            if f64(fret) >= COST_LIMIT:
                return it, fret
        if 2.0 * abs(f64(f32(fp - fret))) <= f64(ftol) * (abs(f64(fp)) + abs(f64(fret))):
            return it, fret
        if it == ITMAX_POWELL:
            return it, fret
        for j in range(1, n + 1):
            ptt[j] = f32(2.0 * f64(p[j]) - f64(pt[j]))
            xit[j] = f32(p[j] - pt[j])
            pt[j] = p[j]
        fptt = f32(func(ptt))
        if fptt < fp:
            s1 = f32(f32(fp - fret) - dele)
            sq1 = 0.0 if s1 == 0.0 else f64(f32(s1 * s1))
            s2 = f32(fp - fptt)
            sq2 = 0.0 if s2 == 0.0 else f64(f32(s2 * s2))
            t = f32(2.0 * (f64(fp) - 2.0 * f64(fret) + f64(fptt)) * sq1 - f64(dele) * sq2)
            if f64(t) < 0.0:
                fret = f32(linmin(p, xit, n, func))
                for j in range(1, n + 1):
                    xi[j, ibig] = xi[j, n]
                    xi[j, n] = xit[j]
                if counter[0] >= it_limit:
                    return it, fret
                if f64(fret) >= COST_LIMIT:
                    return it, fret
        it += 1


# --------------------------------------------------------------------------
# the real thing: oracle/_ref/libpowell_ref.so  (reference src/api_powell.c)
# --------------------------------------------------------------------------
_REF_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libpowell_ref.so")
_COSTFN = ctypes.CFUNCTYPE(ctypes.c_float, ctypes.POINTER(ctypes.c_float))


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


def ref_powell(p, xi, n, ftol, func, counter, it_limit):
    """Same contract as :func:`powell`, executed by the compiled reference."""
    lib = ctypes.CDLL(_REF_PATH)
    lib.powell.restype = None
    lib.powell.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.POINTER(ctypes.c_float)),
                           ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_int),
                           ctypes.POINTER(ctypes.c_float), _COSTFN, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    ncol = xi.shape[1]
    cp = (ctypes.c_float * len(p))(*[float(v) for v in p])
    flat = (ctypes.c_float * xi.size)(*[float(v) for v in xi.reshape(-1)])
    rows = (ctypes.POINTER(ctypes.c_float) * xi.shape[0])()
    base = ctypes.addressof(flat)
    for i in range(xi.shape[0]):
        rows[i] = ctypes.cast(base + 4 * i * ncol, ctypes.POINTER(ctypes.c_float))
    total = ctypes.c_int(counter[0])

    def cb(xptr):
        x = np.array([0.0] + [xptr[j] for j in range(1, n + 1)], dtype=f32)
        total.value = counter[0]
        v = float(func(x))
        total.value = counter[0]
        return v

    it = ctypes.c_int(0)
    fret = ctypes.c_float(0)
    lib.powell(cp, rows, n, float(f32(ftol)), ctypes.byref(it), ctypes.byref(fret), _COSTFN(cb),
               ctypes.byref(total), int(it_limit))
    p[:] = np.frombuffer(cp, dtype=f32)
    xi[:] = np.frombuffer(flat, dtype=f32).reshape(xi.shape)
    return it.value, f32(fret.value)
