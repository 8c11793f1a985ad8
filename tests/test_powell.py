"""Powell optimiser parity (CPU only): the product's csrc/powell.cpp (through the
mil_powell C ABI) and the oracle restatement are compared BIT FOR BIT -- parameters,
direction set, fret, outer-iteration count, evaluation count and the full evaluation
trace -- with oracle/_ref/libpowell_ref.so, compiled from the reference's own
src/api_powell.c."""
import ctypes as C

import numpy as np
import pytest

from oracle import powell_oracle as po
from tests.powell_cases import make_cost, start_point

f32 = np.float32
CASES = [("quad", 3), ("quad", 6), ("quad", 12), ("rosen", 3), ("rosen", 6), ("rosen", 12), ("ridge", 9),
         ("ridge", 12), ("zncc", 6), ("zncc", 12), ("abort", 4)]


def run_product(lib, kind, n, ftol, it_limit, start=-0.5, ld_extra=0):
    from microimagelib_amd import _lib
    func, counter, trace = make_cost(kind, n)
    p1, xi1 = start_point(n, start)
    ld = n + ld_extra
    p = np.ascontiguousarray(p1[1:])
    xi = np.zeros((ld, ld), dtype=f32)
    xi[:n, :n] = xi1[1:, 1:]
    total = C.c_int(0)

    def cb(xptr, _user):
        x = np.zeros(n + 1, dtype=f32)
        for j in range(n):
            x[j + 1] = xptr[j]
        v = float(func(x))
        total.value = counter[0]
        return v

    it = C.c_int(0)
    fret = C.c_float(0)
    rc = lib.mil_powell(p.ctypes.data_as(_lib.c_float_p), xi.ctypes.data_as(_lib.c_float_p), ld, n, float(f32(ftol)),
                        C.byref(it), C.byref(fret), _lib.COST_FN(cb), None, C.byref(total), it_limit)
    assert rc == 0
    return dict(p=p, xi=xi[:n, :n].copy(), fret=f32(fret.value), it=it.value, evals=counter[0], trace=trace)


def run_py(impl, kind, n, ftol, it_limit, start=-0.5):
    func, counter, trace = make_cost(kind, n)
    p, xi = start_point(n, start)
    it, fret = impl(p, xi, n, ftol, func, counter, it_limit)
    return dict(p=p[1:].copy(), xi=xi[1:, 1:].copy(), fret=f32(fret), it=it, evals=counter[0], trace=trace)


def assert_same(a, b, what):
    assert a["evals"] == b["evals"], what
    assert a["trace"] == b["trace"], what
    assert np.array_equal(a["p"].view(np.uint32), b["p"].view(np.uint32)), what
    assert np.array_equal(a["xi"].view(np.uint32), b["xi"].view(np.uint32)), what
    assert np.float32(a["fret"]).view(np.uint32) == np.float32(b["fret"]).view(np.uint32), what


needs_ref = pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libpowell_ref.so not built")


@needs_ref
@pytest.mark.parametrize("kind,n", CASES)
def test_restatement_matches_compiled_reference(kind, n):
    a = run_py(po.powell, kind, n, 1e-4, 3000)
    b = run_py(po.ref_powell, kind, n, 1e-4, 3000)
    assert_same(a, b, f"oracle restatement vs reference api_powell.c ({kind}, n={n})")
    assert a["it"] == b["it"] or kind == "abort"  # the reference leaves *iter unset on the early abort


@needs_ref
@pytest.mark.parametrize("kind,n", CASES)
def test_product_matches_compiled_reference(lib, kind, n):
    a = run_product(lib, kind, n, 1e-4, 3000)
    b = run_py(po.ref_powell, kind, n, 1e-4, 3000)
    assert_same(a, b, f"csrc/powell.cpp vs reference api_powell.c ({kind}, n={n})")


@pytest.mark.parametrize("kind,n", CASES)
def test_product_matches_restatement(lib, kind, n):
    """Runs everywhere (also where the reference tree / _ref build is absent)."""
    a = run_product(lib, kind, n, 1e-4, 3000, ld_extra=3)  # leading dimension > n (cascade layout)
    b = run_py(po.powell, kind, n, 1e-4, 3000)
    assert_same(a, b, f"csrc/powell.cpp vs oracle restatement ({kind}, n={n})")


@needs_ref
@pytest.mark.parametrize("it_limit", [1, 7, 40])
def test_evaluation_budget_exit(lib, it_limit):
    """The reference checks the budget only after each line search (src/api_powell.c:331,355)."""
    a = run_product(lib, "rosen", 6, 1e-6, it_limit)
    b = run_py(po.ref_powell, "rosen", 6, 1e-6, it_limit)
    assert_same(a, b, f"budget {it_limit}")
    assert a["evals"] >= it_limit


@needs_ref
@pytest.mark.parametrize("ftol", [1e-2, 5e-3, 1e-6])
def test_tolerances(lib, ftol):
    a = run_product(lib, "zncc", 9, ftol, 3000, start=0.2)
    b = run_py(po.ref_powell, "zncc", 9, ftol, 3000, start=0.2)
    assert_same(a, b, f"ftol {ftol}")
