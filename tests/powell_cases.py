"""Analytic cost functions shared by the Powell parity tests (float32 arithmetic,
1-based parameter vectors like the reference's NR arrays)."""
import numpy as np

f32 = np.float32


def make_cost(kind, n):
    """Returns (func(x_1based)->float32, counter[list], trace[list])."""
    counter = [0]
    trace = []

    def finish(v):
        counter[0] += 1
        trace.append(float(v))
        return v

    def quad(x):
        v = f32(0)
        for j in range(1, n + 1):
            d = f32(x[j] - f32(0.3 * j))
            v = f32(v + f32(f32(j) * f32(d * d)))
        return finish(f32(f32(v * f32(0.01)) - f32(0.9)))

    def rosen(x):
        v = f32(0)
        for j in range(1, n):
            a = f32(x[j + 1] - f32(x[j] * x[j]))
            b = f32(f32(1) - x[j])
            v = f32(f32(v + f32(f32(100) * f32(a * a))) + f32(b * b))
        return finish(f32(f32(v * f32(1e-3)) - f32(1)))

    def ridge(x):  # narrow curved valley + coupling: exercises direction replacement
        v = f32(0)
        s = f32(0)
        for j in range(1, n + 1):
            s = f32(s + x[j])
            d = f32(x[j] - f32(np.sin(f32(j))))
            v = f32(v + f32(d * d))
        v = f32(v + f32(f32(5) * f32(f32(s - f32(1)) * f32(s - f32(1)))))
        return finish(f32(f32(v * f32(0.02)) - f32(0.95)))

    def abort(x):  # starts above the 1.001 abort threshold
        return finish(f32(1.5))

    def zncc_like(x):  # smooth bump in [-1, 0): shape of a registration cost
        v = f32(0)
        for j in range(1, n + 1):
            d = f32(x[j] - f32(0.7 / j))
            v = f32(v + f32(d * d))
        return finish(f32(-np.exp(f32(-v * f32(0.5)))))

    return {"quad": quad, "rosen": rosen, "ridge": ridge, "abort": abort, "zncc": zncc_like}[kind], counter, trace


def start_point(n, value=-0.5):
    p = np.zeros(n + 1, dtype=f32)
    p[1:] = f32(value)
    xi = np.zeros((n + 1, n + 1), dtype=f32)
    for i in range(1, n + 1):
        xi[i, i] = 1
    return p, xi
