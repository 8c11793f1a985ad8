// Powell's direction-set method driving the registration cost.
//
// Behavioural contract = the reference optimiser (src/api_powell.c): identity start
// directions supplied by the caller, bracket from (0, 1), Brent line minimisation with
// tolerance 0.01 and 100 iterations, and the reference's three extra exits -- cost
// >= 1.001, evaluation budget exhausted, 100 outer iterations.  Variables are float;
// every place where C promotes to double in the reference (double literals, fabs) is
// promoted here too, so the trajectory is bit-identical (tests/test_powell.py checks
// that against the compiled reference).  Unlike the reference there is no global
// state: the line function is a closure, so any number of host threads may optimise
// concurrently (one per GPU in the batch driver).
#include "powell.h"

#include <cmath>
#include <utility>
#include <vector>

namespace mil {

namespace {

// line minimiser constants (reference src/api_powell.c:113-116)
constexpr int kBrentIterations = 100;
constexpr double kGoldenSection = 0.3819660;
constexpr double kAbsEps = 1.0e-10;
// bracketing constants (:189-191)
constexpr double kGrow = 1.618034;
constexpr double kGrowLimit = 100.0;
constexpr double kTinyDenominator = 1.0e-20;
// line tolerance (:254), outer iteration cap (:304), cost abort threshold (:317,332,356)
constexpr double kLineTolerance = 0.01;
constexpr int kOuterIterations = 100;
constexpr double kCostAbort = 1.001;

inline double with_sign(double magnitude, double sign_source) {
    return sign_source >= 0.0 ? std::fabs(magnitude) : -std::fabs(magnitude);
}

struct Triple {
    float ax, bx, cx;
    float fa, fb, fc;
};

// Downhill bracketing of a minimum of `f` starting from the points a, b (:194-247).
template <class Line>
Triple bracket(float a, float b, Line &&f) {
    Triple t;
    t.ax = a;
    t.bx = b;
    t.fa = f(t.ax);
    t.fb = f(t.bx);
    if (t.fb > t.fa) {
        std::swap(t.ax, t.bx);
        std::swap(t.fa, t.fb);
    }
    t.cx = (float)((double)t.bx + kGrow * (double)(t.bx - t.ax));
    t.fc = f(t.cx);
    while (t.fb > t.fc) {
        const float r = (t.bx - t.ax) * (t.fb - t.fc);
        const float q = (t.bx - t.cx) * (t.fb - t.fa);
        const float q_minus_r = q - r;
        const float mag_a = (float)std::fabs((double)q_minus_r);
        const float mag_b = (float)kTinyDenominator;
        const float mag = (mag_a > mag_b) ? mag_a : mag_b;
        const float numer = (t.bx - t.cx) * q - (t.bx - t.ax) * r;
        float u = (float)((double)t.bx - (double)numer / (2.0 * with_sign((double)mag, (double)q_minus_r)));
        const float ulim = (float)((double)t.bx + kGrowLimit * (double)(t.cx - t.bx));
        float fu;
        if ((double)((t.bx - u) * (u - t.cx)) > 0.0) {  // parabolic point between b and c
            fu = f(u);
            if (fu < t.fc) {
                t.ax = t.bx;
                t.bx = u;
                t.fa = t.fb;
                t.fb = fu;
                return t;
            } else if (fu > t.fb) {
                t.cx = u;
                t.fc = fu;
                return t;
            }
            u = (float)((double)t.cx + kGrow * (double)(t.cx - t.bx));
            fu = f(u);
        } else if ((double)((t.cx - u) * (u - ulim)) > 0.0) {  // between c and the limit
            fu = f(u);
            if (fu < t.fc) {
                t.bx = t.cx;
                t.cx = u;
                u = (float)((double)t.cx + kGrow * (double)(t.cx - t.bx));
                t.fb = t.fc;
                t.fc = fu;
                fu = f(u);
            }
        } else if ((double)((u - ulim) * (ulim - t.cx)) >= 0.0) {  // clip to the limit
            u = ulim;
            fu = f(u);
        } else {
            u = (float)((double)t.cx + kGrow * (double)(t.cx - t.bx));
            fu = f(u);
        }
        t.ax = t.bx;
        t.bx = t.cx;
        t.cx = u;
        t.fa = t.fb;
        t.fb = t.fc;
        t.fc = fu;
    }
    return t;
}

// Brent's parabolic/golden line minimisation inside the bracket (:119-184).  Returns the
// best function value; *xmin receives its abscissa.  Hitting the iteration cap returns the
// best point so far instead of failing (:180-182).
template <class Line>
float brent(float ax, float bx, float cx, Line &&f, float tol, float *xmin) {
    float a = (ax < cx ? ax : cx);
    float b = (ax > cx ? ax : cx);
    float x = bx, w = bx, v = bx;
    float fx = f(x);
    float fw = fx, fv = fx;
    float e = 0.0f, d = 0.0f;
    for (int it = 1; it <= kBrentIterations; ++it) {
        const float xm = (float)(0.5 * (double)(a + b));
        const float tol1 = (float)((double)tol * std::fabs((double)x) + kAbsEps);
        const float tol2 = (float)(2.0 * (double)tol1);
        if (std::fabs((double)(x - xm)) <= ((double)tol2 - 0.5 * (double)(b - a))) {
            *xmin = x;
            return fx;
        }
        bool golden = true;
        if (std::fabs((double)e) > (double)tol1) {
            float r = (x - w) * (fx - fv);
            float q = (x - v) * (fx - fw);
            float p = (x - v) * q - (x - w) * r;
            q = (float)(2.0 * (double)(q - r));
            if ((double)q > 0.0) p = -p;
            q = (float)std::fabs((double)q);
            const float etemp = e;
            e = d;
            const bool reject = std::fabs((double)p) >= std::fabs(0.5 * (double)q * (double)etemp) ||
                                p <= q * (a - x) || p >= q * (b - x);
            if (!reject) {
                if (q == 0) {  // reference guard ":149"; unreachable (reject is true when q == 0)
                    *xmin = x;
                    return fx;
                }
                d = p / q;
                const float u_try = x + d;
                if (u_try - a < tol2 || b - u_try < tol2) d = (float)with_sign((double)tol1, (double)(xm - x));
                golden = false;
            }
        }
        if (golden) {
            e = (x >= xm ? a - x : b - x);
            d = (float)(kGoldenSection * (double)e);
        }
        float u;
        if (std::fabs((double)d) >= (double)tol1)
            u = (float)(double)(x + d);
        else
            u = (float)((double)x + with_sign((double)tol1, (double)d));
        const float fu = f(u);
        if (fu <= fx) {
            if (u >= x) a = x; else b = x;
            v = w; w = x; x = u;
            fv = fw; fw = fx; fx = fu;
        } else {
            if (u < x) a = u; else b = u;
            if (fu <= fw || w == x) {
                v = w; w = u;
                fv = fw; fw = fu;
            } else if (fu <= fv || v == x || v == w) {
                v = u;
                fv = fu;
            }
        }
    }
    *xmin = x;
    return fx;
}

// Minimise along direction `dir` from `p`; both are updated (dir scaled by the step) (:273-301).
float line_minimize(float *p, float *dir, int n, const CostFn &cost, std::vector<float> &probe) {
    const std::vector<float> origin(p, p + n), direction(dir, dir + n);
    auto along = [&](float step) -> float {  // f1dim :260-270
        for (int j = 0; j < n; ++j) probe[j] = origin[j] + step * direction[j];
        return cost(probe.data());
    };
    const Triple t = bracket(0.0f, 1.0f, along);
    float step = 0.0f;
    const float best = brent(t.ax, t.bx, t.cx, along, (float)kLineTolerance, &step);
    for (int j = 0; j < n; ++j) {
        dir[j] *= step;
        p[j] += dir[j];
    }
    return best;
}

inline double squared_f(float a) { return a == 0.0 ? 0.0 : (double)(a * a); }  // SQR on a float static

}  // namespace

PowellResult powell_minimize(float *p, float *xi, int ld, int n, float ftol, const CostFn &cost,
                             const int *eval_count, int eval_limit) {
    PowellResult res;
    std::vector<float> pt(n), ptt(n), xit(n), probe(n);
    float fret = cost(p);
    res.fret = fret;
    if ((double)fret >= kCostAbort) return res;  // :317
    for (int j = 0; j < n; ++j) pt[j] = p[j];
    for (int iter = 1;; ++iter) {
        res.iterations = iter;
        const float fp = fret;
        int ibig = -1;
        float del = 0.0f;
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) xit[j] = xi[(size_t)j * ld + i];
            const float fptt = fret;
            fret = line_minimize(p, xit.data(), n, cost, probe);
            res.fret = fret;
            if (std::fabs((double)(fptt - fret)) > (double)del) {
                del = (float)std::fabs((double)(fptt - fret));
                ibig = i;
            }
            if (*eval_count >= eval_limit) return res;    // :331
            if ((double)fret >= kCostAbort) return res;   // :332
        }
        if (2.0 * std::fabs((double)(fp - fret)) <= (double)ftol * (std::fabs((double)fp) + std::fabs((double)fret)))
            return res;  // converged :334
        if (iter == kOuterIterations) return res;  // :340
        for (int j = 0; j < n; ++j) {
            ptt[j] = (float)(2.0 * (double)p[j] - (double)pt[j]);
            xit[j] = p[j] - pt[j];
            pt[j] = p[j];
        }
        const float fptt = cost(ptt.data());
        if (fptt < fp) {
            const float t = (float)(2.0 * ((double)fp - 2.0 * (double)fret + (double)fptt) * squared_f(fp - fret - del) -
                                    (double)del * squared_f(fp - fptt));
            if ((double)t < 0.0) {
                fret = line_minimize(p, xit.data(), n, cost, probe);
                res.fret = fret;
                if (ibig >= 0) {
                    for (int j = 0; j < n; ++j) {
                        xi[(size_t)j * ld + ibig] = xi[(size_t)j * ld + (n - 1)];
                        xi[(size_t)j * ld + (n - 1)] = xit[j];
                    }
                }
                if (*eval_count >= eval_limit) return res;   // :355
                if ((double)fret >= kCostAbort) return res;  // :356
            }
        }
    }
}

}  // namespace mil
