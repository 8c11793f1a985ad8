// Clean-room Powell direction-set minimiser with the control flow, constants and
// float/double arithmetic of the reference optimiser (src/api_powell.c:119-360), but
// re-entrant: no file-scope state, the cost is a callable, vectors are 0-based.
#pragma once

#include <functional>

namespace mil {

using CostFn = std::function<float(const float *x)>;

struct PowellResult {
    int iterations = 0;  // outer iterations started (reference `*iter`)
    float fret = 0.f;    // best cost returned by the last line search
};

// p[n]            in/out parameter vector
// xi[n rows][ld]  in/out direction set, column j = j-th direction
// eval_count      the counter the cost function increments (reference `*totalIt`)
PowellResult powell_minimize(float *p, float *xi, int ld, int n, float ftol, const CostFn &cost,
                             const int *eval_count, int eval_limit);

}  // namespace mil
