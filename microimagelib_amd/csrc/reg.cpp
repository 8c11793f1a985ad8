#include "reg.h"
#include "trace.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <vector>

#include "decon.h"
#include "powell.h"

namespace mil {

static thread_local int g_last_cost_launches = 0;
int reg3d_last_cost_launches() { return g_last_cost_launches; }

namespace {
struct EvalTrace {
    std::vector<float> rows;  // 13 floats per stored evaluation
    int capacity = 0, seen = 0;
};
thread_local EvalTrace g_trace;
inline void trace_evaluation(const float *m12, float cost) {
    EvalTrace &t = g_trace;
    if (t.capacity <= 0) return;
    if (t.seen < t.capacity) {
        t.rows.insert(t.rows.end(), m12, m12 + 12);
        t.rows.push_back(cost);
    }
    ++t.seen;
}
}  // namespace

void reg3d_trace_begin(int capacity) {
    g_trace.rows.clear();
    g_trace.seen = 0;
    g_trace.capacity = capacity > 0 ? capacity : 0;
    if (capacity > 0) g_trace.rows.reserve((size_t)13 * (size_t)capacity);
}

int reg3d_trace_read(float *out13, int max_entries) {
    const int stored = (int)(g_trace.rows.size() / 13);
    const int n = max_entries < stored ? max_entries : stored;
    if (out13 && n > 0) std::memcpy(out13, g_trace.rows.data(), (size_t)n * 13 * sizeof(float));
    return g_trace.seen;
}

// ---------------------------------------------------------------------------
// matrix parameterisations
// ---------------------------------------------------------------------------
void p2matrix(float *m, const float *p) {  // translations first, then the 3x3 block row by row
    m[3] = p[0]; m[7] = p[1]; m[11] = p[2];
    m[0] = p[3]; m[1] = p[4]; m[2] = p[5];
    m[4] = p[6]; m[5] = p[7]; m[6] = p[8];
    m[8] = p[9]; m[9] = p[10]; m[10] = p[11];
}

void matrix2p(float *p, const float *m) {
    p[0] = m[3]; p[1] = m[7]; p[2] = m[11];
    p[3] = m[0]; p[4] = m[1]; p[5] = m[2];
    p[6] = m[4]; p[7] = m[5]; p[8] = m[6];
    p[9] = m[8]; p[10] = m[9]; p[11] = m[10];
}

void matrixmultiply(float *m, const float *a, const float *b) {
    float out[12];
    for (int r = 0; r < 3; ++r) {
        const float a0 = a[4 * r], a1 = a[4 * r + 1], a2 = a[4 * r + 2], a3 = a[4 * r + 3];
        out[4 * r + 0] = a0 * b[0] + a1 * b[4] + a2 * b[8];
        out[4 * r + 1] = a0 * b[1] + a1 * b[5] + a2 * b[9];
        out[4 * r + 2] = a0 * b[2] + a1 * b[6] + a2 * b[10];
        out[4 * r + 3] = a0 * b[3] + a1 * b[7] + a2 * b[11] + a3;
    }
    std::memcpy(m, out, sizeof out);
}

namespace {

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void set_identity(float *m) {
    for (int i = 0; i < 12; ++i) m[i] = 0.f;
    m[0] = m[5] = m[10] = 1.f;
}

}  // namespace

void dof9tomatrix(float *m, const float *p, int dof_num) {
    // p: x y z alpha beta theta a b c; only the first dof_num entries are read, the rest take
    // their defaults (angles 0, scales 1).  M = [diag(a,b,c) | t] * Rz(alpha) * Rx(beta) * Ry(theta).
    float x = p[0], y = p[1], z = p[2];
    float alpha = 0.f, beta = 0.f, theta = 0.f, a = 1.f, b = 1.f, c = 1.f;
    if (dof_num >= 6) {  // degrees / 57.3 evaluated in double, stored as float
        alpha = (float)((double)p[3] / 57.3);
        beta = (float)((double)p[4] / 57.3);
        theta = (float)((double)p[5] / 57.3);
    }
    if (dof_num == 7) a = b = c = p[6];
    if (dof_num == 9) { a = p[6]; b = p[7]; c = p[8]; }
    float ts[12], rot[12], t1[12], t2[12];
    set_identity(ts);
    ts[0] = a; ts[5] = b; ts[10] = c;
    ts[3] = x; ts[7] = y; ts[11] = z;
    set_identity(rot);  // about Z
    rot[0] = std::cos(alpha); rot[1] = std::sin(alpha);
    rot[4] = -std::sin(alpha); rot[5] = std::cos(alpha);
    matrixmultiply(t1, ts, rot);
    set_identity(rot);  // about X
    rot[5] = std::cos(beta); rot[6] = std::sin(beta);
    rot[9] = -std::sin(beta); rot[10] = std::cos(beta);
    matrixmultiply(t2, t1, rot);
    set_identity(rot);  // about Y
    rot[0] = std::cos(theta); rot[2] = -std::sin(theta);
    rot[8] = std::sin(theta); rot[10] = std::cos(theta);
    matrixmultiply(m, t2, rot);
}

void rot2matrix(float *m, float theta, long long sx, long long sy, long long sz, int rot_axis) {
    // shift the rotated box's centre (sNew/2) to the origin, rotate, shift to the volume centre.
    float back[12], rot[12], fwd[12], tmp[12];
    set_identity(back); set_identity(rot); set_identity(fwd);
    const float c = std::cos(theta), s = std::sin(theta);
    long long s_new = 0;
    if (rot_axis == 1) {
        back[7] = (float)(sy / 2); back[11] = (float)(sz / 2);
        rot[5] = c; rot[6] = s; rot[9] = -s; rot[10] = c;
        s_new = (long long)std::round(std::sqrt((double)(sy * sy + sz * sz)));
        fwd[7] = (float)(-s_new / 2); fwd[11] = (float)(-s_new / 2);
    } else if (rot_axis == 2) {
        back[3] = (float)(sx / 2); back[11] = (float)(sz / 2);
        rot[0] = c; rot[2] = -s; rot[8] = s; rot[10] = c;
        s_new = (long long)std::round(std::sqrt((double)(sx * sx + sz * sz)));
        fwd[3] = (float)(-s_new / 2); fwd[11] = (float)(-s_new / 2);
    } else if (rot_axis == 3) {
        back[3] = (float)(sx / 2); back[7] = (float)(sy / 2);
        rot[0] = c; rot[1] = s; rot[4] = -s; rot[5] = c;
        s_new = (long long)std::round(std::sqrt((double)(sx * sx + sy * sy)));
        fwd[3] = (float)(-s_new / 2); fwd[7] = (float)(-s_new / 2);
    }
    matrixmultiply(tmp, back, rot);
    matrixmultiply(m, tmp, fwd);
}

bool check_matrix(const float *m, long long sx, long long sy, long long sz) {
    const float scale_low = 0.5f, scale_up = 1.4f, sum_low = 2.f, sum_up = 4.f, shift_ratio = 0.8f;
    bool ok = true;
    if (m[0] < scale_low || m[0] > scale_up || m[5] < scale_low || m[5] > scale_up || m[10] < scale_low ||
        m[10] > scale_up)
        ok = false;
    const float ssum = m[0] + m[5] + m[10];
    if (ssum < sum_low || ssum > sum_up) ok = false;
    if (std::fabs(m[3]) > shift_ratio * (float)sx || std::fabs(m[7]) > shift_ratio * (float)sy ||
        std::fabs(m[11]) > shift_ratio * (float)sz)
        ok = false;
    return ok;
}

// ---------------------------------------------------------------------------
// 2-D MIP shift search (regChoice 4)
// ---------------------------------------------------------------------------
namespace {

// Mean-subtracted pair and the 2-D ZNCC (-costfunc2D, reference src/api_subfunc.cu:1815-1821 + corrfunc2D :1014-1036).
// A 2-D image is a one-plane volume: with a zero z column the 3-D coordinate expression adds exact zeros, so the
// arithmetic is that of corr2Dkernel.
struct Cost2d {
    Dims d1, d2;
    DeviceBuffer<float> t_ms, s_ms;
    float sd_t = 0.f;
    k::CorrWorkspace corr;
    hipStream_t s;
    Cost2d(const float *d_img1, const float *d_img2, int w1, int h1, int w2, int h2, k::ReduceWorkspace &red, hipStream_t st)
        : d1{w1, h1, 1}, d2{w2, h2, 1}, t_ms((size_t)w1 * h1), s_ms((size_t)w2 * h2), s(st) {
        const size_t n1 = (size_t)w1 * h1, n2 = (size_t)w2 * h2;
        const double sum1 = k::reduce_sync(k::kSum, d_img1, n1, 0.f, red, s);
        k::add_value(t_ms.get(), d_img1, -((float)sum1 / (float)n1), n1, s);
        sd_t = (float)std::sqrt(k::reduce_sync(k::kSumSqShifted, t_ms.get(), n1, 0.f, red, s));
        if (sd_t == 0.f) throw Error(-3, "2-D registration: SD of image 1 is zero (empty image)");
        const double sum2 = k::reduce_sync(k::kSum, d_img2, n2, 0.f, red, s);
        k::add_value(s_ms.get(), d_img2, -((float)sum2 / (float)n2), n2, s);
    }
    static k::Affine12 lift(const float *a) {
        return k::Affine12{{a[0], a[1], 0.f, a[2], a[3], a[4], 0.f, a[5], 0.f, 0.f, 1.f, 0.f}};
    }
    float zncc(const float *a6) {
        double ss = 0, st = 0;
        k::corr_sums_sync(t_ms.get(), s_ms.get(), d1, d2, lift(a6), corr, s, &ss, &st, /*float_products=*/true);
        if (std::sqrt(ss) == 0) return -2.0f;
        return (float)(st / std::sqrt(ss)) / sd_t;
    }
};

// reg2d_shiftalign1 (x_only = false) / reg2d_shiftalignX1 (x_only = true), reference src/api_subfunc.cu:1860-2120:
// exhaustive search of the translation on a (2*total_step)^2 (or 2*total_step) grid.  aff6 = (a0 a1 tx; a3 a4 ty) in/out.
void shift_align_2d(float aff6[6], Cost2d &cost, int w2, int h2, bool x_only, float shift_region, float total_step) {
    float p[6];
    std::memcpy(p, aff6, sizeof p);
    const float off_x = p[2], off_y = p[5];
    const float step_x = (float)w2 * shift_region / total_step, step_y = (float)h2 * shift_region / total_step;
    float best = 0.f, best_x = 0.f, best_y = 0.f;
    for (int i = -(int)total_step; i < (int)total_step; ++i) {
        p[2] = off_x + step_x * (float)i;
        if (x_only) {
            const float c = cost.zncc(p);
            if (c > best) { best = c; best_x = p[2]; }
        } else {
            for (int j = -(int)total_step; j < (int)total_step; ++j) {
                p[5] = off_y + step_y * (float)j;
                const float c = cost.zncc(p);
                if (c > best) { best = c; best_x = p[2]; best_y = p[5]; }
            }
        }
    }
    p[2] = best_x;
    if (!x_only) p[5] = best_y;
    std::memcpy(aff6, p, sizeof p);
}

}  // namespace

void mip_shift_search(float shift[3], const float *d_img1, const float *d_img2, Dims dims, hipStream_t stream) {
    const size_t nxy = (size_t)dims.nx * dims.ny, nzx = (size_t)dims.nz * dims.nx;
    DeviceBuffer<float> p1(nxy > nzx ? nxy : nzx), p2(nxy > nzx ? nxy : nzx);
    k::ReduceWorkspace red;
    k::max_projection(p1.get(), d_img1, dims, 1, stream);  // [ny][nx]
    k::max_projection(p2.get(), d_img2, dims, 1, stream);
    const float shift_region = 0.3f, total_step = 30.f;  // src/api_reg.cpp:463-464
    float t1[6] = {1, 0, 0, 0, 1, 0};
    {
        Cost2d cost(p1.get(), p2.get(), (int)dims.nx, (int)dims.ny, (int)dims.nx, (int)dims.ny, red, stream);
        shift_align_2d(t1, cost, (int)dims.nx, (int)dims.ny, false, shift_region, total_step);
    }
    k::max_projection(p1.get(), d_img1, dims, 2, stream);  // width nz, height nx
    k::max_projection(p2.get(), d_img2, dims, 2, stream);
    float t2[6] = {1, 0, 0, 0, 1, t1[2]};
    {
        Cost2d cost(p1.get(), p2.get(), (int)dims.nz, (int)dims.nx, (int)dims.nz, (int)dims.nx, red, stream);
        shift_align_2d(t2, cost, (int)dims.nz, (int)dims.nx, true, shift_region, total_step);
    }
    shift[0] = t1[2];
    shift[1] = t1[5];
    shift[2] = t2[2];
}

// reg2d, reference src/api_reg.cpp:115-244 with reg2d_affine1 (src/api_subfunc.cu:2233-2336), reg2d_shiftalign1 and
// reg2d_phasor1, on device-resident images.
int reg2d_device(float *d_reg, float *aff6, const float *d_img1, const float *d_img2, int w1, int h1, int w2, int h2,
                 int reg_choice, bool flag_tmx, float ftol, int it_limit, float *records, hipStream_t stream) {
    const Dims d1{w1, h1, 1}, d2{w2, h2, 1};
    k::ReduceWorkspace red;
    float a[6];
    if (flag_tmx) std::memcpy(a, aff6, sizeof a);
    else {  // centred placement (:2279-2280)
        a[0] = 1; a[1] = 0; a[2] = (float)((w2 - w1) / 2);
        a[3] = 0; a[4] = 1; a[5] = (float)((h2 - h1) / 2);
    }
    auto resample = [&](const float *m6) {
        k::affine_resample(d_reg, d_img2, d1, d2, Cost2d::lift(m6), stream, /*strict_window=*/true);
        MIL_HIP(hipStreamSynchronize(stream));
    };
    switch (reg_choice) {
    case 0:  // apply the input matrix (only with an input matrix, :189-194)
        if (flag_tmx) {
            Cost2d cost(d_img1, d_img2, w1, h1, w2, h2, red, stream);
            records[1] = cost.zncc(a);
            records[3] = records[1];
            resample(a);
        }
        return 0;
    case 1: {  // translation grid search, 0.4 of the extent in 40 steps (:196-203)
        Cost2d cost(d_img1, d_img2, w1, h1, w2, h2, red, stream);
        shift_align_2d(a, cost, w2, h2, false, 0.4f, 40.f);
        std::memcpy(aff6, a, sizeof a);
        resample(a);
        return 0;
    }
    case 2: {  // 6-parameter Powell on the 2-D ZNCC (:204-206)
        Cost2d cost(d_img1, d_img2, w1, h1, w2, h2, red, stream);
        float last[6], xi[36];
        std::memcpy(last, a, sizeof a);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) xi[i * 6 + j] = (i == j) ? 1.f : 0.f;
        int evals = 0;
        CostFn f = [&](const float *x) -> float {  // costfunc2D: remembers the matrix of the LAST evaluation
            std::memcpy(last, x, sizeof last);
            ++evals;
            return -cost.zncc(x);
        };
        records[1] = -f(a);
        evals = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const PowellResult r = powell_minimize(a, xi, 6, 6, ftol, f, &evals, it_limit);
        std::memcpy(aff6, last, sizeof last);
        records[3] = -r.fret;
        records[5] = (float)evals;
        records[6] = (float)seconds_since(t0);
        resample(last);
        return 0;
    }
    case 3: {  // pixel-level phasor shift (:207-220)
        if (w1 != w2 || h1 != h2) throw Error(1, "reg2d: image sizes differ, phasor registration needs equal sizes");
        long long sh[3];
        phasor_shift(sh, d_img1, d_img2, d1, stream);
        k::shift_volume(d_reg, d_img2, d1, -sh[0], -sh[1], 0, false, stream);
        MIL_HIP(hipStreamSynchronize(stream));
        aff6[0] = 1; aff6[1] = 0; aff6[2] = (float)sh[0];
        aff6[3] = 0; aff6[4] = 1; aff6[5] = (float)sh[1];
        return 0;
    }
    default: throw Error(1, "reg2d: regChoice must be 0..3");
    }
}

// ---------------------------------------------------------------------------
// 3-D phasor shift (regChoice 1 / 3)
// ---------------------------------------------------------------------------
namespace {

// zncc1, reference src/api_subfunc.cu:2409-2432
float zncc_volumes(const float *a, const float *b, size_t n, k::ReduceWorkspace &red, hipStream_t s) {
    const double sa = k::reduce_sync(k::kSum, a, n, 0.f, red, s), sb = k::reduce_sync(k::kSum, b, n, 0.f, red, s);
    const float ca = -(float)sa / (float)n, cb = -(float)sb / (float)n;
    const double st = k::reduce_dot_sync(a, b, n, ca, cb, red, s);
    const double tt = k::reduce_dot_sync(a, a, n, ca, ca, red, s);
    const double ss = k::reduce_dot_sync(b, b, n, cb, cb, red, s);
    const float den = (float)std::sqrt(tt * ss);
    return den != 0.f ? (float)(st / den) : -2.0f;
}

}  // namespace

void phasor_shift(long long shift[3], const float *d_img1, const float *d_img2, Dims dims, hipStream_t stream) {
    const long long sx = dims.nx, sy = dims.ny, sz = dims.nz;
    const size_t n = (size_t)dims.count(), ns = (size_t)(sx / 2 + 1) * sy * sz;
    k::ReduceWorkspace red;
    {
        Fft3D fft(dims, stream);
        DeviceBuffer<float2> spec1(ns), spec2(ns);
        DeviceBuffer<float> work(n), phasor(n), shifted(n);
        // rocFFT real transforms may overwrite their input: transform copies (:2475-2484)
        MIL_HIP(hipMemcpyAsync(work.get(), d_img1, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
        fft.r2c(work.get(), spec1.get());
        MIL_HIP(hipMemcpyAsync(work.get(), d_img2, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
        fft.r2c(work.get(), spec2.get());
        k::cross_power_norm(spec2.get(), spec1.get(), spec2.get(), ns, stream);
        fft.c2r(spec2.get(), phasor.get());
        k::shift_volume(shifted.get(), phasor.get(), dims, sx / 2, sy / 2, sz / 2, true, stream);  // :2487
        // max3Dgpu :437-471: column maxima on the device, columns scanned x-outer / y-inner on the host
        DeviceBuffer<float> col_max((size_t)sx * sy);
        DeviceBuffer<int> col_arg((size_t)sx * sy);
        k::argmax_columns(col_max.get(), col_arg.get(), shifted.get(), dims, stream);
        std::vector<float> h_max((size_t)sx * sy);
        std::vector<int> h_arg((size_t)sx * sy);
        MIL_HIP(hipMemcpyAsync(h_max.data(), col_max.get(), h_max.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
        MIL_HIP(hipMemcpyAsync(h_arg.data(), col_arg.get(), h_arg.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
        MIL_HIP(hipStreamSynchronize(stream));
        float peak = h_max[0];
        long long px = 0, py = 0, pz = 0;  // like the reference: (0,0,0) with the value of column 0 until something is larger
        for (long long i = 0; i < sx; ++i)
            for (long long j = 0; j < sy; ++j) {
                const float t = h_max[(size_t)(i + j * sx)];
                if (peak < t) {
                    peak = t;
                    px = i;
                    py = j;
                    pz = h_arg[(size_t)(i + j * sx)];
                }
            }
        shift[0] = px - sx / 2;
        shift[1] = py - sy / 2;
        shift[2] = pz - sz / 2;
    }
    // wrap-around disambiguation, only for shifts beyond a quarter of the extent (:2497-2588)
    const long long beta = 4, d[3] = {sx, sy, sz};
    long long ab[3];
    for (int a = 0; a < 3; ++a) ab[a] = shift[a] < 0 ? -shift[a] : shift[a];
    if (!(ab[0] > sx / beta || ab[1] > sy / beta || ab[2] > sz / beta)) return;
    DeviceBuffer<float> moved(n), crop1(n), crop2(n);
    k::shift_volume(moved.get(), d_img2, dims, -shift[0], -shift[1], -shift[2], true, stream);
    long long size[3][2], org[3][2];
    for (int a = 0; a < 3; ++a) {
        size[a][0] = d[a] - ab[a];
        size[a][1] = ab[a];
        if (shift[a] > 0) { org[a][0] = 0; org[a][1] = d[a] - ab[a]; }
        else { org[a][0] = ab[a]; org[a][1] = 0; }
    }
    float best = -3.f;
    int ind[3] = {0, 0, 0};
    for (int i = 0; i < 2; ++i) {
        if (!(size[0][i] > sx / beta)) continue;
        for (int j = 0; j < 2; ++j) {
            if (!(size[1][j] > sy / beta)) continue;
            for (int kk = 0; kk < 2; ++kk) {
                if (!(size[2][kk] > sz / beta)) continue;
                const Dims c{size[0][i], size[1][j], size[2][kk]};
                k::crop_offset(crop1.get(), d_img1, c, dims, org[0][i], org[1][j], org[2][kk], stream);
                k::crop_offset(crop2.get(), moved.get(), c, dims, org[0][i], org[1][j], org[2][kk], stream);
                const float cc = zncc_volumes(crop1.get(), crop2.get(), (size_t)c.count(), red, stream);
                if (best < cc) {
                    best = cc;
                    ind[0] = i; ind[1] = j; ind[2] = kk;
                }
            }
        }
    }
    for (int a = 0; a < 3; ++a)
        if (ind[a] == 1) shift[a] = shift[a] > 0 ? shift[a] - d[a] : shift[a] + d[a];
}

// ---------------------------------------------------------------------------
// driver
// ---------------------------------------------------------------------------

int reg3d_affine_device(float *d_reg, float *iTmx, const float *d_img1, const float *d_img2, Dims dims,
                        int aff_method, bool flag_tmx, float ftol, int it_limit, bool verbose, float *records,
                        hipStream_t stream, const RegOptions &opt) {
    TraceRange trace("mil::reg3d_affine");
    const auto t_start = std::chrono::steady_clock::now();
    const size_t n = (size_t)dims.count();
    k::Affine12 aff;  // the reference's global affCoef: matrix of the most recent cost evaluation

    if (aff_method == 0) {  // no registration: apply the input matrix or copy (reference :2767-2781)
        if (flag_tmx) {
            std::memcpy(aff.m, iTmx, sizeof aff.m);
            k::affine_resample(d_reg, d_img2, dims, dims, aff, stream);
        } else {
            MIL_HIP(hipMemcpyAsync(d_reg, d_img2, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
            set_identity(iTmx);
        }
        MIL_HIP(hipStreamSynchronize(stream));
        records[7] = (float)seconds_since(t_start);
        if (verbose) printf("\t... no registration performed!\n");
        return 0;
    }
    if (aff_method < 0 || aff_method > 7) throw Error(1, "reg3d_affine: affMethod must be 0..7");

    float aff_initial[12];
    set_identity(aff_initial);
    float p12[12], xi12[12 * 12], p9[9] = {0, 0, 0, 0, 0, 0, 1, 1, 1}, xi9[9 * 9];
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) xi12[i * 12 + j] = (i == j) ? 1.f : 0.f;
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) xi9[i * 9 + j] = (i == j) ? 1.f : 0.f;  // NOT reset between cascade stages

    DeviceBuffer<float> src_ms(n), tgt_ms(n);
    k::ReduceWorkspace red;
    k::CorrWorkspace corr;

    // source: optional pre-transform by the input matrix, then mean subtraction (:2817-2845)
    const float *src_for_stats = d_img2;
    if (flag_tmx) {
        if (aff_method == 5) {
            std::memcpy(aff_initial, iTmx, sizeof aff_initial);
        } else {
            std::memcpy(aff.m, iTmx, sizeof aff.m);
            k::affine_resample(d_reg, d_img2, dims, dims, aff, stream);
            src_for_stats = d_reg;
        }
    }
    const double sum2 = k::reduce_sync(k::kSum, src_for_stats, n, 0.f, red, stream);
    k::add_value(src_ms.get(), src_for_stats, -(float)sum2 / (float)n, n, stream);
    const double sqr2 = k::reduce_sync(k::kSumSqShifted, src_ms.get(), n, 0.f, red, stream);
    if ((float)std::sqrt(sqr2) == 0.f)
        throw Error(-3, "reg3d_affine: SD of image 2 is zero (empty image, or empty after the initial transform)");
    // target (:2859-2868)
    const double sum1 = k::reduce_sync(k::kSum, d_img1, n, 0.f, red, stream);
    const float mean1_neg = -(float)sum1 / (float)n;
    k::add_value(tgt_ms.get(), d_img1, mean1_neg, n, stream);
    const double sqr1 = k::reduce_sync(k::kSumSqShifted, tgt_ms.get(), n, 0.f, red, stream);
    const float sd_t = (float)std::sqrt(sqr1);
    if (sd_t == 0.f) throw Error(-3, "reg3d_affine: SD of image 1 is zero (empty image)");

    // The cost kernel reads the source through a copy with a replicated one-voxel border (corr_padded.hip); volumes
    // beyond its 32-bit offset range keep the general kernel.  Same sums either way.
    k::PaddedSource src_pad;
    const bool padded = k::PaddedSource::eligible(dims, dims) && k::texture_weight_bits() == 0 &&
                        !(getenv("MIL_CORR_UNPADDED") && getenv("MIL_CORR_UNPADDED")[0] == '1');
    if (padded) {
        src_pad.build(src_ms.get(), dims, stream);
        src_ms.release_after(stream);  // only this stream ever touched it
    }

    int evals = 0, launches = 0;
    bool dof9 = false;
    int dof_num = 12;
    // Powell re-evaluates points it has just visited (every line minimisation starts by evaluating its own starting
    // point, src/api_powell.c:291-293 / mnbrak).  The cost is a deterministic function of the 12 matrix floats, so a
    // repeat is answered from the last few results instead of another pass over both volumes: same value, same
    // evaluation count, same trajectory.
    struct Memo {
        float m[12];
        float value;
    };
    constexpr int kMemo = 32;  // 4 entries catch the start-of-line repeats (9 % of the evaluations), 32 another 2.5 %
    Memo memo[kMemo];
    int memo_n = 0, memo_next = 0;
    const bool memo_off = getenv("MIL_REG_NOMEMO") && getenv("MIL_REG_NOMEMO")[0] == '1';  // per call: the tests flip it
    CostFn cost = [&](const float *x) -> float {  // costfunc :2377-2388 + corrfunc :954-988
        if (dof9) dof9tomatrix(aff.m, x, dof_num);
        else p2matrix(aff.m, x);
        ++evals;
        if (!memo_off)
            for (int i = 0; i < memo_n; ++i)
                if (std::memcmp(memo[i].m, aff.m, sizeof aff.m) == 0) {
                    trace_evaluation(aff.m, memo[i].value);
                    return memo[i].value;
                }
        ++launches;
        double ss = 0, st = 0;
        if (padded) k::corr_sums_padded_sync(tgt_ms.get(), src_pad, dims, aff, corr, stream, &ss, &st);
        else k::corr_sums_sync(tgt_ms.get(), src_ms.get(), dims, dims, aff, corr, stream, &ss, &st);
        const double root = std::sqrt(ss);
        const float zncc = (root == 0) ? -2.0f : (float)(st / root) / sd_t;
        std::memcpy(memo[memo_next].m, aff.m, sizeof aff.m);
        memo[memo_next].value = -zncc;
        memo_next = (memo_next + 1) % kMemo;
        if (memo_n < kMemo) ++memo_n;
        trace_evaluation(aff.m, -zncc);
        return -zncc;
    };

    matrix2p(p12, aff_initial);
    const auto t_opt = std::chrono::steady_clock::now();
    records[1] = -cost(p12);  // initial ZNCC (:2881)
    records[4] = 0.f;
    if (verbose) printf("\t... initial cross correlation value: %f;\n", records[1]);
    evals = 0;
    launches = 0;

    PowellResult r;
    auto stage9 = [&](int n_dof, float tol) {
        dof9 = true;
        dof_num = n_dof;
        TraceRange stage("mil::powell (<= 9 DOF stage)");
        r = powell_minimize(p9, xi9, 9, n_dof, tol, cost, &evals, it_limit);
        if (opt.matrix_from_optimum) dof9tomatrix(aff.m, p9, n_dof);
    };
    auto stage12 = [&](float tol) {
        dof9 = false;
        dof_num = 12;
        TraceRange stage("mil::powell (12 DOF stage)");
        r = powell_minimize(p12, xi12, 12, 12, tol, cost, &evals, it_limit);
        if (opt.matrix_from_optimum) p2matrix(aff.m, p12);
    };
    switch (aff_method) {
    case 1: stage9(3, ftol); break;
    case 2: stage9(6, ftol); break;
    case 3: stage9(7, ftol); break;
    case 4: stage9(9, ftol); break;
    case 5: stage12(ftol); break;
    case 6:  // rigid body first, then 12 DOF (:2915-2929)
        stage9(6, (float)0.01);
        records[2] = -r.fret;
        if (verbose) printf("\t... cross correlation value after 6 DOF: %f;\n", -r.fret);
        matrix2p(p12, aff.m);
        stage12(ftol);
        break;
    case 7:  // 3 -> 6 -> 9 -> 12 DOF (:2930-2954)
        stage9(3, (float)0.01);
        if (verbose) printf("\t... cross correlation value after 3 DOF: %f;\n", -r.fret);
        stage9(6, (float)0.01);
        if (verbose) printf("\t... cross correlation value after 6 DOF: %f;\n", -r.fret);
        stage9(9, (float)0.005);
        records[2] = -r.fret;
        if (verbose) printf("\t... cross correlation value after 9 DOF: %f;\n", -r.fret);
        matrix2p(p12, aff.m);
        stage12(ftol);
        break;
    }
    if (flag_tmx && aff_method != 5) {  // compose with the input matrix (:2958-2961)
        float composed[12];
        matrixmultiply(composed, iTmx, aff.m);
        std::memcpy(aff.m, composed, sizeof composed);
    }
    std::memcpy(iTmx, aff.m, sizeof aff.m);
    records[3] = -r.fret;
    records[5] = (float)evals;
    records[6] = (float)seconds_since(t_opt);
    g_last_cost_launches = launches;
    if (verbose) {
        printf("\t... optimized cross correlation value: %f;\n", records[3]);
        printf("\t... total sub iteration number: %d;\n", evals);
        if (launches != evals) printf("\t... (%d of them answered from the last results)\n", evals - launches);
        printf("\t... time cost for all iterations: %f s;\n", records[6]);
    }
    // final resample of the RAW source (:2974-2977)
    k::affine_resample(d_reg, d_img2, dims, dims, aff, stream);
    MIL_HIP(hipStreamSynchronize(stream));
    records[7] = (float)seconds_since(t_start);
    if (verbose) printf("\t... time cost for registration: %f s;\n", records[7]);
    return 0;
}

}  // namespace mil
