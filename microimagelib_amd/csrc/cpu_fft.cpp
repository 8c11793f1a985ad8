// Host 3-D real transforms of the gpuMemMode 0 path -- see cpu_fft.h.
//
// Algorithm: Stockham autosort (decimation in frequency, natural order in and out), radices 8 / 4 / 2 and generic odd
// primes, on a tile of kW neighbouring columns held as separate re / im planes `[n][kW]` so that every inner loop is
// a unit-stride loop over s * kW floats that the compiler vectorises (AVX-512 / AVX2 clones picked at load time).
// Real rows use the packed half-length complex transform + Hermitian untangle.
#include "cpu_fft.h"

#include <cmath>
#include <cstring>
#include <thread>

namespace mil {
namespace cpu {

namespace {

constexpr int kW = 16;  // columns per tile: 16 complex = one 128-byte line pair of the interleaved spectrum

// feature-based clones: the resolver tests CPU features (Intel and AMD alike), not a vendor model name
#define MIL_CPU_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

struct Stage {
    int r, m, s;                 // radix, n_cur / r, stride (product of the radices before)
    std::vector<float> twr, twi;  // w_{n_cur}^{j p}, index (j - 1) * m + p, j = 1 .. r-1
    std::vector<float> rr, ri;    // r-th roots of unity, e^{-2 pi i k / r}
};

MIL_CPU_CLONES
void pass_r2(const float *__restrict xr, const float *__restrict xi, float *__restrict yr, float *__restrict yi,
             int m, int sw, const float *twr, const float *twi, bool inv) {
    for (int p = 0; p < m; ++p) {
        const float wr = twr[p], wi = inv ? -twi[p] : twi[p];
        const float *ar = xr + (size_t)p * sw, *ai = xi + (size_t)p * sw;
        const float *br = xr + (size_t)(p + m) * sw, *bi = xi + (size_t)(p + m) * sw;
        float *o0r = yr + (size_t)(2 * p) * sw, *o0i = yi + (size_t)(2 * p) * sw;
        float *o1r = o0r + sw, *o1i = o0i + sw;
#pragma GCC ivdep
        for (int i = 0; i < sw; ++i) {
            const float dr = ar[i] - br[i], di = ai[i] - bi[i];
            o0r[i] = ar[i] + br[i];
            o0i[i] = ai[i] + bi[i];
            o1r[i] = dr * wr - di * wi;
            o1i[i] = dr * wi + di * wr;
        }
    }
}

MIL_CPU_CLONES
void pass_r4(const float *__restrict xr, const float *__restrict xi, float *__restrict yr, float *__restrict yi,
             int m, int sw, const float *twr, const float *twi, bool inv) {
    const float sg = inv ? -1.f : 1.f;  // forward: (a1 - a3) * (-i); inverse: * (+i)
    for (int p = 0; p < m; ++p) {
        const float w1r = twr[p], w1i = sg * twi[p];
        const float w2r = twr[m + p], w2i = sg * twi[m + p];
        const float w3r = twr[2 * m + p], w3i = sg * twi[2 * m + p];
        const float *a0r = xr + (size_t)p * sw, *a0i = xi + (size_t)p * sw;
        const float *a1r = a0r + (size_t)m * sw, *a1i = a0i + (size_t)m * sw;
        const float *a2r = a1r + (size_t)m * sw, *a2i = a1i + (size_t)m * sw;
        const float *a3r = a2r + (size_t)m * sw, *a3i = a2i + (size_t)m * sw;
        float *o0r = yr + (size_t)(4 * p) * sw, *o0i = yi + (size_t)(4 * p) * sw;
        float *o1r = o0r + sw, *o1i = o0i + sw;
        float *o2r = o1r + sw, *o2i = o1i + sw;
        float *o3r = o2r + sw, *o3i = o2i + sw;
#pragma GCC ivdep
        for (int i = 0; i < sw; ++i) {
            const float t0r = a0r[i] + a2r[i], t0i = a0i[i] + a2i[i];
            const float t1r = a0r[i] - a2r[i], t1i = a0i[i] - a2i[i];
            const float t2r = a1r[i] + a3r[i], t2i = a1i[i] + a3i[i];
            const float dr = a1r[i] - a3r[i], di = a1i[i] - a3i[i];
            const float t3r = sg * di, t3i = -sg * dr;
            o0r[i] = t0r + t2r;
            o0i[i] = t0i + t2i;
            const float b1r = t1r + t3r, b1i = t1i + t3i;
            const float b2r = t0r - t2r, b2i = t0i - t2i;
            const float b3r = t1r - t3r, b3i = t1i - t3i;
            o1r[i] = b1r * w1r - b1i * w1i;
            o1i[i] = b1r * w1i + b1i * w1r;
            o2r[i] = b2r * w2r - b2i * w2i;
            o2i[i] = b2r * w2i + b2i * w2r;
            o3r[i] = b3r * w3r - b3i * w3i;
            o3i[i] = b3r * w3i + b3i * w3r;
        }
    }
}

// radix 8 = (a_k +- a_{k+4}) then two radix-4 butterflies; w8 = e^{-+ i pi / 4}
MIL_CPU_CLONES
void pass_r8(const float *__restrict xr, const float *__restrict xi, float *__restrict yr, float *__restrict yi,
             int m, int sw, const float *twr, const float *twi, bool inv) {
    const float sg = inv ? -1.f : 1.f;
    const float h = 0.70710678118654752440f;
    for (int p = 0; p < m; ++p) {
        float wr[8], wi[8];
        wr[0] = 1.f;
        wi[0] = 0.f;
        for (int j = 1; j < 8; ++j) {
            wr[j] = twr[(size_t)(j - 1) * m + p];
            wi[j] = sg * twi[(size_t)(j - 1) * m + p];
        }
        const float *ar[8], *ai[8];
        float *orr[8], *oi[8];
        for (int k = 0; k < 8; ++k) {
            ar[k] = xr + (size_t)(p + k * m) * sw;
            ai[k] = xi + (size_t)(p + k * m) * sw;
            orr[k] = yr + (size_t)(8 * p + k) * sw;
            oi[k] = yi + (size_t)(8 * p + k) * sw;
        }
#pragma GCC ivdep
        for (int i = 0; i < sw; ++i) {
            // first split
            const float b0r = ar[0][i] + ar[4][i], b0i = ai[0][i] + ai[4][i];
            const float b1r = ar[1][i] + ar[5][i], b1i = ai[1][i] + ai[5][i];
            const float b2r = ar[2][i] + ar[6][i], b2i = ai[2][i] + ai[6][i];
            const float b3r = ar[3][i] + ar[7][i], b3i = ai[3][i] + ai[7][i];
            const float c0r = ar[0][i] - ar[4][i], c0i = ai[0][i] - ai[4][i];
            const float d1r = ar[1][i] - ar[5][i], d1i = ai[1][i] - ai[5][i];
            const float d2r = ar[2][i] - ar[6][i], d2i = ai[2][i] - ai[6][i];
            const float d3r = ar[3][i] - ar[7][i], d3i = ai[3][i] - ai[7][i];
            // c_k = d_k * w8^k;  forward w8 = (1 - i) / sqrt 2, w8^2 = -i, w8^3 = (-1 - i) / sqrt 2
            const float c1r = h * (d1r + sg * d1i), c1i = h * (d1i - sg * d1r);
            const float c2r = sg * d2i, c2i = -sg * d2r;
            const float c3r = h * (-d3r + sg * d3i), c3i = h * (-d3i - sg * d3r);
            // radix 4 on b -> outputs 0, 2, 4, 6
            {
                const float e0r = b0r + b2r, e0i = b0i + b2i, e1r = b0r - b2r, e1i = b0i - b2i;
                const float e2r = b1r + b3r, e2i = b1i + b3i;
                const float fr = b1r - b3r, fi = b1i - b3i;
                const float e3r = sg * fi, e3i = -sg * fr;
                orr[0][i] = e0r + e2r;
                oi[0][i] = e0i + e2i;
                const float v2r = e1r + e3r, v2i = e1i + e3i;
                const float v4r = e0r - e2r, v4i = e0i - e2i;
                const float v6r = e1r - e3r, v6i = e1i - e3i;
                orr[2][i] = v2r * wr[2] - v2i * wi[2];
                oi[2][i] = v2r * wi[2] + v2i * wr[2];
                orr[4][i] = v4r * wr[4] - v4i * wi[4];
                oi[4][i] = v4r * wi[4] + v4i * wr[4];
                orr[6][i] = v6r * wr[6] - v6i * wi[6];
                oi[6][i] = v6r * wi[6] + v6i * wr[6];
            }
            // radix 4 on c -> outputs 1, 3, 5, 7
            {
                const float e0r = c0r + c2r, e0i = c0i + c2i, e1r = c0r - c2r, e1i = c0i - c2i;
                const float e2r = c1r + c3r, e2i = c1i + c3i;
                const float fr = c1r - c3r, fi = c1i - c3i;
                const float e3r = sg * fi, e3i = -sg * fr;
                const float v1r = e0r + e2r, v1i = e0i + e2i;
                const float v3r = e1r + e3r, v3i = e1i + e3i;
                const float v5r = e0r - e2r, v5i = e0i - e2i;
                const float v7r = e1r - e3r, v7i = e1i - e3i;
                orr[1][i] = v1r * wr[1] - v1i * wi[1];
                oi[1][i] = v1r * wi[1] + v1i * wr[1];
                orr[3][i] = v3r * wr[3] - v3i * wi[3];
                oi[3][i] = v3r * wi[3] + v3i * wr[3];
                orr[5][i] = v5r * wr[5] - v5i * wi[5];
                oi[5][i] = v5r * wi[5] + v5i * wr[5];
                orr[7][i] = v7r * wr[7] - v7i * wi[7];
                oi[7][i] = v7r * wi[7] + v7i * wr[7];
            }
        }
    }
}

// generic radix (odd primes 3, 5, 7, 11, 13, ...): O(r^2) per butterfly
MIL_CPU_CLONES
void pass_generic(const float *__restrict xr, const float *__restrict xi, float *__restrict yr,
                  float *__restrict yi, int r, int m, int sw, const float *twr, const float *twi, const float *rr,
                  const float *ri, bool inv) {
    const float sg = inv ? -1.f : 1.f;
    for (int p = 0; p < m; ++p) {
        for (int j = 0; j < r; ++j) {
            float *outr = yr + (size_t)(r * p + j) * sw, *outi = yi + (size_t)(r * p + j) * sw;
            const float *a0r = xr + (size_t)p * sw, *a0i = xi + (size_t)p * sw;
    #pragma GCC ivdep
        for (int i = 0; i < sw; ++i) {
                outr[i] = a0r[i];
                outi[i] = a0i[i];
            }
            for (int k = 1; k < r; ++k) {
                const int e = (j * k) % r;
                const float cr = rr[e], ci = sg * ri[e];
                const float *akr = xr + (size_t)(p + k * m) * sw, *aki = xi + (size_t)(p + k * m) * sw;
        #pragma GCC ivdep
        for (int i = 0; i < sw; ++i) {
                    outr[i] += akr[i] * cr - aki[i] * ci;
                    outi[i] += akr[i] * ci + aki[i] * cr;
                }
            }
            if (j > 0) {
                const float wr = twr[(size_t)(j - 1) * m + p], wi = sg * twi[(size_t)(j - 1) * m + p];
        #pragma GCC ivdep
        for (int i = 0; i < sw; ++i) {
                    const float vr = outr[i], vi = outi[i];
                    outr[i] = vr * wr - vi * wi;
                    outi[i] = vr * wi + vi * wr;
                }
            }
        }
    }
}

// Hermitian untangle of the packed real transform, all kW lanes of a tile at once.  zr / zi: [h][kW]; out: [h+1][kW].
MIL_CPU_CLONES
void untangle_forward(const float *__restrict zr, const float *__restrict zi, float *__restrict ur,
                      float *__restrict ui, int h, const float *c, const float *s) {
    for (int k = 0; k <= h; ++k) {
        const int ka = k == h ? 0 : k, kb = k == 0 ? 0 : h - k;
        const float *ar = zr + (size_t)ka * kW, *ai = zi + (size_t)ka * kW;
        const float *br = zr + (size_t)kb * kW, *bi = zi + (size_t)kb * kW;
        float *outr = ur + (size_t)k * kW, *outi = ui + (size_t)k * kW;
        const float ck = c[k], sk = s[k];
#pragma GCC ivdep
        for (int l = 0; l < kW; ++l) {
            const float er = 0.5f * (ar[l] + br[l]), ei = 0.5f * (ai[l] - bi[l]);  // even part (conj Z[h-k])
            const float dr = 0.5f * (ar[l] - br[l]), di = 0.5f * (ai[l] + bi[l]);  // (Z - conj Z') / 2
            const float orr = di, oi = -dr;                                        // / i
            outr[l] = er + (orr * ck - oi * sk);
            outi[l] = ei + (orr * sk + oi * ck);
        }
    }
}

// xr / xi: [h+1][kW] spectrum rows (transposed); out zr / zi: [h][kW]
MIL_CPU_CLONES
void untangle_inverse(const float *__restrict xr, const float *__restrict xi, float *__restrict zr,
                      float *__restrict zi, int h, const float *c, const float *s) {
    for (int k = 0; k < h; ++k) {
        const float *ar = xr + (size_t)k * kW, *ai = xi + (size_t)k * kW;
        const float *br = xr + (size_t)(h - k) * kW, *bi = xi + (size_t)(h - k) * kW;
        float *outr = zr + (size_t)k * kW, *outi = zi + (size_t)k * kW;
        const float ck = c[k], sk = -s[k];  // e^{+2 pi i k / nx}
#pragma GCC ivdep
        for (int l = 0; l < kW; ++l) {
            const float er = ar[l] + br[l], ei = ai[l] - bi[l];
            const float dr = ar[l] - br[l], di = ai[l] + bi[l];
            const float pr = dr * ck - di * sk, pi = dr * sk + di * ck;
            outr[l] = er - pi;  // + i * (pr + i pi)
            outi[l] = ei + pr;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------ 1-D plan
class Fft1D {
public:
    explicit Fft1D(int n) : n_(n) {
        if (n < 1) throw Error(-1, "Fft1D: bad length");
        int rest = n, s = 1;
        while (rest > 1) {
            int r;
            if (rest % 8 == 0 && rest != 16) r = 8;
            else if (rest % 4 == 0) r = 4;
            else if (rest % 2 == 0) r = 2;
            else {
                r = 3;
                while (rest % r) r += 2;
            }
            Stage st;
            st.r = r;
            st.m = rest / r;
            st.s = s;
            st.twr.resize((size_t)(r - 1) * st.m);
            st.twi.resize((size_t)(r - 1) * st.m);
            for (int j = 1; j < r; ++j)
                for (int p = 0; p < st.m; ++p) {
                    const double ang = -2.0 * M_PI * (double)j * (double)p / (double)rest;
                    st.twr[(size_t)(j - 1) * st.m + p] = (float)std::cos(ang);
                    st.twi[(size_t)(j - 1) * st.m + p] = (float)std::sin(ang);
                }
            st.rr.resize((size_t)r);
            st.ri.resize((size_t)r);
            for (int k = 0; k < r; ++k) {
                const double ang = -2.0 * M_PI * (double)k / (double)r;
                st.rr[(size_t)k] = (float)std::cos(ang);
                st.ri[(size_t)k] = (float)std::sin(ang);
            }
            stages_.push_back(std::move(st));
            rest /= r;
            s *= r;
        }
    }
    int n() const { return n_; }
    // Transforms the tile (re, im: [n][kW]); `sre, sim` is scratch of the same size.  Returns true when the result
    // lies in the scratch pair.
    bool run(float *re, float *im, float *sre, float *sim, bool inverse) const {
        float *xr = re, *xi = im, *yr = sre, *yi = sim;
        bool in_scratch = false;
        for (const Stage &st : stages_) {
            const int sw = st.s * kW;
            if (st.r == 8) pass_r8(xr, xi, yr, yi, st.m, sw, st.twr.data(), st.twi.data(), inverse);
            else if (st.r == 4) pass_r4(xr, xi, yr, yi, st.m, sw, st.twr.data(), st.twi.data(), inverse);
            else if (st.r == 2) pass_r2(xr, xi, yr, yi, st.m, sw, st.twr.data(), st.twi.data(), inverse);
            else pass_generic(xr, xi, yr, yi, st.r, st.m, sw, st.twr.data(), st.twi.data(), st.rr.data(),
                              st.ri.data(), inverse);
            std::swap(xr, yr);
            std::swap(xi, yi);
            in_scratch = !in_scratch;
        }
        return in_scratch;
    }

private:
    int n_;
    std::vector<Stage> stages_;
};

// ------------------------------------------------------------------------------------------ threads
void parallel_for(long long n, int threads, const std::function<void(long long, long long, int)> &fn) {
    if (n <= 0) return;
    if (threads > n) threads = (int)n;
    if (threads <= 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errors((size_t)threads);
    pool.reserve((size_t)threads - 1);
    auto work = [&](int t) {
        const long long b = n * t / threads, e = n * (t + 1) / threads;
        try {
            fn(b, e, t);
        } catch (...) {
            errors[(size_t)t] = std::current_exception();
        }
    };
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    for (auto &e : errors)
        if (e) std::rethrow_exception(e);
}

int default_threads() {
    const char *v = getenv("MIL_CPU_THREADS");
    if (!v || !*v) return 1;
    const int t = atoi(v);
    if (t > 0) return t;
    const unsigned hw = std::thread::hardware_concurrency();
    return hw ? (int)hw : 1;
}

// ------------------------------------------------------------------------------------------ 3-D plan
namespace {
struct Tile {
    std::vector<float> buf;
    float *re, *im, *sre, *sim;
    explicit Tile(int n) : buf((size_t)4 * n * kW + 64) {
        // 64-byte aligned planes
        float *base = buf.data();
        while (((uintptr_t)base) & 63) ++base;
        re = base;
        im = re + (size_t)n * kW;
        sre = im + (size_t)n * kW;
        sim = sre + (size_t)n * kW;
    }
};
}  // namespace

Rfft3D::Rfft3D(Dims box, int threads) : box_(box), threads_(threads < 1 ? 1 : threads) {
    if (box.nx < 2 || box.nx % 2 || box.ny < 1 || box.nz < 1) throw Error(-1, "Rfft3D: nx must be even");
    fx_.reset(new Fft1D((int)box.nx / 2));
    fy_.reset(new Fft1D((int)box.ny));
    fz_.reset(new Fft1D((int)box.nz));
    const int h = (int)box.nx / 2;
    unt_c_.resize((size_t)h + 1);
    unt_s_.resize((size_t)h + 1);
    for (int k = 0; k <= h; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)box.nx;
        unt_c_[(size_t)k] = (float)std::cos(ang);
        unt_s_[(size_t)k] = (float)std::sin(ang);
    }
}

Rfft3D::~Rfft3D() = default;

void Rfft3D::columns(float2 *spec, bool along_z, bool inverse) const {
    const long long nxh1 = box_.nx / 2 + 1, ny = box_.ny, nz = box_.nz;
    const int n = (int)(along_z ? nz : ny);
    if (n == 1) return;
    const long long stride = along_z ? ny * nxh1 : nxh1;       // between consecutive elements of a column
    const long long outer = along_z ? ny : nz;                 // columns sets
    const long long outer_stride = along_z ? nxh1 : ny * nxh1;
    const long long xtiles = (nxh1 + kW - 1) / kW;
    const Fft1D &plan = along_z ? *fz_ : *fy_;
    parallel_for(outer * xtiles, threads_, [&](long long b, long long e, int) {
        Tile t(n);
        for (long long it = b; it < e; ++it) {
            const long long o = it / xtiles, x0 = (it % xtiles) * kW;
            const int w = (int)std::min<long long>(kW, nxh1 - x0);
            float2 *base = spec + o * outer_stride + x0;
            for (int j = 0; j < n; ++j) {
                const float2 *row = base + (long long)j * stride;
                float *r = t.re + (size_t)j * kW, *i = t.im + (size_t)j * kW;
                for (int l = 0; l < w; ++l) {
                    r[l] = row[l].x;
                    i[l] = row[l].y;
                }
                for (int l = w; l < kW; ++l) r[l] = i[l] = 0.f;
            }
            const bool sc = plan.run(t.re, t.im, t.sre, t.sim, inverse);
            const float *rr = sc ? t.sre : t.re, *ri = sc ? t.sim : t.im;
            for (int j = 0; j < n; ++j) {
                float2 *row = base + (long long)j * stride;
                const float *r = rr + (size_t)j * kW, *i = ri + (size_t)j * kW;
                for (int l = 0; l < w; ++l) row[l] = make_float2(r[l], i[l]);
            }
        }
    });
}

void Rfft3D::forward(const float *real, float2 *spec) const {
    const long long nx = box_.nx, rows = box_.ny * box_.nz, nxh1 = nx / 2 + 1;
    const int h = (int)nx / 2;
    const long long tiles = (rows + kW - 1) / kW;
    const float *uc = unt_c_.data(), *us = unt_s_.data();
    parallel_for(tiles, threads_, [&](long long b, long long e, int) {
        Tile t(h + 1);
        for (long long it = b; it < e; ++it) {
            const long long r0 = it * kW;
            const int w = (int)std::min<long long>(kW, rows - r0);
            for (int l = 0; l < kW; ++l) {
                if (l < w) {
                    const float *row = real + (r0 + l) * nx;
                    for (int j = 0; j < h; ++j) {
                        t.re[(size_t)j * kW + l] = row[2 * j];
                        t.im[(size_t)j * kW + l] = row[2 * j + 1];
                    }
                } else {
                    for (int j = 0; j < h; ++j) t.re[(size_t)j * kW + l] = t.im[(size_t)j * kW + l] = 0.f;
                }
            }
            const bool sc = fx_->run(t.re, t.im, t.sre, t.sim, false);
            const float *zr = sc ? t.sre : t.re, *zi = sc ? t.sim : t.im;
            float *ur = sc ? t.re : t.sre, *ui = sc ? t.im : t.sim;  // the free pair: [h + 1][kW]
            // X[k] = (Z[k] + conj Z[h-k]) / 2 + e^{-2 pi i k / nx} * (Z[k] - conj Z[h-k]) / (2 i),  Z[h] = Z[0];
            // all 16 rows of the tile at once (unit stride over the lanes), then a transposing store
            untangle_forward(zr, zi, ur, ui, h, uc, us);
            for (int l = 0; l < w; ++l) {
                float2 *out = spec + (r0 + l) * nxh1;
                for (int k = 0; k <= h; ++k) out[k] = make_float2(ur[(size_t)k * kW + l], ui[(size_t)k * kW + l]);
            }
        }
    });
    columns(spec, false, false);
    columns(spec, true, false);
}

void Rfft3D::inverse(float2 *spec, float *real) const {
    columns(spec, true, true);
    columns(spec, false, true);
    const long long nx = box_.nx, rows = box_.ny * box_.nz, nxh1 = nx / 2 + 1;
    const int h = (int)nx / 2;
    const long long tiles = (rows + kW - 1) / kW;
    const float *uc = unt_c_.data(), *us = unt_s_.data();
    parallel_for(tiles, threads_, [&](long long b, long long e, int) {
        Tile t(h + 1);
        for (long long it = b; it < e; ++it) {
            const long long r0 = it * kW;
            const int w = (int)std::min<long long>(kW, rows - r0);
            // transposing load of the 16 spectrum rows into [h + 1][kW] planes ...
            for (int l = 0; l < kW; ++l) {
                if (l < w) {
                    const float2 *in = spec + (r0 + l) * nxh1;
                    for (int k = 0; k <= h; ++k) {
                        t.sre[(size_t)k * kW + l] = in[k].x;
                        t.sim[(size_t)k * kW + l] = in[k].y;
                    }
                } else {
                    for (int k = 0; k <= h; ++k) t.sre[(size_t)k * kW + l] = t.sim[(size_t)k * kW + l] = 0.f;
                }
            }
            // ... Z[k] = (X[k] + conj X[h-k]) + i e^{+2 pi i k / nx} (X[k] - conj X[h-k]),  k = 0 .. h-1, all lanes at once
            untangle_inverse(t.sre, t.sim, t.re, t.im, h, uc, us);
            const bool sc = fx_->run(t.re, t.im, t.sre, t.sim, true);
            const float *zr = sc ? t.sre : t.re, *zi = sc ? t.sim : t.im;
            for (int l = 0; l < w; ++l) {
                float *row = real + (r0 + l) * nx;
                for (int j = 0; j < h; ++j) {
                    row[2 * j] = zr[(size_t)j * kW + l];
                    row[2 * j + 1] = zi[(size_t)j * kW + l];
                }
            }
        }
    });
}

}  // namespace cpu
}  // namespace mil
