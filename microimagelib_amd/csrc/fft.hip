// LDS-tiled 3-D real FFT / fused convolution kernels for gfx950 (see fftconv.h).
//
// The column kernels (y pass, z convolution) work on a tile of N x 16 complex64 held in LDS as
// float2 tile[N][17] (the +1 column of padding keeps both the row-wise staging and the column-wise
// butterflies at <= 2-way bank conflicts; for the packed DC/Nyquist column it also carries the
// Hermitian partner column).  A 16-complex tile row is one 128-byte line of HBM: 8 lanes x float4,
// so a wave moves 8 lines per instruction.  The x kernel holds 8 volume rows as tile[nx/2][9].
//
// Transforms are in-place mixed-radix Cooley-Tukey (radix 8 first, then 4 / 2, then 3, 5, 7, 11, 13):
//   DIF  natural order in  -> digit-reversed out   (forward y / z, inverse x)
//   DIT  digit-reversed in -> natural order out    (inverse y / z, forward x)
// DIT is the exact transpose of DIF, so forward-DIF followed by inverse-DIT needs no
// reordering pass: the spectrum simply lives in digit-reversed y/z order, and the OTF is
// produced by the same forward kernels so it is scrambled identically.
// All transforms are unnormalised, like cuFFT / FFTW in the reference.
#include "fftconv.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace mil {
namespace fx {

namespace {

#include "fft_dev.h"

// ---------------------------------------------------------------------------
// y pass / z convolution: N x 16 column tiles
// ---------------------------------------------------------------------------
enum ColMode { kColForward = 0, kColInverse = 1, kColConv = 2, kColToOtf = 3 };

struct ColArgs {
    float2 *S;             // spectrum (in place)
    const float2 *otf;     // kColConv
    float2 *otf_out;       // kColToOtf
    const float2 *tw;      // exp(-2*pi*i*k/N)
    long long row_stride;  // float2 elements between consecutive tile rows
    long long outer_stride;  // float2 elements between tiles of consecutive outer index
    int groups;            // xs / 16
    int mode;
    int reverse;           // walk the tiles last-to-first (ping-pong with the previous kernel: Infinity Cache reuse)
    // z axis only: column 0 packs the DC and Nyquist x-planes, G = F0 + i*FN (both are spectra of real planes).
    // Their OTF product needs the Hermitian partner conj(G(-ky, -kz)):
    const unsigned short *partner;  // tile row holding -ky, per outer index (scrambled y positions)
    const unsigned short *neg;      // position holding -kz, per scrambled z position
    const float2 *otf_q;            // kColConv: Q plane [outer][N]
    float2 *otf_q_out;              // kColToOtf
    // Copy of column 0, [z][ky]: written by the forward y pass, read by the z kernels for the partner row.
    // (z-conv works in place, so the partner's column 0 in S may already hold that row's RESULT.)
    float2 *col0;
    int col0_stride;  // ny
    int ablate;       // timing experiments only (MIL_FX_ABLATE & 16: skip the OTF product, & 32: skip both z transforms)
    int otf_cols;     // kColToOtf: width of the OTF's column groups, [ky][group][kz][otf_cols] (16 = this file's own tiling; 8 / 4 = one
                      // block per workgroup of the two-pass yz kernels, see ConvPlan::set_yz_otf_layout)
};

template <int N, int NT, int MODE>
__global__ __launch_bounds__(NT) void col_kernel(ColArgs a) {
    __shared__ float2 tile[N * kWP];
    __shared__ float2 tw[N];
    const unsigned bid = a.reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int g = bid % a.groups;
    const long long outer = bid / a.groups;
    float2 *base = a.S + outer * a.outer_stride + (long long)g * kW;
    const long long otf_tile = (long long)bid * N * kW;  // [outer][g][N][16]
    constexpr bool kZ = (MODE == kColConv || MODE == kColToOtf);
    const bool packed = kZ && g == 0;  // this tile's column 0 is the packed DC/Nyquist column (wave-uniform)
    for (int k = threadIdx.x; k < N; k += NT) tw[k] = a.tw[k];
    tile_load<N, NT>(tile, base, a.row_stride);
    if constexpr (kZ) {
        if (packed) {  // 17th column: column 0 of the row that holds -ky (the LDS padding column carries it)
            const float2 *pbase = a.col0 + a.partner[outer];
            for (int k = threadIdx.x; k < N; k += NT) tile[k * kWP + kW] = pbase[(long long)k * a.col0_stride];
        }
    }
    // z-conv: when one batch of registers covers the workgroup's OTF block, request it now so that it arrives
    // under the forward transform instead of stalling between the two transforms
    constexpr bool kPrefetch = (MODE == kColConv) && (N * 8 <= NT * kBatch);
    float4 o[kBatch];
    if constexpr (kPrefetch) {
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int idx = j * NT + (int)threadIdx.x;
            if (idx < N * 8) o[j] = *reinterpret_cast<const float4 *>(a.otf + otf_tile + 2 * idx);
        }
    }
    lds_barrier();
    if constexpr (MODE == kColInverse) {
        fft_dit<N, N, +1, kW, NT>(tile, tw);
        tile_store<N, NT>(tile, base, a.row_stride);
        return;
    }
    if constexpr (kZ) {
        if (a.ablate & 32) {
        } else if (packed) fft_dif<N, N, -1, kW + 1, NT, kWP>(tile, tw);  // 17 columns, same LDS stride
        else fft_dif<N, N, -1, kW, NT>(tile, tw);
    } else {
        fft_dif<N, N, -1, kW, NT>(tile, tw);
    }
    if constexpr (MODE == kColForward) {
        tile_store<N, NT>(tile, base, a.row_stride);
        if (g == 0 && a.col0) {  // y pass: outer = z, tile row = scrambled ky
            for (int k = threadIdx.x; k < N; k += NT) a.col0[outer * N + k] = tile[k * kWP];
        }
        return;
    }
    if constexpr (MODE == kColToOtf) {
        // G = H0 + i*HN for the PSF.  Store P = (H0+HN)/2 in the column-0 slot and Q = (H0-HN)/2 aside, so that the
        // product of a packed column is  P*G + Q*conj(G(-k))  (two complex multiplies, no un-mixing at run time).
        if (packed) {
            float2 pv[(N + NT - 1) / NT], qv[(N + NT - 1) / NT];
#pragma unroll
            for (int i = 0; i < (N + NT - 1) / NT; ++i) {
                const int k = i * NT + (int)threadIdx.x;
                if (k < N) {
                    const float2 gk = tile[k * kWP], gp = tile[a.neg[k] * kWP + kW];
                    const float2 gb = make_float2(gp.x, -gp.y);                                   // conj(G(-k))
                    const float2 h0 = make_float2(0.5f * (gk.x + gb.x), 0.5f * (gk.y + gb.y));    // (G + Gb)/2
                    const float2 hn = make_float2(0.5f * (gk.y - gb.y), -0.5f * (gk.x - gb.x));   // (G - Gb)/(2i)
                    pv[i] = make_float2(0.5f * (h0.x + hn.x), 0.5f * (h0.y + hn.y));
                    qv[i] = make_float2(0.5f * (h0.x - hn.x), 0.5f * (h0.y - hn.y));
                }
            }
            lds_barrier();
#pragma unroll
            for (int i = 0; i < (N + NT - 1) / NT; ++i) {
                const int k = i * NT + (int)threadIdx.x;
                if (k < N) {
                    tile[k * kWP] = pv[i];
                    a.otf_q_out[outer * N + k] = qv[i];
                }
            }
            lds_barrier();
        }
        if (a.otf_cols == kW) {
            tile_store<N, NT>(tile, a.otf_out + otf_tile, kW);
        } else {  // the 16 columns of this tile as 16 / otf_cols narrower groups, [kz][otf_cols] each
            const int oc = a.otf_cols;
            float2 *dst = a.otf_out + otf_tile;  // the groups of this (ky, 16-column group) follow each other: same N * 16 elements
            for (int idx = (int)threadIdx.x; idx < N * 8; idx += NT) {
                const int kz = idx >> 3, c2 = 2 * (idx & 7);
                const float2 *q = tile + kz * kWP + c2;
                const float2 x = q[0], y = q[1];
                *reinterpret_cast<float4 *>(dst + ((long long)(c2 / oc) * N + kz) * oc + c2 % oc) = make_float4(x.x, x.y, y.x, y.y);
            }
        }
        return;
    }
    if constexpr (MODE == kColConv) {
        float2 keep[(N + NT - 1) / NT];  // packed column: Q * conj(G(-k)), added after the plain product
        if (packed) {
#pragma unroll
            for (int i = 0; i < (N + NT - 1) / NT; ++i) {
                const int k = i * NT + (int)threadIdx.x;
                if (k < N) {
                    const float2 gp = tile[a.neg[k] * kWP + kW];
                    keep[i] = c_mul(a.otf_q[outer * N + k], make_float2(gp.x, -gp.y));
                }
            }
        }
        if (a.ablate & 16) {
        } else if constexpr (kPrefetch) {
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int idx = j * NT + (int)threadIdx.x;
                if (idx < N * 8) {
                    float2 *q = tile + (idx >> 3) * kWP + 2 * (idx & 7);
                    q[0] = c_mul(q[0], make_float2(o[j].x, o[j].y));
                    q[1] = c_mul(q[1], make_float2(o[j].z, o[j].w));
                }
            }
        } else {
            tile_mul<N, NT>(tile, a.otf + otf_tile);
        }
        if (packed) {
            lds_barrier();  // column 0 now holds P*G
#pragma unroll
            for (int i = 0; i < (N + NT - 1) / NT; ++i) {
                const int k = i * NT + (int)threadIdx.x;
                if (k < N) tile[k * kWP] = c_add(tile[k * kWP], keep[i]);
            }
        }
        lds_barrier();
        if (!(a.ablate & 32)) fft_dit<N, N, +1, kW, NT>(tile, tw);
        tile_store<N, NT>(tile, base, a.row_stride);
    }
}

// ---------------------------------------------------------------------------
// x pass: RB (= 8) volume rows per workgroup, tile[k][row], packed real <-> complex
// ---------------------------------------------------------------------------
struct XArgs {
    float2 *S;
    const float *rin;    // kXForward
    float *rout;         // kXInverse output / estimate E (in-out) for the update modes
    const float *M;      // kXRatio
    const float2 *tw;    // exp(-2*pi*i*k/nx), k in [0, nx)
    const unsigned short *pos;  // pos[n], n in [0, nx/2)
    float eps;
    int reverse;
    int ablate;  // timing experiments only (MIL_FX_ABLATE): 1 skip x^-1 FFT, 2 skip x FFT
};

// RB rows per workgroup; LDS tile[k][row] with row stride RB + 1
template <int NH, int MODE, int RB>
__global__ __launch_bounds__(kNT) void x_kernel(XArgs a) {
    constexpr int NX = 2 * NH;
    constexpr int XS = ((NH + kW - 1) / kW) * kW + kW;  // row pitch: the columns plus one never-touched group (see ConvPlan)
    constexpr int WP = RB + 1;
    __shared__ float2 tile[NH * WP];
    __shared__ float2 tw[NH];            // w_NH^k = w_NX^(2k)
    __shared__ unsigned short pos[NH];
    const unsigned bid = a.reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const long long row0 = (long long)bid * RB;
    for (int k = threadIdx.x; k < NH; k += kNT) {
        tw[k] = a.tw[2 * k];
        pos[k] = a.pos[k];
    }
    if constexpr (MODE != kXForward) {
        // ---- spectrum rows -> packed half-length complex sequence Z' in tile[k][row].
        // Spectrum columns are stored in mirrored pairs: column 2j = X[j], column 2j+1 = X[NH-j] (1 <= j < NH/2), so one
        // float4 is exactly the pair the Hermitian fold needs.  Column 0 packs the two REAL row values (X[0], X[NH]) into
        // one complex number and column 1 holds the self-paired X[NH/2]: NH columns in all, no padding column.
        {
            constexpr int TOTAL = RB * (NH / 2);
            const float2 *src = a.S + row0 * XS;
#pragma unroll 1
            for (int it0 = 0; it0 < TOTAL; it0 += kNT * kBatch) {
                float4 v[kBatch];
                float2 w[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int idx = it0 + j * kNT + (int)threadIdx.x;
                    if (idx < TOTAL) {
                        const int row = idx / (NH / 2), k = idx % (NH / 2);
                        v[j] = *reinterpret_cast<const float4 *>(src + (long long)row * XS + 2 * k);
                        w[j] = a.tw[k];
                    }
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int idx = it0 + j * kNT + (int)threadIdx.x;
                    if (idx < TOTAL) {
                        const int row = idx / (NH / 2), k = idx % (NH / 2);
                        float2 x = make_float2(v[j].x, v[j].y), y = make_float2(v[j].z, v[j].w);
                        if (k == 0) {  // packed column: X[0] = (re, 0), X[NH] = (im, 0); column 1 = X[NH/2]
                            tile[(NH / 2) * WP + row] = make_float2(2.f * y.x, -2.f * y.y);  // Z'[NH/2] = 2 conj(X[NH/2])
                            y = make_float2(x.y, 0.f);
                            x.y = 0.f;
                        }
                        const float sx = x.x + y.x, sy = x.y - y.y, dx = x.x - y.x, dy = x.y + y.y;
                        const float ux = w[j].x * dx + w[j].y * dy, uy = w[j].x * dy - w[j].y * dx;
                        tile[k * WP + row] = make_float2(sx - uy, sy + ux);
                        if (k != 0) tile[(NH - k) * WP + row] = make_float2(sx + uy, ux - sy);
                    }
                }
            }
        }
        lds_barrier();
        if (!(a.ablate & 1)) fft_dif<NH, NH, +1, RB>(tile, tw);  // element n of z' now sits at position pos[n]
        // ---- real-space epilogue, 4 consecutive x per lane
        {
            constexpr int TOTAL = RB * (NH / 2);
            for (int idx = threadIdx.x; idx < TOTAL; idx += kNT) {
                const int row = idx / (NH / 2), q = idx % (NH / 2);
                float2 *p0 = tile + pos[2 * q] * WP + row, *p1 = tile + pos[2 * q + 1] * WP + row;
                const float2 z0 = *p0, z1 = *p1;
                float4 t = make_float4(z0.x, z0.y, z1.x, z1.y);
                const long long off = (row0 + row) * NX + 4 * q;
                if (MODE == kXInverse) {
                    *reinterpret_cast<float4 *>(a.rout + off) = t;
                } else if (MODE == kXRatio) {  // div3Dkernel: T = A / T (unguarded)
                    const float4 m = *reinterpret_cast<const float4 *>(a.M + off);
                    *p0 = make_float2(m.x / t.x, m.y / t.y);
                    *p1 = make_float2(m.z / t.z, m.w / t.w);
                } else {  // multi3Dkernel + maxvalue3Dgpukernel: E = max(E * T, eps)
                    float4 e = *reinterpret_cast<const float4 *>(a.rout + off);
                    e.x = e.x * t.x; e.y = e.y * t.y; e.z = e.z * t.z; e.w = e.w * t.w;
                    e.x = (e.x > a.eps) ? e.x : a.eps;
                    e.y = (e.y > a.eps) ? e.y : a.eps;
                    e.z = (e.z > a.eps) ? e.z : a.eps;
                    e.w = (e.w > a.eps) ? e.w : a.eps;
                    *reinterpret_cast<float4 *>(a.rout + off) = e;
                    *p0 = make_float2(e.x, e.y);
                    *p1 = make_float2(e.z, e.w);
                }
            }
        }
        if constexpr (MODE == kXInverse || MODE == kXUpdateLast) return;
        lds_barrier();
    } else {
        // ---- real rows -> packed complex at digit-reversed positions
        constexpr int TOTAL = RB * (NH / 2);
        const float *src = a.rin + row0 * NX;
#pragma unroll 1
        for (int it0 = 0; it0 < TOTAL; it0 += kNT * kBatch) {
            float4 v[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int idx = it0 + j * kNT + (int)threadIdx.x;
                if (idx < TOTAL) v[j] = *reinterpret_cast<const float4 *>(src + 4 * idx);
            }
            if (it0 == 0) lds_barrier();  // pos[] staged
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int idx = it0 + j * kNT + (int)threadIdx.x;
                if (idx < TOTAL) {
                    const int row = idx / (NH / 2), q = idx % (NH / 2);
                    tile[pos[2 * q] * WP + row] = make_float2(v[j].x, v[j].y);
                    tile[pos[2 * q + 1] * WP + row] = make_float2(v[j].z, v[j].w);
                }
            }
        }
        lds_barrier();
    }
    // ---- forward: packed complex FFT, then split into the Hermitian half spectrum
    if (!(a.ablate & 2)) fft_dit<NH, NH, -1, RB>(tile, tw);
    // ---- split into the Hermitian half spectrum and store the mirrored pairs (X[k], X[NH-k])
    {
        constexpr int TOTAL = RB * (NH / 2);
        float2 *dst = a.S + row0 * XS;
#pragma unroll 4
        for (int idx = threadIdx.x; idx < TOTAL; idx += kNT) {
            const int row = idx / (NH / 2), k = idx % (NH / 2);
            const float2 x = tile[k * WP + row], y = tile[(k == 0 ? 0 : NH - k) * WP + row];
            const float2 w = a.tw[k];
            const float sx = x.x + y.x, sy = x.y - y.y, dx = x.x - y.x, dy = x.y + y.y;
            const float tx = w.x * dx - w.y * dy, ty = w.x * dy + w.y * dx;
            float4 o = make_float4(0.5f * (sx + ty), 0.5f * (sy - tx), 0.5f * (sx - ty), 0.5f * (-sy - tx));
            if (k == 0) {  // (X[0], X[NH]) are real: pack them; X[NH/2] = conj(Z[NH/2]) rides in column 1
                const float2 m = tile[(NH / 2) * WP + row];
                o = make_float4(o.x, o.z, m.x, -m.y);
            }
            *reinterpret_cast<float4 *>(dst + (long long)row * XS + 2 * k) = o;
        }
    }
}

// ---------------------------------------------------------------------------
// test aid: private OTF order -> natural order
// ---------------------------------------------------------------------------
__global__ void otf_to_natural_kernel(float2 *__restrict__ nat, const float2 *__restrict__ otf,
                                      const float2 *__restrict__ otf_q, const unsigned short *__restrict__ freq_y,
                                      const unsigned short *__restrict__ freq_z, int nxh1, int ny, int nz, int groups, int cols) {
    // one thread per private element (ky_s, g, kz_s, c): `groups` column groups of `cols` columns
    const long long total = (long long)ny * groups * nz * cols;
    const int nh = nxh1 - 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        long long r = i / cols;
        const int pz = (int)(r % nz);
        r /= nz;
        const int g = (int)(r % groups);
        const int py = (int)(r / groups);
        const int col = g * cols + c;  // packed / mirrored-pair column order of the x pass
        if (col >= nh) continue;
        float2 *row = nat + ((long long)freq_z[pz] * ny + freq_y[py]) * nxh1;
        const float2 v = otf[i];
        if (col == 0) {  // P = (H0+HN)/2, Q = (H0-HN)/2
            const float2 q = otf_q[(long long)py * nz + pz];
            row[0] = make_float2(v.x + q.x, v.y + q.y);
            row[nh] = make_float2(v.x - q.x, v.y - q.y);
        } else if (col == 1) {
            row[nh / 2] = v;
        } else {
            row[(col & 1) ? nh - (col >> 1) : (col >> 1)] = v;
        }
    }
}

bool pow2(long long n) { return n > 0 && (n & (n - 1)) == 0; }

// lengths the column kernels are instantiated for: powers of two and the 64*k sizes of snapTransformSize
bool col_length_ok(long long n) {
    if (n < 16 || n > 1024) return false;
    return pow2(n) || (n >= 192 && n % 64 == 0);
}

void freq_rec(int n, int pos_base, int freq_mul, int freq_add, unsigned short *out) {
    // position p of a length-n sub-transform holds frequency r*freq(p % m) + p / m
    if (n == 1) {
        out[pos_base] = (unsigned short)freq_add;
        return;
    }
    const int r = pick_radix(n);
    const int m = n / r;
    for (int k2 = 0; k2 < r; ++k2) freq_rec(m, pos_base + k2 * m, freq_mul * r, freq_add + k2 * freq_mul, out);
}

template <class T>
void upload(DeviceBuffer<T> &buf, const std::vector<T> &host) {
    buf.alloc(host.size());
    MIL_HIP(hipMemcpy(buf.get(), host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
}

std::vector<float2> twiddles(int n) {
    std::vector<float2> t((size_t)n);
    for (int k = 0; k < n; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)n;
        t[(size_t)k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    return t;
}

#ifdef MIL_FX_EXPERIMENTS
int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}
#endif

// Column workgroups: LDS (N x 17 complex) caps long transforms at 1-3 workgroups per CU, so they get 512
// threads to keep 16+ waves per CU in flight (measured at N = 512: 0.099 -> 0.092 ms per y pass);
// short ones fit 4+ workgroups of 256 threads.
template <int N>
constexpr int col_threads() {
    return N >= 320 ? 512 : kNT;
}

template <int N>
void launch_col(const ColArgs &a, long long blocks, hipStream_t s) {
    constexpr int NT = col_threads<N>();
    const dim3 grid((unsigned)blocks), block(NT);
    switch (a.mode) {
        case kColForward: hipLaunchKernelGGL((col_kernel<N, NT, kColForward>), grid, block, 0, s, a); break;
        case kColInverse: hipLaunchKernelGGL((col_kernel<N, NT, kColInverse>), grid, block, 0, s, a); break;
        case kColConv: hipLaunchKernelGGL((col_kernel<N, NT, kColConv>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((col_kernel<N, NT, kColToOtf>), grid, block, 0, s, a); break;
    }
}

void dispatch_col(int n, const ColArgs &a, long long blocks, hipStream_t s) {
    if (blocks <= 0 || blocks > 0x7fffffffLL) throw Error(-1, "fused FFT: grid out of range");
    switch (n) {
        case 16: launch_col<16>(a, blocks, s); break;
        case 32: launch_col<32>(a, blocks, s); break;
        case 64: launch_col<64>(a, blocks, s); break;
        case 128: launch_col<128>(a, blocks, s); break;
        case 256: launch_col<256>(a, blocks, s); break;
        case 512: launch_col<512>(a, blocks, s); break;
        case 1024: launch_col<1024>(a, blocks, s); break;
        case 192: launch_col<192>(a, blocks, s); break;
        case 320: launch_col<320>(a, blocks, s); break;
        case 384: launch_col<384>(a, blocks, s); break;
        case 448: launch_col<448>(a, blocks, s); break;
        case 576: launch_col<576>(a, blocks, s); break;
        case 640: launch_col<640>(a, blocks, s); break;
        case 704: launch_col<704>(a, blocks, s); break;
        case 768: launch_col<768>(a, blocks, s); break;
        case 832: launch_col<832>(a, blocks, s); break;
        case 896: launch_col<896>(a, blocks, s); break;
        case 960: launch_col<960>(a, blocks, s); break;
        default: throw Error(-1, "fused FFT: unsupported column length");
    }
    MIL_HIP(hipGetLastError());
}

template <int NH, int RB>
void launch_x(int mode, const XArgs &a, long long rows, hipStream_t s) {
    const dim3 grid((unsigned)(rows / RB)), block(kNT);
    switch (mode) {
        case kXForward: hipLaunchKernelGGL((x_kernel<NH, kXForward, RB>), grid, block, 0, s, a); break;
        case kXInverse: hipLaunchKernelGGL((x_kernel<NH, kXInverse, RB>), grid, block, 0, s, a); break;
        case kXRatio: hipLaunchKernelGGL((x_kernel<NH, kXRatio, RB>), grid, block, 0, s, a); break;
        case kXUpdate: hipLaunchKernelGGL((x_kernel<NH, kXUpdate, RB>), grid, block, 0, s, a); break;
        case kXUpdateLast: hipLaunchKernelGGL((x_kernel<NH, kXUpdateLast, RB>), grid, block, 0, s, a); break;
        default: throw Error(-1, "fused FFT: unknown x mode");
    }
}

void dispatch_x(int nh, int mode, const XArgs &a, long long rows, hipStream_t s) {
    if (rows <= 0 || rows % 16 || rows / 8 > 0x7fffffffLL) throw Error(-1, "fused FFT: row count out of range");
#define MIL_X_CASE(NHV)                      \
    case NHV:                                \
        launch_x<NHV, 8>(mode, a, rows, s);  \
        break;
    switch (nh) {
        MIL_X_CASE(8)
        MIL_X_CASE(16)
        MIL_X_CASE(32)
        MIL_X_CASE(64)
        MIL_X_CASE(128)
        MIL_X_CASE(256)
        MIL_X_CASE(512)
        MIL_X_CASE(1024)
        MIL_X_CASE(96)
        MIL_X_CASE(160)
        MIL_X_CASE(192)
        MIL_X_CASE(224)
        MIL_X_CASE(288)
        MIL_X_CASE(320)
        MIL_X_CASE(352)
        MIL_X_CASE(384)
        MIL_X_CASE(416)
        MIL_X_CASE(448)
        MIL_X_CASE(480)
        default: throw Error(-1, "fused FFT: unsupported row length");
    }
#undef MIL_X_CASE
    MIL_HIP(hipGetLastError());
}


}  // namespace

void digit_reversal_freq(int n, unsigned short *freq_of_pos) { freq_rec(n, 0, 1, 0, freq_of_pos); }

void digit_reversal_pos(int n, unsigned short *pos_of_elem) {
    std::vector<unsigned short> f((size_t)n);
    digit_reversal_freq(n, f.data());
    for (int p = 0; p < n; ++p) pos_of_elem[f[(size_t)p]] = (unsigned short)p;
}

bool ConvPlan::supported(Dims b) {
    const bool x_ok = (pow2(b.nx) && b.nx >= 16 && b.nx <= 2048) || (b.nx >= 192 && b.nx <= 1024 && b.nx % 64 == 0);
    return x_ok && col_length_ok(b.ny) && col_length_ok(b.nz);
}

ConvPlan::ConvPlan(Dims box) : box_(box) {
    if (!supported(box)) throw Error(-1, "ConvPlan: unsupported box (need powers of two or multiples of 64, 16..1024, x up to 2048)");
    const int nx = (int)box.nx, ny = (int)box.ny, nz = (int)box.nz, nh = nx / 2;
    xs_ = ((nh + kW - 1) / kW) * kW;
    // Row pitch = columns + 16: nx/2 is a power of two for the common boxes, and power-of-two row / plane strides make the
    // strided y and z passes camp on a few HBM channels (measured: z-conv 5 % slower at pitch 256 than at 272).
    pitch_ = xs_ + kW;
    col0_.alloc((size_t)box.ny * box.nz);
#ifdef MIL_FX_EXPERIMENTS  // timing-only ablations (results invalid); compiled out of the product library
    ablate_ = env_int("MIL_FX_ABLATE", 0);
#endif
    upload(tw_x_, twiddles(nx));
    upload(tw_y_, twiddles(ny));
    upload(tw_z_, twiddles(nz));
    std::vector<unsigned short> p((size_t)nh), fy((size_t)ny), fz((size_t)nz);
    digit_reversal_pos(nh, p.data());
    digit_reversal_freq(ny, fy.data());
    digit_reversal_freq(nz, fz.data());
    upload(pos_x_, p);
    upload(freq_y_, fy);
    upload(freq_z_, fz);
    // position of the negated frequency, per position (Hermitian partner of the packed DC/Nyquist column)
    auto negated = [](const std::vector<unsigned short> &freq) {
        const int n = (int)freq.size();
        std::vector<unsigned short> pos((size_t)n), neg((size_t)n);
        for (int q = 0; q < n; ++q) pos[freq[(size_t)q]] = (unsigned short)q;
        for (int q = 0; q < n; ++q) neg[(size_t)q] = pos[(size_t)((n - freq[(size_t)q]) % n)];
        return neg;
    };
    upload(neg_y_, negated(fy));
    upload(neg_z_, negated(fz));
    init_two_pass();
}

void ConvPlan::x_pass(XMode mode, float2 *S, const float *rin, float *rout, const float *M, float eps,
                      hipStream_t s, bool reverse) const {
    if (!S || (mode == kXForward && !rin) || (mode != kXForward && mode != kXRatio && !rout) || (mode == kXRatio && !M))
        throw Error(-1, "x_pass: missing operand");
    XArgs a;
    a.S = S; a.rin = rin; a.rout = rout; a.M = M;
    a.tw = tw_x_.get(); a.pos = pos_x_.get(); a.eps = eps; a.reverse = reverse ? 1 : 0;
    a.ablate = ablate_;
    dispatch_x((int)box_.nx / 2, (int)mode, a, box_.ny * box_.nz, s);
}

void ConvPlan::y_pass(float2 *S, bool inverse, hipStream_t s, bool reverse) const {
    ColArgs a;
    a.S = S; a.otf = nullptr; a.otf_out = nullptr; a.tw = tw_y_.get();
    a.partner = nullptr; a.neg = nullptr; a.otf_q = nullptr; a.otf_q_out = nullptr;
    a.col0 = inverse ? nullptr : col0_.get(); a.col0_stride = (int)box_.ny;
    a.row_stride = pitch_;
    a.outer_stride = (long long)box_.ny * pitch_;
    a.groups = xs_ / kW;
    a.mode = inverse ? kColInverse : kColForward;
    a.reverse = reverse ? 1 : 0;
    a.ablate = 0;
    a.otf_cols = kW;
    dispatch_col((int)box_.ny, a, (long long)a.groups * box_.nz, s);
}

void ConvPlan::z_conv(float2 *S, const float2 *otf, hipStream_t s, bool reverse) const {
    if (otf_cols_ != kW) throw Error(-1, "z_conv: this plan's OTFs are tiled for the two-pass yz kernel (set_yz_otf_layout)");
    ColArgs a;
    a.S = S; a.otf = otf; a.otf_out = nullptr; a.tw = tw_z_.get();
    a.partner = neg_y_.get(); a.neg = neg_z_.get();
    a.otf_q = otf + (size_t)xs_ * box_.ny * box_.nz; a.otf_q_out = nullptr;
    a.col0 = col0_.get(); a.col0_stride = (int)box_.ny;
    a.row_stride = (long long)box_.ny * pitch_;
    a.outer_stride = pitch_;
    a.groups = xs_ / kW;
    a.mode = kColConv;
    a.reverse = reverse ? 1 : 0;
    a.ablate = ablate_;
    a.otf_cols = kW;
    dispatch_col((int)box_.nz, a, (long long)a.groups * box_.ny, s);
}

void ConvPlan::z_forward_to_otf(const float2 *S, float2 *otf, hipStream_t s) const {
    ColArgs a;
    a.S = const_cast<float2 *>(S); a.otf = nullptr; a.otf_out = otf; a.tw = tw_z_.get();
    a.partner = neg_y_.get(); a.neg = neg_z_.get();
    a.otf_q = nullptr; a.otf_q_out = otf + (size_t)xs_ * box_.ny * box_.nz;
    a.col0 = col0_.get(); a.col0_stride = (int)box_.ny;
    a.row_stride = (long long)box_.ny * pitch_;
    a.outer_stride = pitch_;
    a.groups = xs_ / kW;
    a.mode = kColToOtf;
    a.reverse = 0;
    a.ablate = 0;
    a.otf_cols = otf_cols_;
    dispatch_col((int)box_.nz, a, (long long)a.groups * box_.ny, s);
}

void ConvPlan::otf_to_natural(float2 *natural, const float2 *otf, hipStream_t s) const {
    const int nxh1 = (int)box_.nx / 2 + 1;
    hipLaunchKernelGGL(otf_to_natural_kernel, dim3(2048), dim3(256), 0, s, natural, otf,
                       otf + (size_t)xs_ * box_.ny * box_.nz, freq_y_.get(), freq_z_.get(), nxh1, (int)box_.ny,
                       (int)box_.nz, xs_ / otf_cols_, otf_cols_);
    MIL_HIP(hipGetLastError());
}

}  // namespace fx
}  // namespace mil
