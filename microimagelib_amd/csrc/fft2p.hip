// Two passes per convolution instead of four (see fftconv.h, "two-pass path").
//
// The classic sequence of one half view update is  y | z-conv | y^-1 | [x^-1 . op . x]  -- four trips of the spectrum
// through HBM.  The y transform of length 512 is three radix-8 stages; its LAST stage only combines 8 ADJACENT rows and
// its first two only rows with the same y mod 8.  So the last stage moves into the z kernel and the first two into the
// x kernel:
//
//   yz_conv_kernel   8 adjacent rows x 16 columns x all z per workgroup:  y stage(8) . z . OTF . z^-1 . y stage(8)^-1
//   xy_kernel        the 64 rows y = n2 + 8m of one z plane per workgroup:  y head^-1 . x^-1 . op . x . y head
//
// Both tiles (256 KB / 128 KB at nz = 256) are larger than the LDS, so they live in REGISTERS (64 complex per thread); LDS
// only carries the exchanges between "all z of one (row, column)" and "all rows of one (z, column)" ownership.  Positions,
// twiddles and the OTF tiling are exactly those of the classic kernels (same in-place digit-reversed order), so both
// paths share OTFs and can be mixed stage by stage (MIL_FX_TWOPASS, tests/test_gpu_fft.py).
//
// Boxes.  Every column length of this library starts its in-place transform with TWO radix-8 stages (fft_dev.h, pick_radix),
// so for any ny = 64 T the "head" is 64 rows y = n2 + T m and the "tail" a transform of length T over T ADJACENT rows (one or
// two small stages, primes 3 .. 13 included); the xy kernel gives every spectrum column a thread and runs its x transform as
// NH = nx / 2 = 16 x M: LDS stage(s) of length M and a radix-16 stage in registers, fused with the real-space operation.
//   xy_kernel<MODE, NY, NH>   this file: nx = 256, 512, 1024 and ny = 128 .. 1024 in powers of two, as compile-time constants;
//                             fft2p_any_xy.hip: every nx = 192 .. 1024 and ny = 128 .. 1024 in multiples of 64 (the sizes
//                             snapTransformSize produces, reference src/api_subfunc.cu:57-87), ny a run-time argument
//   yz_conv_kernel            this file: ny = 512, nz = 256 -- register tiles of 8 rows x 8 columns x all z, two workgroups per CU
//   yz_lds_kernel<NY, NZ>     fft2p_any_yz.hip: the tile of T rows x 2 / 4 / 8 / 16 columns x all z FITS the LDS (T x nz <= 4096) and
//                             the same sequence runs on the classic LDS stage code; any nx
// The narrow tiles are dealt so that the pieces of a 128-byte row segment land on the same XCD.
// => nx = 192 .. 1024, ny = 128 .. 1024, nz = 64 .. 1024 in multiples of 64 with (ny / 64) x nz <= 4096, ... x 512 x 256 on register tiles for
// nx = 256, 512, 1024.  Larger yz tiles (ny x nz = 1024 x 512, 1024 x 1024 ...) keep the classic sequence.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "fft2p.hip is written for gfx950 (v_permlane32_swap, 160 KB of LDS per workgroup): build with --offload-arch=gfx950"
#endif
#include "fft2p_any.h"
#include "fftconv.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace mil {
namespace fx {

namespace {

#include "fft2p_dev.h"

unsigned long long *g_stamps = nullptr;  // set through mil_fx_set_stamps (timing experiments)

}  // namespace

namespace {
// boxes this file's instantiations (NH and NY compile-time constants) cover: nx, ny powers of two ...
bool static_xy(Dims b) {
    return (b.nx == 256 || b.nx == 512 || b.nx == 1024) && (b.ny == 1024 || b.ny == 512 || b.ny == 256 || b.ny == 128);
}
// The register yz kernel (ny = 512, nz = 256, nx a power of two) was THE yz kernel of rounds 2 - 5; at the end of round 5 the LDS
// kernel of fft2p_any_yz.hip (4-column tiles paired on an XCD, fused y tail, OTF requested up front, per-workgroup OTF tiling)
// overtook it on that very box -- 175 against 189 us per launch, 1 411 against 1 367 it/s at 512 x 512 x 256 -- and runs every
// (ny, nz) now.  MIL_FX_YZ_REGISTER=1 (read when a plan is created) brings the register kernel back for comparisons.
bool static_yz(Dims b) {
    if (!(static_xy(b) && b.ny == kNY && b.nz == kNZ)) return false;
    const char *e = getenv("MIL_FX_YZ_REGISTER");
    return e && e[0] == '1';
}
any::Box any_box(Dims b, int xs, int pitch) { return any::Box{(int)b.nx, (int)b.ny, (int)b.nz, xs, pitch}; }
}  // namespace

// nx = 192 .. 1024 and ny = 128 .. 1024 in multiples of 64; nz a multiple of 64 with (ny / 64) x nz <= 4096 -- the yz tile of
// ny / 64 rows x 2 columns x nz must fit the LDS twice over -- or 512 x 256 (register yz kernel, nx a power of two).  Powers of
// two in x and y run this file's xy instantiations, everything else those of fft2p_any_*.hip.
bool ConvPlan::two_pass_supported(Dims b) {
    if (b.nx % 64 || b.nx < 192 || b.nx > 1024 || b.ny % 64 || b.ny < 128 || b.ny > 1024) return false;
    if (static_yz(b)) return true;
    return any::yz_covers(any::Box{(int)b.nx, (int)b.ny, (int)b.nz, (int)b.nx / 2, (int)b.nx / 2 + kW});
}

int ConvPlan::yz_tile_cols(Dims b) {
    if (static_yz(b)) return 8;  // the register kernel's tiles
    return any::yz_cols(any_box(b, (int)b.nx / 2, (int)b.nx / 2 + kW));
}

void ConvPlan::set_yz_otf_layout(bool on) {
    if (on && !two_pass_) throw Error(-1, "set_yz_otf_layout: the box has no two-pass path");
    yz_otf_ = on;
    otf_cols_ = !on ? kW : yz_any_ ? any::yz_cols(any_box(box_, xs_, pitch_)) : 8;
}

void ConvPlan::init_two_pass() {
    if (!two_pass_supported(box_) || pitch_ != (int)box_.nx / 2 + kW || xs_ != (int)box_.nx / 2) return;
    const int ny = (int)box_.ny, T = ny / 64;
    // pairs of T-row blocks holding ky and -ky
    std::vector<unsigned short> fy((size_t)ny), pos((size_t)ny), pairs;
    digit_reversal_freq(ny, fy.data());
    for (int q = 0; q < ny; ++q) pos[fy[(size_t)q]] = (unsigned short)q;
    for (int b = 0; b < ny / T; ++b) {
        const int nb = pos[(size_t)((ny - fy[(size_t)(T * b)]) % ny)] / T;
        for (int j = 0; j < T; ++j)
            if (pos[(size_t)((ny - fy[(size_t)(T * b + j)]) % ny)] / T != nb) return;  // blocks do not pair up: classic path
        if (nb >= b) {
            pairs.push_back((unsigned short)b);
            pairs.push_back((unsigned short)nb);
        }
    }
    xy_any_ = !static_xy(box_);
    yz_any_ = !static_yz(box_);
    if (const char *e = getenv("MIL_FX_ANY")) {  // diagnostics: 1 / 2 / 3 = run-time-geometry xy / yz / both kernels wherever they cover the box
        const any::Box ab = any_box(box_, xs_, pitch_);
        if ((atoi(e) & 1) && any::xy_covers(ab)) xy_any_ = true;
        if ((atoi(e) & 2) && any::yz_covers(ab)) yz_any_ = true;
    }
    // every translation unit that launches kernels for this box fills its own constant tables
    fill_tw_tables(tw_y_.get(), ny, tw_z_.get(), (int)box_.nz);
    if (xy_any_) any::xy_fill_tables(tw_y_.get(), ny);
    if (yz_any_) any::yz_fill_tables(tw_y_.get(), ny);
    n_pairs_ = (int)pairs.size() / 2;
    pairs_.alloc(pairs.size());
    MIL_HIP(hipMemcpy(pairs_.get(), pairs.data(), pairs.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    two_pass_ = true;
    // non-temporal hints on the once-per-iteration streams (Aux<STREAM>) where the working set -- two OTFs, image, estimate,
    // spectrum: five volumes -- is far beyond the 256 MB Infinity Cache; MIL_FX_STREAM=0 / 1 forces the choice
    stream_ = box_.count() * 4 >= kStreamVolumeBytes;
    if (const char *e = getenv("MIL_FX_STREAM")) stream_ = e[0] == '1';
}

namespace {
// run f(std::integral_constant<int, NH>) for the plan's NH
template <class F>
void with_nh(long long nx, F &&f) {
    if (nx == 256) f(std::integral_constant<int, 128>{});
    else if (nx == 512) f(std::integral_constant<int, 256>{});
    else if (nx == 1024) f(std::integral_constant<int, 512>{});
    else throw Error(-1, "two-pass path: nx must be 256, 512 or 1024");
}
}  // namespace

void ConvPlan::y_part(float2 *S, int part, hipStream_t s) const {
    if (!two_pass_) throw Error(-1, "two-pass path: unsupported box");
    if (part < 0 || part > 3) throw Error(-1, "y_part: unknown part");
    if (any::y_part_covers((int)box_.ny)) {
        any::y_part(any_box(box_, xs_, pitch_), S, tw_y_.get(), col0_.get(), part, s);
        return;
    }
    PartArgs a;
    a.S = S; a.tw = tw_y_.get(); a.col0 = (part == 2) ? col0_.get() : nullptr;
    a.row_stride = pitch_; a.outer_stride = (long long)box_.ny * pitch_; a.groups = xs_ / kW;
    const dim3 grid((unsigned)(a.groups * box_.nz));
#define MIL_PART(N, NT)                                                                                        \
    switch (part) {                                                                                            \
        case 0: hipLaunchKernelGGL((col_part_kernel<N, NT, 0>), grid, dim3(NT), 0, s, a); break;               \
        case 1: hipLaunchKernelGGL((col_part_kernel<N, NT, 1>), grid, dim3(NT), 0, s, a); break;               \
        case 2: hipLaunchKernelGGL((col_part_kernel<N, NT, 2>), grid, dim3(NT), 0, s, a); break;               \
        default: hipLaunchKernelGGL((col_part_kernel<N, NT, 3>), grid, dim3(NT), 0, s, a); break;              \
    }
    if (box_.ny == 1024) { MIL_PART(1024, 512) }
    else if (box_.ny == 512) { MIL_PART(512, 512) }
    else if (box_.ny == 256) { MIL_PART(256, 256) }
    else { MIL_PART(128, 256) }
#undef MIL_PART
    MIL_HIP(hipGetLastError());
}

void ConvPlan::yz_conv(float2 *S, const float2 *otf, hipStream_t s) const {
    if (!two_pass_) throw Error(-1, "two-pass path: unsupported box");
    if (!yz_otf_) throw Error(-1, "yz_conv: this plan's OTFs are tiled for z_conv (set_yz_otf_layout)");
    if (yz_any_) {
        any::yz_conv(any_box(box_, xs_, pitch_), S, otf, neg_y_.get(), neg_z_.get(), pairs_.get(), n_pairs_, tw_z_.get(), s);
        return;
    }
    YzArgs a;
    a.S = S; a.otf = otf; a.otf_q = otf + (size_t)xs_ * box_.ny * box_.nz;
    a.neg_y = neg_y_.get(); a.neg_z = neg_z_.get(); a.pairs = pairs_.get();
    a.pair_blocks = n_pairs_;
    a.stamps = g_stamps;
    a.pitch = pitch_; a.groups = xs_ / kW; a.otf_stream = 0;
    // The register kernel of 512 x 256.  Its tiles are 8 columns wide: half the LDS per workgroup, so a CU holds two whose load /
    // exchange / store phases overlap, and the two halves of each 128-byte row segment go to workgroups t and t + 8 (same XCD, same
    // L2) so HBM still moves whole lines (profiles/r03/bench_box_paired8.txt: +5 % against unpaired 16-column tiles; unpaired
    // 8-column tiles gain nothing -- the 64-byte segments cost what the overlap wins: bench_box_narrow.txt).
    const dim3 grid((unsigned)(64 * (xs_ / 8) + n_pairs_));  // 64 blocks of 8 rows
    with_nh(box_.nx, [&](auto nh_tag) {
        constexpr int NH = decltype(nh_tag)::value;
        if (stream_) hipLaunchKernelGGL((yz_conv_kernel<NH, 8, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((yz_conv_kernel<NH, 8, false>), grid, dim3(256), 0, s, a);
    });
    MIL_HIP(hipGetLastError());
}

void ConvPlan::xy_pass(XMode mode, float2 *S, float *E, const float *M, float eps, hipStream_t s) const {
    if (!two_pass_) throw Error(-1, "two-pass path: unsupported box");
    if (!S || (mode == kXRatio ? !M : !E)) throw Error(-1, "xy_pass: missing operand");
    if (mode != kXRatio && mode != kXUpdate && mode != kXUpdateLast) throw Error(-1, "xy_pass: mode must be ratio / update");
    if (xy_any_) {
        any::xy_pass(any_box(box_, xs_, pitch_), (int)mode, S, E, M, tw_x_.get(), eps, stream_, s);
        return;
    }
    XyArgs a;
    a.S = S; a.E = E; a.M = M; a.tw_x = tw_x_.get(); a.eps = eps;
    a.spec_bytes = (long long)pitch_ * box_.ny * box_.nz * 8;
    a.real_bytes = (long long)box_.nx * box_.ny * box_.nz * 4;
    a.stamps = (g_stamps && mode == kXRatio) ? g_stamps + 2048 * 16 : nullptr;
    a.t = (int)box_.ny / 64;
    const int T = (int)box_.ny / 64;
    const dim3 grid((unsigned)(box_.nz * T));
    with_nh(box_.nx, [&](auto nh_tag) {
        constexpr int NH = decltype(nh_tag)::value;
        const dim3 block(NH);
#define MIL_XY_S(NY, STREAM)                                                                                     \
    switch (mode) {                                                                                              \
        case kXRatio: hipLaunchKernelGGL((xy_kernel<kXRatio, NY, NH, STREAM>), grid, block, 0, s, a); break;     \
        case kXUpdate: hipLaunchKernelGGL((xy_kernel<kXUpdate, NY, NH, STREAM>), grid, block, 0, s, a); break;   \
        default: hipLaunchKernelGGL((xy_kernel<kXUpdateLast, NY, NH, STREAM>), grid, block, 0, s, a); break;     \
    }
#define MIL_XY(NY)                 \
    if (stream_) {                 \
        MIL_XY_S(NY, true)         \
    } else {                       \
        MIL_XY_S(NY, false)        \
    }
        if (box_.ny == 1024) { MIL_XY(1024) }
        else if (box_.ny == 512) { MIL_XY(512) }
        else if (box_.ny == 256) { MIL_XY(256) }
        else { MIL_XY(128) }
#undef MIL_XY
#undef MIL_XY_S
    });
    MIL_HIP(hipGetLastError());
}

}  // namespace fx
}  // namespace mil

// timing experiments (tools/twopass_stamps.py): device buffer of 2 * 2048 * 16 uint64 the two-pass kernels write wall-clock
// stamps into, or null
extern "C" int mil_fx_set_stamps(unsigned long long *d_stamps) {
    mil::fx::g_stamps = d_stamps;
    return 0;
}
