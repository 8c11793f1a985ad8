// Fused 3-D real convolution for the Richardson-Lucy loop, hand-written for gfx950.
//
// The reference runs every view update as  cufftExecR2C -> multicomplex3Dgpu -> cufftExecC2R ->
// div3Dgpu -> cufftExecR2C -> multicomplex3Dgpu -> cufftExecC2R -> multi3Dgpu -> maxvalue3Dgpu
// (src/api_subfunc.cu:3404-3416): 4 library FFTs (3-4 kernels each) + 5 element-wise passes.
// Here the transforms are our own LDS-tiled kernels so the element-wise work rides inside them:
//
//   x pass   rows of the volume; packed real<->complex transform; the RL ratio / estimate update
//            sits between the inverse and the next forward x transform of the SAME rows
//   y pass   128-byte column tiles, in place
//   z conv   forward z transform, complex OTF product, inverse z transform in one kernel, in place
//
// One view update = 8 launches moving 18 C' + 3 R bytes instead of ~26 C + 12 R (C' = packed half
// spectrum = R bytes, R = real volume).  Spectra live in a PRIVATE layout: nx/2 complex columns (the real
// DC and Nyquist row values share column 0, mirrored pairs (X[j], X[nx/2-j]) after that), y and z in the
// digit-reversed order the in-place transforms leave them in, OTFs tiled per z-conv workgroup plus a small
// "Q" plane for the packed column; nothing outside this module depends on it.  Boxes: every size snapTransformSize
// produces up to 1024 per axis (powers of two from 16, and the multiples of 64 from 192: radices
// 8/4/2 plus 3, 5, 7, 11, 13), x also 2048; anything else uses the rocFFT path in decon.cpp.
#pragma once

#include "mil_common.h"

namespace mil {
namespace fx {

enum XMode {
    kXForward = 0,     // real rows -> spectrum rows
    kXInverse = 1,     // spectrum rows -> real rows (unnormalised)
    kXRatio = 2,       // spectrum -> T' ; T = M / T' ; T -> spectrum         (div3Dgpu fused)
    kXUpdate = 3,      // spectrum -> T' ; E = max(E*T', eps) ; E -> spectrum  (multi3Dgpu + maxvalue3Dgpu fused)
    kXUpdateLast = 4,  // as kXUpdate without the trailing forward transform
};

class ConvPlan {
public:
    static bool supported(Dims box);
    explicit ConvPlan(Dims box);
    Dims box() const { return box_; }
    int xs() const { return xs_; }  // padded spectral x extent (complex columns)
    // complex64 elements of a spectrum / OTF buffer: xs*ny*nz tile elements + the ny*nz "Q" plane an OTF
    // carries for the packed DC/Nyquist column (spectra simply do not use the tail)
    size_t spectrum_count() const { return (size_t)pitch_ * box_.ny * box_.nz + (size_t)box_.ny * box_.nz; }

    // x pass over all ny*nz rows.  S: private spectrum.  Arguments by mode:
    //   kXForward     rin (real in)                      -> S
    //   kXInverse     S                                  -> rout (real out)
    //   kXRatio       S, M = measured image              -> S
    //   kXUpdate(Last) S, rout = estimate E (in and out) -> S (not for Last)
    // reverse: walk the tiles last-to-first.  Alternating the direction between consecutive kernels lets
    // each kernel start on the lines its predecessor wrote last (still in the 256 MB Infinity Cache).
    void x_pass(XMode mode, float2 *S, const float *rin, float *rout, const float *M, float eps, hipStream_t s,
                bool reverse = false) const;
    void y_pass(float2 *S, bool inverse, hipStream_t s, bool reverse = false) const;
    // S <- inverse_z( forward_z(S) * otf ), otf in the tiled private order written by z_forward_to_otf.
    void z_conv(float2 *S, const float2 *otf, hipStream_t s, bool reverse = false) const;
    void z_forward_to_otf(const float2 *S, float2 *otf, hipStream_t s) const;
    // test aid: private OTF order -> natural [nz][ny][nx/2+1] (rocFFT / numpy rfftn order)
    void otf_to_natural(float2 *natural, const float2 *otf, hipStream_t s) const;

    // ---- two-pass path (csrc/fft2p.hip, fft2p_any_*.hip; nx, ny multiples of 64 from 192 / 128, (ny / 64) x nz <= 4096): the
    // tail of the y transform (its stages of length ny / 64) rides in the z kernel and its first two radix-8 stages in the x
    // kernel, so a half view update is  yz_conv | xy_pass  instead of  y | z_conv | y^-1 | x_pass.  Same spectrum positions
    // and OTFs as the classic kernels.
    static bool two_pass_supported(Dims box);
    static int yz_tile_cols(Dims box);  // columns of a yz workgroup's tile (= width of the OTF's column groups), two-pass boxes
    bool two_pass() const { return two_pass_; }
    // partial classic y passes: 0 forward head (stages longer than 8), 1 inverse head, 2 forward tail (the stage of
    // length 8; also refreshes the packed-column copy z_conv reads), 3 inverse tail
    void y_part(float2 *S, int part, hipStream_t s) const;
    // S (y head done) <- y tail, z, * otf, z^-1, y tail^-1
    void yz_conv(float2 *S, const float2 *otf, hipStream_t s) const;
    // OTF tiling.  z_forward_to_otf writes [ky][column group][kz][columns of a group]; the groups are 16 columns wide (what
    // z_conv reads) unless the plan is told that its OTFs feed yz_conv: then a group is as wide as a yz workgroup's tile (8 or
    // 4 columns, 16 for the smallest boxes), so that the workgroup's OTF block is contiguous instead of a half or a quarter of
    // every 128-byte line -- which also lets the loads carry the stream hint without losing the other workgroups' share of the
    // line in L2.  Set before the first OTF is generated; z_conv refuses the narrow tiling, yz_conv the wide one.
    void set_yz_otf_layout(bool on);
    bool yz_otf_layout() const { return yz_otf_; }
    int otf_cols() const { return otf_cols_; }
    // S (y head done) <- y head^-1, x^-1, ratio / update, x, y head     (mode: kXRatio, kXUpdate, kXUpdateLast)
    void xy_pass(XMode mode, float2 *S, float *E, const float *M, float eps, hipStream_t s) const;

private:
    Dims box_;
    int xs_ = 0;     // spectral columns (multiple of 16)
    int pitch_ = 0;  // row pitch of a spectrum in complex elements (xs_ + 16)
    mutable DeviceBuffer<float2> col0_;  // [z][ky] copy of the packed column, forward y pass -> z kernels
    int ablate_ = 0;   // timing experiments only (build with -DMIL_FX_EXPERIMENTS, then MIL_FX_ABLATE); always 0 otherwise
    DeviceBuffer<float2> tw_x_, tw_y_, tw_z_;   // exp(-2*pi*i*k/n), k in [0, n)
    DeviceBuffer<unsigned short> pos_x_;        // position of element n after digit reversal, n in [0, nx/2)
    DeviceBuffer<unsigned short> freq_y_, freq_z_;  // frequency held at scrambled position p
    DeviceBuffer<unsigned short> neg_y_, neg_z_;    // position holding the negated frequency of position p
    void init_two_pass();
    bool two_pass_ = false;
    bool stream_ = false;  // fft2p.hip: non-temporal hints on OTF / image / estimate accesses (large boxes)
    bool yz_otf_ = false;  // OTFs tiled for yz_conv (otf_cols_ columns per group) instead of z_conv (16)
    int otf_cols_ = 16;
    bool xy_any_ = false, yz_any_ = false;  // the box runs the run-time-geometry kernels of fft2p_any_xy.hip / fft2p_any_yz.hip
    DeviceBuffer<unsigned short> pairs_;  // two-pass path: pairs of 8-row blocks holding ky and -ky
    int n_pairs_ = 0;
};

// Host-side tables (also used by the CPU tests through the C ABI).
void digit_reversal_freq(int n, unsigned short *freq_of_pos);  // freq[p]
void digit_reversal_pos(int n, unsigned short *pos_of_elem);   // pos[k], inverse of freq

}  // namespace fx
}  // namespace mil
