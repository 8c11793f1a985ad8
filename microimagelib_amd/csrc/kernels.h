// Internal C++ launch interface of the hand-written gfx950 kernels.
// Every launcher enqueues on `s` and returns without synchronising (except the
// *_sync reductions, which hand a scalar back to the host).
#pragma once

#include "mil_common.h"

namespace mil {
namespace k {

// ---- RL element-wise (kernels.hip) -----------------------------------------
void rl_ratio(float *T, const float *A, size_t n, hipStream_t s);              // T = A / T
void rl_update(float *E, const float *T, size_t n, float eps, hipStream_t s);   // E = max(E*T, eps)
void cmul_inplace(float2 *S, const float2 *OTF, size_t n, hipStream_t s);      // S *= OTF (complex)
void clamp_min(float *out, const float *in, size_t n, float eps, hipStream_t s);
void init_single(float *A, float *E, size_t n, float eps, hipStream_t s);       // A=max(A,eps); E=A
void init_dual(float *A, float *B, float *E, size_t n, float eps, hipStream_t s);
void fill(float *out, float v, size_t n, hipStream_t s);
void scale(float *out, const float *in, float c, size_t n, hipStream_t s);      // out = in * c
void add_value(float *out, const float *in, float c, size_t n, hipStream_t s);  // out = in + c
void u16_to_f32(float *out, const unsigned short *in, size_t n, hipStream_t s);   // (float)v: readtifstack's widening
void f32_to_u16(unsigned short *out, const float *in, size_t n, hipStream_t s);   // (unsigned short)f: writetifstack's C cast

// ---- reductions (float64 accumulation, deterministic two-stage) ------------
enum ReduceOp { kSum = 0, kSumSqShifted = 1 };
struct ReduceWorkspace {
    DeviceBuffer<double> partials;  // kMaxReduceBlocks * 2
    double *host = nullptr;         // pinned, 2 doubles
    ReduceWorkspace();
    ~ReduceWorkspace();
    ReduceWorkspace(const ReduceWorkspace &) = delete;
    ReduceWorkspace &operator=(const ReduceWorkspace &) = delete;
};
// sum(in) or sum((in+c)^2) with the square formed in float32 like the reference's
// addvaluegpu -> multi3Dgpu -> sum3Dgpu chain.  Synchronises `s`.
double reduce_sync(ReduceOp op, const float *in, size_t n, float c, ReduceWorkspace &ws, hipStream_t s);

// ---- geometry (all x-fastest, output-driven, coalesced) --------------------
// out[(i - P/2) mod F] = scale * psf[flip ? P-1-i : i]; rest of the F box zero.  P <= F per axis.
void pad_psf(float *out, const float *psf, Dims F, Dims P, float scale, bool flip, hipStream_t s);
// centred placement with edge replication, offset (F - S)/2.
void pad_stack(float *out, const float *in, Dims F, Dims S, hipStream_t s);
// centred crop, offset (F - S)/2.
void crop_stack(float *out, const float *in, Dims S, Dims F, hipStream_t s);
// centred crop-or-zero-pad, offset trunc((O - I)/2).
void align_size(float *out, const float *in, Dims O, Dims I, hipStream_t s);
// out = in reversed on all three axes.
void flip3(float *out, const float *in, Dims d, hipStream_t s);
// 90-degree rotations about Y (rotbyyaxiskernel), direction +1 / -1; `in` dims given.
void rot_y90(float *out, const float *in, Dims in_dims, int direction, hipStream_t s);
// maximum-intensity projections with accumulator starting at 0 (maxprojectionkernel).
void max_projection(float *out, const float *in, Dims d, int direction, hipStream_t s);

// out[p] = in[p - d]: circular (circshiftgpu) or zero-filled (imshiftgpu), reference include/cukernel.cuh:457-489.
void shift_volume(float *out, const float *in, Dims d, long long dx, long long dy, long long dz, bool wrap, hipStream_t s);
// crop of extent S at offset (ox, oy, oz) of a volume F (cropgpu2, reference src/api_subfunc.cu:1759).
void crop_offset(float *out, const float *in, Dims S, Dims F, long long ox, long long oy, long long oz, hipStream_t s);
// out = conj(s1) * s2 / |conj(s1) * s2| (0 where the product vanishes): conj3Dgpu + multicomplexnorm3Dgpu.
void cross_power_norm(float2 *out, const float2 *s1, const float2 *s2, size_t n, hipStream_t s);
// per (x, y) column: first maximum along z and its index (maxZkernel).
void argmax_columns(float *col_max, int *col_arg, const float *in, Dims d, hipStream_t s);
// sum over i of float32((a[i]+ca) * (b[i]+cb)), accumulated in float64.  Synchronises `s`.
double reduce_dot_sync(const float *a, const float *b, size_t n, float ca, float cb, ReduceWorkspace &ws, hipStream_t s);

// out[p] = in[p - d]: circular (circshiftgpu) or zero-filled (imshiftgpu), reference include/cukernel.cuh:457-489.
void shift_volume(float *out, const float *in, Dims d, long long dx, long long dy, long long dz, bool wrap, hipStream_t s);
// crop of extent S at offset (ox, oy, oz) of a volume F (cropgpu2, reference src/api_subfunc.cu:1759).
void crop_offset(float *out, const float *in, Dims S, Dims F, long long ox, long long oy, long long oz, hipStream_t s);
// out = conj(s1) * s2 / |conj(s1) * s2| (0 where the product vanishes): conj3Dgpu + multicomplexnorm3Dgpu.
void cross_power_norm(float2 *out, const float2 *s1, const float2 *s2, size_t n, hipStream_t s);
// per (x, y) column: first maximum along z and its index (maxZkernel).
void argmax_columns(float *col_max, int *col_arg, const float *in, Dims d, hipStream_t s);
// sum over i of float32((a[i]+ca) * (b[i]+cb)), accumulated in float64.  Synchronises `s`.
double reduce_dot_sync(const float *a, const float *b, size_t n, float ca, float cb, ReduceWorkspace &ws, hipStream_t s);

// ---- affine (affine.hip) ---------------------------------------------------
struct Affine12 {
    float m[12];
};
// Interpolation weights: 0 = exact float32 (default), 8 = rounded to 8 fractional bits (CUDA texture-unit emulation,
// a comparison aid: affine.hip make_tap).  Process-wide; MIL_TEX_WEIGHT_BITS=8 sets it from the environment.  With 8 the
// registration evaluates its cost with the general kernel (the padded-source kernels implement exact weights only).
int texture_weight_bits();
void set_texture_weight_bits(int bits);  // 0 | 8 | anything else: back to the environment's choice
bool corr_fences();
void corr_set_fences(int on);           // 1 / 0; -1: back to MIL_CORR_FENCES
void corr_set_chunk_wgs_per_cu(int n);  // experiments: workgroups per CU of the chunked cost kernels (0: the built-in choice)
// out[x,y,z] = trilinear(src, M*(x,y,z,1)) inside the window, 0 outside.
// strict_window: lower bound 0 < t instead of 0 <= t (the reference's 2-D transform kernel).
void affine_resample(float *out, const float *src, Dims out_dims, Dims src_dims, const Affine12 &m,
                     hipStream_t s, bool strict_window = false);
// pieces of affine_resample (affine.hip, resample_march.hip)
bool small_source(Dims src);  // fetch_linear_small's range: 32-bit offsets, planes below 2^24 voxels, exact float32 weights
bool resample_march_eligible(Dims out, Dims src);
void release_march_ticket(hipStream_t s);  // the stream is about to be destroyed
void affine_resample_march(float *out, const float *src, Dims out_dims, Dims src_dims, const Affine12 &m, hipStream_t s,
                           bool strict_window);
bool resample_mip2_eligible(Dims box, Dims src);
void resample_mip2_z(float *out_plane, const float *src, Dims box, Dims src_dims, const Affine12 &m, hipStream_t s,
                     bool plane_is_zero = false);
// mip3dgpu about y through the transposed volume (csrc/resample_march.hip): true if `m` is a rotation-series matrix about y
// that the wave-uniform kernel can run on transpose_xy(src); `mt` = the matrix for the transposed problem (box and source with
// x and y exchanged), whose plane comes out transposed
bool mip_about_y_by_transpose(const Affine12 &m, Dims box, Dims src_dims, Affine12 &mt);
// out[z][x][y] = in[z][y][x] (dims of `in`)
void transpose_xy(float *out, const float *in, Dims in_dims, hipStream_t s);
// affine_resample into `box` followed by the maximum projection along z, fused: out_plane[box.ny][box.nx] (mip3dgpu's
// inner pair, src/apifunc.cpp:610-613); same bits as the two kernels one after the other
// plane_is_zero: the caller has zeroed the plane on `s` already (a whole stack of them at once: mip_series)
void resample_mip_z(float *out_plane, const float *src, Dims box, Dims src_dims, const Affine12 &m, hipStream_t s,
                    bool plane_is_zero = false);
constexpr int kCorrCounters = 64;  // first-level completion counters of a cost kernel (+ 1 top-level)
struct CorrTail {  // what the last workgroup of a cost kernel needs (corr_finish.h)
    double *partials;   // [2][nblocks]
    int nblocks;
    unsigned *counter;  // [kCorrCounters + 1] workgroups finished so far, by group, and groups finished
    double *host;       // pinned {sum s*s, sum s*t, ticket}
    double ticket;
    int fenced;         // MIL_CORR_FENCES=1: release / acquire fences around the completion counters (diagnostic)
};
struct CorrWorkspace {
    DeviceBuffer<double> partials;
    DeviceBuffer<unsigned> counter;
    double *host = nullptr;  // pinned, coherent: {sum s*s, sum s*t, ticket of the evaluation they belong to}
    double ticket = 0.0;     // evaluations launched with this workspace
    bool counters_zeroed = false;
    CorrWorkspace();
    ~CorrWorkspace();
    CorrWorkspace(const CorrWorkspace &) = delete;
    CorrWorkspace &operator=(const CorrWorkspace &) = delete;
    CorrTail next(size_t nblocks, hipStream_t s);  // sizes the partials, takes a ticket (first call: zeroes the counters on s)
    void wait(hipStream_t s);             // until the evaluation of the last ticket has published its sums
};
// Enqueue the fused interpolate + correlate + reduce (ONE launch); the result lands in ws.host, ws.wait() returns
// once it is there (corr_sums_sync does both).
// float_products: form s*s and s*t in float32 before the float64 sums (the reference's 2-D cost).
void corr_sums_async(const float *target, const float *src, Dims t_dims, Dims s_dims, const Affine12 &m,
                     CorrWorkspace &ws, hipStream_t s, bool float_products = false);
void corr_sums_sync(const float *target, const float *src, Dims t_dims, Dims s_dims, const Affine12 &m,
                    CorrWorkspace &ws, hipStream_t s, double *ss, double *st, bool float_products = false);


// Source volume of the 3-D registration with a one-voxel replicated border -- the texture unit's clamp addressing
// materialised in memory, so the cost kernels need neither index clamps nor edge selects (corr_padded.hip).  Held twice:
// plain, [sz+2][sy+2][pitch >= sx+2] floats, and as row pairs, [sz+2][sy+1][sx+2] float2 = (row y, row y+1), which turns the four
// taps of a source plane into ONE 16-byte gather.  corr_sums_padded_* pick the kernel per matrix; every choice returns
// the same sums, bit for bit, as corr_sums_* on the unpadded volume.
struct PaddedSource {
    DeviceBuffer<float> data;   // plain layout
    DeviceBuffer<float> pairs;  // row-pair layout (absent when it would exceed the kernels' 32-bit offsets)
    bool has_pairs = false;
    Dims dims{0, 0, 0};  // unpadded extents
    long long pitch = 0;  // row pitch of the plain layout in floats: sx + 2 rounded up to a multiple of 4
    static long long plain_pitch(Dims source);
    static bool eligible(Dims target, Dims source);  // all byte offsets fit 32 bits
    static bool row_pairs_fit(Dims source);
    void build(const float *src, Dims dims, hipStream_t s);
};
// test / profiling aid: 0 plain + plane reuse | 1 row pairs | 2 row pairs + plane reuse | 3-5 the same with two voxels
// per lane | -1 (default) per matrix
void corr_padded_force_kernel(int kernel);
// host logic only: the kernel the per-matrix choice takes when both layouts are held (0 / 1 / 2 as above)
int corr_padded_pick_kernel(const Affine12 &m, bool wide = false);
void corr_sums_padded_async(const float *target, const PaddedSource &src, Dims t_dims, const Affine12 &m,
                            CorrWorkspace &ws, hipStream_t s);
void corr_sums_padded_sync(const float *target, const PaddedSource &src, Dims t_dims, const Affine12 &m,
                           CorrWorkspace &ws, hipStream_t s, double *ss, double *st);

}  // namespace k
}  // namespace mil
