// 3-D affine registration on device-resident volumes: parameterisations, the
// correlation cost driven by Powell, and the final resample.
#pragma once

#include "kernels.h"

namespace mil {

// 3x4 row-major affine helpers (0-based parameter vectors).
void p2matrix(float *m, const float *p12);                         // reference src/api_subfunc.cu:557
void matrix2p(float *p12, const float *m);                         // :569
void matrixmultiply(float *m, const float *m1, const float *m2);   // :584
void dof9tomatrix(float *m, const float *p9, int dof_num);         // :715
void rot2matrix(float *m, float theta, long long sx, long long sy, long long sz, int rot_axis);  // :626
bool check_matrix(const float *m, long long sx, long long sy, long long sz);                     // api_reg.cpp:247

struct RegOptions {
    // false (default): hand back / resample with the matrix of the LAST cost evaluation, as the
    // reference does through its global affCoef.  true: recompute from the
    // optimiser's final parameter vector.
    bool matrix_from_optimum = false;
};

// reg3d_affine1, reference src/api_subfunc.cu:2733-2994.  d_img1 = target, d_img2 = source,
// both `dims`; d_reg receives the resampled source.  records: >= 8 floats ([1..7] written).
// Throws Error; returns 0.
// Cost-kernel launches of this thread's last reg3d_affine_device optimisation; records[5] counts cost EVALUATIONS as the
// reference does, a few per cent of which repeat a point just visited and are answered without a launch.
int reg3d_last_cost_launches();
// Test / audit aid: record what this thread's next optimisations evaluate.  capacity > 0 starts a fresh trace of up to that
// many evaluations (13 floats each: the 12 matrix floats of the reference's global affCoef at that evaluation, then the
// cost returned to Powell), capacity 0 stops tracing.  reg3d_trace_read copies entries [0, max) and returns how many
// evaluations were seen (which may exceed the capacity; the surplus is counted, not stored).
void reg3d_trace_begin(int capacity);
int reg3d_trace_read(float *out13, int max_entries);
int reg3d_affine_device(float *d_reg, float *iTmx, const float *d_img1, const float *d_img2, Dims dims,
                        int aff_method, bool flag_tmx, float ftol, int it_limit, bool verbose, float *records,
                        hipStream_t stream, const RegOptions &opt = RegOptions());

// 2-D MIP pre-registration of regChoice 4, reference src/api_reg.cpp:448-497 with reg2d_shiftalign1 /
// reg2d_shiftalignX1 (src/api_subfunc.cu:1860-2120): exhaustive 60 x 60 search of the XY shift on the z projections,
// then 60 steps of the z shift on the y projections with the x shift held.  shift = {x, y, z} in voxels.
void mip_shift_search(float shift[3], const float *d_img1, const float *d_img2, Dims dims, hipStream_t stream);

// reg2d on device-resident images (w x h, x fastest): regChoice 0 apply aff6 | 1 translation grid search |
// 2 6-parameter Powell | 3 phasor shift.  aff6 = (a0 a1 tx; a3 a4 ty), target pixel -> source coordinate, in/out.
// records: >= 8 floats ([1], [3], [5], [6] written where the reference writes them).
int reg2d_device(float *d_reg, float *aff6, const float *d_img1, const float *d_img2, int w1, int h1, int w2, int h2,
                 int reg_choice, bool flag_tmx, float ftol, int it_limit, float *records, hipStream_t stream);

// 3-D phasor pre-registration of regChoice 1 / 3, reference reg3d_phasor1 (src/api_subfunc.cu:2465-2590): peak of the
// inverse transform of the normalised cross-power spectrum, with the reference's wrap-around disambiguation by the
// ZNCC of the overlapping crops.  shift = {x, y, z}: source(p) ~ target(p - shift).
void phasor_shift(long long shift[3], const float *d_img1, const float *d_img2, Dims dims, hipStream_t stream);

}  // namespace mil
