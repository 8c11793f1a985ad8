// Two-pass path, boxes outside the powers of two (csrc/fft2p_any_xy.hip, csrc/fft2p_any_yz.hip): the interface
// ConvPlan (csrc/fft2p.hip) calls.  snapTransformSize (reference src/api_subfunc.cu:57-87) rounds every extent above 128
// that is not a power of two to a multiple of 64, so real stacks land on 320, 384, 448, 576, 640 ...; these files hold
// the same kernels as fft2p.hip built with the y extent (xy kernel) or the row geometry (yz kernel) as run-time
// arguments, one instantiation per NH = nx / 2 and per (ny, nz).
#pragma once

#include "fftconv.h"

namespace mil {
namespace fx {
namespace any {

struct Box {
    int nx, ny, nz;
    int xs, pitch;  // spectrum columns (nx / 2) and row pitch (xs + 16)
};

// ---- fft2p_any_xy.hip: y head^-1 . x^-1 . op . x . y head for nx = 192 .. 1024 (multiples of 64), ny = 128 .. 1024
bool xy_covers(const Box &b);
void xy_fill_tables(const float2 *tw_y_dev, int ny);  // this translation unit's constant twiddle table
void xy_pass(const Box &b, int mode, float2 *S, float *E, const float *M, const float2 *tw_x, float eps, bool stream,
             hipStream_t s);

// ---- fft2p_any_yz.hip: y tail . z . OTF . z^-1 . y tail^-1 on LDS tiles for (ny / 64) x nz <= 4096, nz a multiple of 64;
// the partial classic y passes (head / tail) for the 64 k extents
bool yz_covers(const Box &b);
int yz_cols(const Box &b);  // columns of a workgroup's tile = width of the OTF's column groups yz_conv expects
void yz_fill_tables(const float2 *tw_y_dev, int ny);
void yz_conv(const Box &b, float2 *S, const float2 *otf, const unsigned short *neg_y, const unsigned short *neg_z,
             const unsigned short *pairs, int n_pairs, const float2 *tw_z, hipStream_t s);
bool y_part_covers(int ny);
void y_part(const Box &b, float2 *S, const float2 *tw_y, float2 *col0, int part, hipStream_t s);

}  // namespace any
}  // namespace fx
}  // namespace mil
