// TEST INFRASTRUCTURE ONLY -- our own extern "C" wrappers around the reference's plain-C++ host helpers.
//
// oracle/Makefile (target `subfunc`) pipes line ranges of /root/reference/src/api_subfunc.cu -- the helpers that need
// nothing but libc: snapTransformSize, the element-wise CPU loops, the matrix helpers, itrilerp2 /
// affinetransformcpu / corrfunccpu2 and the CPU geometry helpers -- into g++ and appends THIS file, so the reference
// code is compiled where it lies (no copy in the repository, no stand-in header; `fComplex` comes from the reference's
// own include/apifunc_internal.h:13-21 by line range as well).  The result, oracle/_ref/libsubfunc_ref.so, pins
// oracle/decon_oracle.py's and oracle/reg_oracle.py's geometry / matrix / trilinear pieces to reference code
// (tests/test_oracle_pinned.py).  Templates and C++-linkage functions cannot be dlsym'ed, hence the wrappers.
extern "C" {
int ref_snapTransformSize(int n) { return snapTransformSize(n); }
double ref_sumcpu(float *a, size_t n) { return sumcpu<float>(a, n); }
void ref_addvaluecpu(float *o, float *a, float b, size_t n) { addvaluecpu<float>(o, a, b, n); }
void ref_multicpu(float *o, float *a, float *b, size_t n) { multicpu<float>(o, a, b, n); }
void ref_divcpu(float *o, float *a, float *b, size_t n) { divcpu<float>(o, a, b, n); }
void ref_maxvaluecpu(float *o, float *a, float b, size_t n) { maxvaluecpu<float>(o, a, b, n); }
void ref_multicomplexcpu(float *o, float *a, float *b, size_t n) {
    multicomplexcpu((fComplex *)o, (fComplex *)a, (fComplex *)b, n);
}
void ref_p2matrix(float *m, float *x) { p2matrix(m, x); }
void ref_matrix2p(float *m, float *x) { matrix2p(m, x); }
void ref_matrixmultiply(float *m, float *m1, float *m2) { matrixmultiply(m, m1, m2); }
void ref_rot2matrix(float *out, float theta, long long sx, long long sy, long long sz, int axis) {
    rot2matrix(out, theta, sx, sy, sz, axis);
}
void ref_dof9tomatrix(float *out, float *p, int dof) { dof9tomatrix(out, p, dof); }
float ref_itrilerp2(float dx1, float dx2, float dy1, float dy2, float dz1, float dz2, float *q) {
    return itrilerp2(dx1, dx2, dy1, dy2, dz1, dz2, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
}
// h_s / h_t are [i][j][k] with i slowest (the reference's CPU layout); aff maps (i, j, k) -> source (i, j, k)
void ref_affinetransformcpu(float *out, float *src, float *aff, int sx, int sy, int sz, int sx2, int sy2, int sz2) {
    affinetransformcpu(out, src, aff, sx, sy, sz, sx2, sy2, sz2);
}
double ref_corrfunccpu2(float *target, float *src, float *aff, int sx, int sy, int sz, int sx2, int sy2, int sz2) {
    return corrfunccpu2(target, src, aff, sx, sy, sz, sx2, sy2, sz2);
}
void ref_flipcpu(float *o, float *i, long long sx, long long sy, long long sz) { flipcpu<float>(o, i, sx, sy, sz); }
void ref_padPSFcpu(float *o, float *i, long long sx, long long sy, long long sz, long long sx2, long long sy2,
                   long long sz2) {
    padPSFcpu<float>(o, i, sx, sy, sz, sx2, sy2, sz2);
}
void ref_padstackcpu(float *o, float *i, long long sx, long long sy, long long sz, long long sx2, long long sy2,
                     long long sz2) {
    padstackcpu<float>(o, i, sx, sy, sz, sx2, sy2, sz2);
}
void ref_cropcpu(float *o, float *i, long long sx, long long sy, long long sz, long long sx2, long long sy2,
                 long long sz2) {
    cropcpu<float>(o, i, sx, sy, sz, sx2, sy2, sz2);
}
void ref_alignsize3Dcpu(float *o, float *i, long long sx, long long sy, long long sz, long long sx2, long long sy2,
                        long long sz2) {
    alignsize3Dcpu<float>(o, i, sx, sy, sz, sx2, sy2, sz2);
}
}
