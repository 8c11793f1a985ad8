// TEST INFRASTRUCTURE ONLY -- extern "C" wrapper around the reference's checkmatrix (src/api_reg.cpp:246-262), which
// oracle/Makefile (target `regcheck`) pipes into g++ by line range in front of this file: the reference code is compiled
// where it lies, nothing of it is written into the repository.  The function has C++ linkage there, hence the wrapper.
extern "C" int ref_checkmatrix(float *m, long long sx, long long sy, long long sz) { return checkmatrix(m, sx, sy, sz) ? 1 : 0; }
