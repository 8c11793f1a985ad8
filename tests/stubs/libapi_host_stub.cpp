// TEST STUB (not product): host-only stand-ins for the compute entry points of include/libapi.h that
// microimagelib_amd/cli/spim_fusion_batch_main.cpp calls, so that the tool's multi-worker orchestration (round-robin
// dealing, the regMode-2 token hand-off, ordered log, read-ahead / write-behind) runs under ThreadSanitizer on a machine
// without a GPU.  The file functions are the real ones (csrc/tiff_io.cpp is compiled next to this file).  Every function
// is a cheap deterministic map of its inputs -- NOT the algorithms -- so a run with N workers must reproduce the
// one-worker files byte for byte; a per-device sleep shuffles the order in which workers finish.
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include "../../include/libapi.h"

namespace {
size_t count3(const unsigned int *s) { return (size_t)s[0] * s[1] * s[2]; }
double checksum(const float *v, size_t n) {
    double s = 0;
    for (size_t i = 0; i < n; i += 7) s += v[i] * (double)((i % 13) + 1);
    return s;
}
void jitter(int device) { usleep((useconds_t)(1000 * (1 + (device * 7) % 5))); }
}  // namespace

extern "C" void queryDevice() {}

extern "C" bool checkmatrix(float *m, long long, long long, long long) { return m[0] > 0.5f && m[0] < 1.4f; }

extern "C" int reg3d(float *h_reg, float *iTmx, float *h_img1, float *h_img2, unsigned int *imSize1, unsigned int *, int regChoice,
                     int, bool inputTmx, float, int, int deviceNum, int, bool, float *records) {
    jitter(deviceNum);
    {   // MIL_STUB_FAIL_REG_CALL=k: the k-th registration call of the process fails like a library error (nothing written)
        static std::atomic<int> calls{0};
        const char *fail = getenv("MIL_STUB_FAIL_REG_CALL");
        if (fail && ++calls == atoi(fail)) return -1;
    }
    const size_t n = count3(imSize1);
    float start[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    if (inputTmx) std::memcpy(start, iTmx, sizeof start);
    if (regChoice != 0) {  // "registration": the result depends on both volumes AND on the starting matrix
        const double c = std::fmod(checksum(h_img1, n) + 3.0 * checksum(h_img2, n), 1000.0) / 1000.0;
        for (int j = 0; j < 12; ++j) start[j] = (float)(0.9 * start[j] + 0.1 * (j % 5 == 0 ? 1.0 : 0.0) + 1e-3 * c * (j + 1));
    }
    std::memcpy(iTmx, start, sizeof start);
    for (size_t i = 0; i < n; ++i) h_reg[i] = h_img2[i] + iTmx[3] + 2.f * iTmx[7] + 3.f * iTmx[11];
    records[1] = 0.3f;
    records[3] = 0.8f;
    records[7] = 0.f;
    return 0;
}

extern "C" int decon_dualview(float *h_decon, float *h_img1, float *h_img2, unsigned int *imSize, float *, float *, unsigned int *,
                              bool, int itNumForDecon, int deviceNum, int, bool, float *deconRecords, bool, float *, float *) {
    jitter(deviceNum + 3);
    const size_t n = count3(imSize);
    for (size_t i = 0; i < n; ++i) h_decon[i] = 0.5f * (h_img1[i] + h_img2[i]) + (float)itNumForDecon;
    for (int i = 0; i < 10; ++i) deconRecords[i] = 0.f;
    return 0;
}

extern "C" int imresize3d(float *o, float *in, long long sx1, long long sy1, long long sz1, long long sx2, long long sy2,
                          long long sz2, int) {
    for (long long z = 0; z < sz1; ++z)
        for (long long y = 0; y < sy1; ++y)
            for (long long x = 0; x < sx1; ++x)
                o[x + sx1 * (y + sy1 * z)] = in[(x * sx2 / sx1) + sx2 * ((y * sy2 / sy1) + sy2 * (z * sz2 / sz1))];
    return 0;
}

extern "C" int imoperation3D(float *o, unsigned int *sizeOut, float *in, unsigned int *sizeIn, int, int) {
    sizeOut[0] = sizeIn[2]; sizeOut[1] = sizeIn[1]; sizeOut[2] = sizeIn[0];
    for (unsigned z = 0; z < sizeIn[2]; ++z)
        for (unsigned y = 0; y < sizeIn[1]; ++y)
            for (unsigned x = 0; x < sizeIn[0]; ++x)
                o[z + sizeOut[0] * (y + sizeOut[1] * x)] = in[x + sizeIn[0] * (y + sizeIn[1] * z)];
    return 0;
}

extern "C" int mp2dgpu(float *h_MP, unsigned int *sizeMP, float *h_img, unsigned int *s, bool flagZ, bool, bool) {
    sizeMP[0] = s[0]; sizeMP[1] = s[1]; sizeMP[2] = s[1]; sizeMP[3] = s[2]; sizeMP[4] = s[2]; sizeMP[5] = s[0];
    if (flagZ)
        for (unsigned y = 0; y < s[1]; ++y)
            for (unsigned x = 0; x < s[0]; ++x) {
                float m = 0.f;
                for (unsigned z = 0; z < s[2]; ++z) m = std::fmax(m, h_img[x + s[0] * (y + s[1] * z)]);
                h_MP[x + s[0] * y] = m;
            }
    return 0;
}

extern "C" int mip3dgpu(float *, unsigned int *sizeMP, float *, unsigned int *s, int, long long projectNum) {
    sizeMP[0] = s[0]; sizeMP[1] = s[1]; sizeMP[2] = (unsigned)projectNum;
    return 0;
}

// ---- include/mil_device.h stage entries the tool's device-resident path calls: "device memory" is host memory here, every
// stage is the stub of its libapi.h twin above, so the device path must write the host path's files byte for byte.
#include "../../include/mil_device.h"

namespace {
thread_local int t_device = 0;
}

extern "C" int mil_set_device(int device) { t_device = device; return 0; }
// STUB_DEVICE_FREE_BYTES: what the "device" reports free (default: plenty) -- the tool's choice of one or two workers
extern "C" int mil_dev_mem_info(int, size_t *free_bytes, size_t *total_bytes) {
    const char *e = getenv("STUB_DEVICE_FREE_BYTES");
    const size_t f = e ? (size_t)atoll(e) : (size_t)64 << 30;
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = (size_t)64 << 30;
    return 0;
}
extern "C" int mil_dev_alloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? 0 : -1; }
extern "C" void mil_dev_free(void *p) { free(p); }
extern "C" int mil_host_alloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? 0 : -1; }
extern "C" void mil_host_free(void *p) { free(p); }
extern "C" int mil_stream_create(void **s) { *s = malloc(1); return 0; }
extern "C" void mil_stream_destroy(void *s) { free(s); }
extern "C" int mil_stream_sync(void *) { return 0; }
extern "C" int mil_upload(void *d, const void *h, size_t bytes, void *) { std::memcpy(d, h, bytes); return 0; }
extern "C" int mil_download(void *h, const void *d, size_t bytes, void *) { std::memcpy(h, d, bytes); return 0; }
extern "C" int mil_u16_to_f32(float *o, const unsigned short *in, size_t n, void *) {
    for (size_t i = 0; i < n; ++i) o[i] = (float)in[i];
    return 0;
}
extern "C" int mil_f32_to_u16(unsigned short *o, const float *in, size_t n, void *) {
    for (size_t i = 0; i < n; ++i) o[i] = (unsigned short)(long long)in[i];
    return 0;
}
extern "C" int mil_imresize3d(float *o, const float *in, long long sx, long long sy, long long sz, long long sx2, long long sy2,
                              long long sz2, void *) {
    return imresize3d(o, const_cast<float *>(in), sx, sy, sz, sx2, sy2, sz2, t_device);
}
extern "C" int mil_rot_y90(float *o, const float *in, int nx, int ny, int nz, int direction, void *) {
    unsigned int so[3], si[3] = {(unsigned)nx, (unsigned)ny, (unsigned)nz};
    return imoperation3D(o, so, const_cast<float *>(in), si, direction == 1 ? 1 : 2, t_device);
}
extern "C" int mil_reg3d(float *d_reg, float *iTmx, const float *a, const float *b, const unsigned int *s1, const unsigned int *s2,
                         int regChoice, int affMethod, int flagTmx, float ftol, int itLimit, int verbose, float *records, void *) {
    return reg3d(d_reg, iTmx, const_cast<float *>(a), const_cast<float *>(b), const_cast<unsigned int *>(s1),
                 const_cast<unsigned int *>(s2), regChoice, affMethod, flagTmx != 0, ftol, itLimit, t_device, -1, verbose != 0, records);
}
extern "C" int mil_decon_dualview(float *d, const float *a, const float *b, const unsigned int *size, const float *p1, const float *p2,
                                  const unsigned int *psize, int constInit, int iters, int verbose, float *rec, int unmatch,
                                  const float *bp1, const float *bp2, void *) {
    return decon_dualview(d, const_cast<float *>(a), const_cast<float *>(b), const_cast<unsigned int *>(size), const_cast<float *>(p1),
                          const_cast<float *>(p2), const_cast<unsigned int *>(psize), constInit != 0, iters, t_device, -1, verbose != 0, rec,
                          unmatch != 0, const_cast<float *>(bp1), const_cast<float *>(bp2));
}
extern "C" int mil_max_projection(float *o, const float *in, int nx, int ny, int nz, int direction, void *) {
    if (direction != 1) return 0;  // the host stub fills the Z projection only
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            float m = 0.f;
            for (int z = 0; z < nz; ++z) m = std::fmax(m, in[x + (size_t)nx * (y + (size_t)ny * z)]);
            o[x + (size_t)nx * y] = m;
        }
    return 0;
}
extern "C" int mil_mip3d(float *o, const float *, int nx, int ny, int, int, long long projectNum, int *size3, void *) {
    size3[0] = nx; size3[1] = ny; size3[2] = (int)projectNum;
    if (o) std::memset(o, 0, sizeof(float) * (size_t)nx * ny * projectNum);
    return 0;
}
