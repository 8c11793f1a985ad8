// libapi.h registration / affine-transform entry points (host pointers).
// Mirrors reg3d, reg_3dgpu, atrans3dgpu, checkmatrix of the reference
// (src/api_reg.cpp:58-84, 247-262, 264-652).
#include <chrono>
#include <cstring>

#include "../../include/libapi.h"
#include "reg.h"
#include "trace.h"

using namespace mil;

namespace {

float free_mb() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0.f;
    return (float)free_b / 1048576.0f;
}

struct StreamGuard {
    hipStream_t s = nullptr;
    StreamGuard() { MIL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~StreamGuard() {
        if (s) (void)hipStreamDestroy(s);
    }
};

}  // namespace

extern "C" bool checkmatrix(float *iTmx, long long int sx, long long int sy, long long int sz) {
    return check_matrix(iTmx, sx, sy, sz);
}

extern "C" int atrans3dgpu(float *h_reg, float *iTmx, float *h_img2, unsigned int *imSize1, unsigned int *imSize2,
                           int deviceNum) {
    return guarded("atrans3dgpu", [&]() -> int {
        MIL_HIP(hipSetDevice(deviceNum));
        const Dims out{(long long)imSize1[0], (long long)imSize1[1], (long long)imSize1[2]};
        const Dims in{(long long)imSize2[0], (long long)imSize2[1], (long long)imSize2[2]};
        StreamGuard sg;
        DeviceBuffer<float> d_out((size_t)out.count()), d_in((size_t)in.count());
        MIL_HIP(hipMemcpyAsync(d_in.get(), h_img2, (size_t)in.count() * sizeof(float), hipMemcpyHostToDevice, sg.s));
        k::Affine12 m;
        std::memcpy(m.m, iTmx, sizeof m.m);
        k::affine_resample(d_out.get(), d_in.get(), out, in, m, sg.s);
        MIL_HIP(hipMemcpyAsync(h_reg, d_out.get(), (size_t)out.count() * sizeof(float), hipMemcpyDeviceToHost, sg.s));
        MIL_HIP(hipStreamSynchronize(sg.s));
        return 0;
    });
}

namespace {

// reg3d (reference src/api_reg.cpp:264-607) between its transfers: target and source (already aligned to the target's box)
// in device memory, the registered source left in d_reg.  Shared by the host-pointer entry reg3d and the device-pointer
// entry mil_reg3d.  records[8..10] and [7] are the caller's.
void reg3d_core(float *d_reg, float *iTmx, const float *d_t, const float *d_s, Dims d1, int regChoice, int affMethod, bool flagTmx,
                float FTOL, int itLimit, bool verbose, float *records, hipStream_t s) {
    const int method = (regChoice == 0) ? 0 : affMethod;  // :412
    if (regChoice == 1 || regChoice == 3) {  // phasor: pixel-level translation (:418-446)
        long long sh[3];
        phasor_shift(sh, d_t, d_s, d1, s);
        for (int j = 0; j < 12; ++j) iTmx[j] = 0.f;
        iTmx[0] = iTmx[5] = iTmx[10] = 1.f;
        iTmx[3] = (float)sh[0]; iTmx[7] = (float)sh[1]; iTmx[11] = (float)sh[2];
        if (regChoice == 1) {  // the shifted source IS the result
            k::shift_volume(d_reg, d_s, d1, -sh[0], -sh[1], -sh[2], false, s);
            return;
        }
        flagTmx = true;  // regChoice 3: affine from that translation
    }
    if (regChoice == 4) {  // 2-D MIP shifts first, then affine from that translation (:448-497)
        if (verbose) printf("\t... 2D MIP registration ... \n");
        float shift[3];
        mip_shift_search(shift, d_t, d_s, d1, s);
        for (int j = 0; j < 12; ++j) iTmx[j] = 0.f;
        iTmx[0] = iTmx[5] = iTmx[10] = 1.f;
        iTmx[3] = shift[0]; iTmx[7] = shift[1]; iTmx[11] = shift[2];
        if (verbose) {
            printf("\t... shift translation, X: %2.1f; Y: %2.1f; Z: %2.1f\n", shift[0], shift[1], shift[2]);
            printf("\t... 2D MIP registration completed. \n\t... 3D registration ... \n");
        }
        flagTmx = true;
    }
    reg3d_affine_device(d_reg, iTmx, d_t, d_s, d1, method, flagTmx, FTOL, itLimit, verbose, records, s);
}

int reg3d_check_args(int regChoice, int gpuMemMode) {
    if (gpuMemMode == 0) {  // same answer as the reference, src/api_reg.cpp:390-393
        printf("\n ****CPU registraion function is under developing **** \n");
        return -1;
    }
    if (gpuMemMode < -1 || gpuMemMode > 2) {
        printf("\n****Wrong gpuMemMode setup, no deconvolution performed !!! ****\n");
        return 1;  // :591-593
    }
    if (regChoice < 0 || regChoice > 4) {
        printf("\n*** Wrong registration choice is setup, no registraiton performed !!! **** \n");
        return 1;  // :512-514
    }
    return 0;
}

}  // namespace

extern "C" int reg3d(float *h_reg, float *iTmx, float *h_img1, float *h_img2, unsigned int *imSize1,
                     unsigned int *imSize2, int regChoice, int affMethod, bool flagTmx, float FTOL, int itLimit,
                     int deviceNum, int gpuMemMode, bool verbose, float *records) {
    if (const int bad = reg3d_check_args(regChoice, gpuMemMode)) return bad;
    return guarded("reg3d", [&]() -> int {
        TraceRange trace("libapi::reg3d");
        const auto t0 = std::chrono::steady_clock::now();
        MIL_HIP(hipSetDevice(deviceNum));
        records[8] = free_mb();
        if (verbose) printf("\t... GPU free memory before registration is %.0f MB\n", records[8]);
        const Dims d1{(long long)imSize1[0], (long long)imSize1[1], (long long)imSize1[2]};
        const Dims d2{(long long)imSize2[0], (long long)imSize2[1], (long long)imSize2[2]};
        const size_t n1 = (size_t)d1.count(), n2 = (size_t)d2.count();
        records[0] = 1.f;  // modes 1, 2 and auto all run fully resident (288 GB HBM)

        StreamGuard sg;
        DeviceBuffer<float> d_t(n1), d_s(n1), d_reg(n1);
        if (d1 == d2) {
            MIL_HIP(hipMemcpyAsync(d_s.get(), h_img2, n2 * sizeof(float), hipMemcpyHostToDevice, sg.s));
        } else {  // centred crop / zero-pad of the source to the target box (:401-406)
            DeviceBuffer<float> raw(n2);
            MIL_HIP(hipMemcpyAsync(raw.get(), h_img2, n2 * sizeof(float), hipMemcpyHostToDevice, sg.s));
            k::align_size(d_s.get(), raw.get(), d1, d2, sg.s);
            MIL_HIP(hipStreamSynchronize(sg.s));
        }
        MIL_HIP(hipMemcpyAsync(d_t.get(), h_img1, n1 * sizeof(float), hipMemcpyHostToDevice, sg.s));
        records[9] = free_mb();
        reg3d_core(d_reg.get(), iTmx, d_t.get(), d_s.get(), d1, regChoice, affMethod, flagTmx, FTOL, itLimit, verbose, records, sg.s);
        MIL_HIP(hipMemcpyAsync(h_reg, d_reg.get(), n1 * sizeof(float), hipMemcpyDeviceToHost, sg.s));
        MIL_HIP(hipStreamSynchronize(sg.s));
        records[10] = free_mb();
        records[7] = (float)std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (verbose && regChoice != 1) printf("\t... registration done !!! \n");
        return 0;
    });
}

// reg3d on device-resident volumes (include/mil_device.h): the same stages, the same arithmetic, no transfers.  d_reg,
// d_img1 and d_img2 must not overlap; the volumes are only read.  The call returns when d_reg is complete (the registration
// synchronises `stream` with every cost evaluation anyway).
extern "C" int mil_reg3d(float *d_reg, float *iTmx, const float *d_img1, const float *d_img2, const unsigned int *imSize1,
                         const unsigned int *imSize2, int regChoice, int affMethod, int flagTmx, float FTOL, int itLimit,
                         int verbose, float *records, void *stream) {
    if (const int bad = reg3d_check_args(regChoice, 1)) return bad;
    return guarded("mil_reg3d", [&]() -> int {
        TraceRange trace("mil_reg3d");
        const auto t0 = std::chrono::steady_clock::now();
        hipStream_t s = (hipStream_t)stream;
        records[8] = free_mb();
        if (verbose) printf("\t... GPU free memory before registration is %.0f MB\n", records[8]);
        const Dims d1{(long long)imSize1[0], (long long)imSize1[1], (long long)imSize1[2]};
        const Dims d2{(long long)imSize2[0], (long long)imSize2[1], (long long)imSize2[2]};
        records[0] = 1.f;
        DeviceBuffer<float> aligned;
        const float *d_s = d_img2;
        if (!(d1 == d2)) {  // centred crop / zero-pad of the source to the target box (:401-406)
            aligned.alloc((size_t)d1.count());
            k::align_size(aligned.get(), d_img2, d1, d2, s);
            d_s = aligned.get();
        }
        records[9] = free_mb();
        reg3d_core(d_reg, iTmx, d_img1, d_s, d1, regChoice, affMethod, flagTmx != 0, FTOL, itLimit, verbose != 0, records, s);
        MIL_HIP(hipStreamSynchronize(s));
        aligned.release();
        records[10] = free_mb();
        records[7] = (float)std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (verbose && regChoice != 1) printf("\t... registration done !!! \n");
        return 0;
    });
}

extern "C" int reg_3dgpu(float *h_reg, float *iTmx, float *h_img1, float *h_img2, unsigned int *imSize1,
                         unsigned int *imSize2, int affMethod, int inputTmx, float FTOL, int itLimit,
                         int /*subBgTrigger*/, int deviceNum, float *regRecords) {
    // reference src/api_reg.cpp:609-652: 2-D MIP pre-registration unless an input matrix is given,
    // then a retry with plain affine registration when the matrix fails the sanity check.
    int regChoice = 4;
    bool flagTmx = false;
    if (inputTmx == 1) {
        flagTmx = true;
        regChoice = 2;
    }
    int status = reg3d(h_reg, iTmx, h_img1, h_img2, imSize1, imSize2, regChoice, affMethod, flagTmx, FTOL, itLimit,
                       deviceNum, 1, false, regRecords);
    const bool ok = (status == 0) && checkmatrix(iTmx, imSize1[0], imSize1[1], imSize1[2]);
    if (!ok) {
        status = reg3d(h_reg, iTmx, h_img1, h_img2, imSize1, imSize2, 2, affMethod, flagTmx, FTOL, itLimit, deviceNum,
                       1, false, regRecords);
    }
    return status;
}

extern "C" int reg2d(float *h_reg, float *iTmx, float *h_img1, float *h_img2, unsigned int *imSize1, unsigned int *imSize2,
                     int regChoice, bool flagTmx, float FTOL, int itLimit, int deviceNum, int gpuMemMode, bool verbose,
                     float *records) {
    // reference src/api_reg.cpp:115-244.  iTmx: 6 floats (a0 a1 tx; a3 a4 ty); records: 11 floats.
    if (gpuMemMode == 0) {
        printf("\n **** 2D CPU registration is currently not supported !!! **** \n");
        return regChoice >= 0 && regChoice <= 2 ? 0 : 1;  // the reference prints and returns 0 for choices 0..2 (:170-181)
    }
    if (gpuMemMode != 1) {
        printf("\n****Wrong gpuMemMode setup, no deconvolution performed !!! ****\n");
        return 1;  // :226-228
    }
    if (regChoice < 0 || regChoice > 3) {
        printf("\n ****Wrong registration choice is setup, no registraiton performed !!! **** \n");
        return 1;  // :221-223
    }
    return guarded("reg2d", [&]() -> int {
        const auto t0 = std::chrono::steady_clock::now();
        MIL_HIP(hipSetDevice(deviceNum));
        records[8] = free_mb();
        records[0] = (float)gpuMemMode;
        const int w1 = (int)imSize1[0], h1 = (int)imSize1[1], w2 = (int)imSize2[0], h2 = (int)imSize2[1];
        if (regChoice == 3 && (w1 != w2 || h1 != h2)) {
            printf("\n ****Image size of the 2D images is not matched, processing stop !!! **** \n");
            return 1;  // :208-211
        }
        StreamGuard sg;
        const size_t n1 = (size_t)w1 * h1, n2 = (size_t)w2 * h2;
        DeviceBuffer<float> d1(n1), d2(n2), d_reg(n1);
        MIL_HIP(hipMemcpyAsync(d1.get(), h_img1, n1 * sizeof(float), hipMemcpyHostToDevice, sg.s));
        MIL_HIP(hipMemcpyAsync(d2.get(), h_img2, n2 * sizeof(float), hipMemcpyHostToDevice, sg.s));
        records[9] = free_mb();
        const bool writes_output = !(regChoice == 0 && !flagTmx);  // choice 0 without a matrix does nothing upstream
        reg2d_device(d_reg.get(), iTmx, d1.get(), d2.get(), w1, h1, w2, h2, regChoice, flagTmx, FTOL, itLimit, records, sg.s);
        if (writes_output) MIL_HIP(hipMemcpyAsync(h_reg, d_reg.get(), n1 * sizeof(float), hipMemcpyDeviceToHost, sg.s));
        MIL_HIP(hipStreamSynchronize(sg.s));
        records[10] = free_mb();
        records[7] = (float)std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (verbose) printf("Total time cost for whole processing is %2.3f s\n", records[7]);
        return 0;
    });
}
