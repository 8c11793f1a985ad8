// libapi.h helpers around the hot path: device query, maximum-intensity projections,
// resize / rotate / size alignment.  Reference: src/apifunc.cpp:328-644.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/libapi.h"
#include "reg.h"

using namespace mil;

namespace {

struct StreamGuard {
    hipStream_t s = nullptr;
    StreamGuard() { MIL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~StreamGuard() {
        if (s) (void)hipStreamDestroy(s);
    }
};

}  // namespace

extern "C" void queryDevice() {  // reference src/apifunc.cpp:328-394
    printf(" \n ===========================================\n");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        printf("There are no available device(s) that support HIP\n");
        return;
    }
    printf("Detected %d HIP capable device(s)\n", count);
    for (int dev = 0; dev < count; ++dev) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess) continue;
        printf("\nDevice %d: \"%s\" (%s)\n", dev, p.name, p.gcnArchName);
        printf("  Total amount of global memory:                 %.0f MBytes\n", p.totalGlobalMem / 1048576.0);
        printf("  Compute units:                                 %d\n", p.multiProcessorCount);
        printf("  GPU max clock rate:                            %.0f MHz\n", p.clockRate * 1e-3);
        printf("  Memory bus width:                              %d-bit\n", p.memoryBusWidth);
        printf("  L2 cache size:                                 %d bytes\n", p.l2CacheSize);
        printf("  Shared memory (LDS) per workgroup:             %zu bytes\n", p.sharedMemPerBlock);
        printf("  Wavefront size:                                %d\n", p.warpSize);
        printf("  Maximum threads per workgroup:                 %d\n", p.maxThreadsPerBlock);
    }
    printf(" ===========================================\n\n");
}

extern "C" int alignsize3d(float *h_odata, float *h_idata, long long int sx, long long int sy, long long int sz,
                           long long int sx2, long long int sy2, long long int sz2, int gpuMemMode) {
    if (gpuMemMode < 0 || gpuMemMode > 2) {
        printf("\n****Wrong gpuMemMode setup, processing stopped !!! ****\n");
        return 1;  // reference src/apifunc.cpp:422-424
    }
    return guarded("alignsize3d", [&]() -> int {
        // The reference indexes [x*sy*sz + y*sz + z] (its sz is the contiguous axis).
        const Dims out{sz, sy, sx}, in{sz2, sy2, sx2};
        StreamGuard sg;
        DeviceBuffer<float> d_o((size_t)out.count()), d_i((size_t)in.count());
        MIL_HIP(hipMemcpyAsync(d_i.get(), h_idata, (size_t)in.count() * sizeof(float), hipMemcpyHostToDevice, sg.s));
        k::align_size(d_o.get(), d_i.get(), out, in, sg.s);
        MIL_HIP(hipMemcpyAsync(h_odata, d_o.get(), (size_t)out.count() * sizeof(float), hipMemcpyDeviceToHost, sg.s));
        MIL_HIP(hipStreamSynchronize(sg.s));
        return 0;
    });
}

extern "C" int imresize3d(float *h_odata, float *h_idata, long long int sx, long long int sy, long long int sz,
                          long long int sx2, long long int sy2, long long int sz2, int deviceNum) {
    // diagonal scaling matrix through the affine resampler (reference src/apifunc.cpp:429-446)
    float m[12] = {0};
    m[0] = float(sx2) / float(sx);
    m[5] = float(sy2) / float(sy);
    m[10] = float(sz2) / float(sz);
    unsigned int out[3] = {(unsigned)sx, (unsigned)sy, (unsigned)sz}, in[3] = {(unsigned)sx2, (unsigned)sy2, (unsigned)sz2};
    (void)atrans3dgpu(h_odata, m, h_idata, out, in, deviceNum);
    return 0;
}

extern "C" int imoperation3D(float *h_odata, unsigned int *sizeOut, float *h_idata, unsigned int *sizeIn, int opChoice,
                             int deviceNum) {
    if (opChoice == 0) return 0;
    if (opChoice != 1 && opChoice != 2) {
        printf("\n*** Wrong operation choice !!! **** \n");
        return 1;
    }
    return guarded("imoperation3D", [&]() -> int {
        MIL_HIP(hipSetDevice(deviceNum));
        const Dims in{(long long)sizeIn[0], (long long)sizeIn[1], (long long)sizeIn[2]};
        const size_t n = (size_t)in.count();
        StreamGuard sg;
        DeviceBuffer<float> d_o(n), d_i(n);
        MIL_HIP(hipMemcpyAsync(d_i.get(), h_idata, n * sizeof(float), hipMemcpyHostToDevice, sg.s));
        k::rot_y90(d_o.get(), d_i.get(), in, opChoice == 1 ? 1 : -1, sg.s);
        MIL_HIP(hipMemcpyAsync(h_odata, d_o.get(), n * sizeof(float), hipMemcpyDeviceToHost, sg.s));
        MIL_HIP(hipStreamSynchronize(sg.s));
        sizeOut[0] = sizeIn[2];
        sizeOut[1] = sizeIn[1];
        sizeOut[2] = sizeIn[0];
        return 0;
    });
}

extern "C" int mp2dgpu(float *h_MP, unsigned int *sizeMP, float *h_img, unsigned int *sizeImg, bool flagZProj,
                       bool flagXProj, bool flagYProj) {
    (void)flagYProj;  // upstream gates the Y projection on flagZProj (src/apifunc.cpp:498); kept for parity
    return guarded("mp2dgpu", [&]() -> int {
        const Dims d{(long long)sizeImg[0], (long long)sizeImg[1], (long long)sizeImg[2]};
        const size_t sx = d.nx, sy = d.ny, sz = d.nz;
        const size_t n_mp = sx * sy + sy * sz + sz * sx;
        StreamGuard sg;
        DeviceBuffer<float> d_img((size_t)d.count()), d_mp(n_mp);
        MIL_HIP(hipMemsetAsync(d_mp.get(), 0, n_mp * sizeof(float), sg.s));
        MIL_HIP(hipMemcpyAsync(d_img.get(), h_img, (size_t)d.count() * sizeof(float), hipMemcpyHostToDevice, sg.s));
        if (flagZProj) k::max_projection(d_mp.get(), d_img.get(), d, 1, sg.s);
        if (flagXProj) k::max_projection(d_mp.get() + sx * sy, d_img.get(), d, 3, sg.s);
        if (flagZProj) k::max_projection(d_mp.get() + sx * sy + sy * sz, d_img.get(), d, 2, sg.s);
        sizeMP[0] = sx; sizeMP[1] = sy; sizeMP[2] = sy;
        sizeMP[3] = sz; sizeMP[4] = sz; sizeMP[5] = sx;
        MIL_HIP(hipMemcpyAsync(h_MP, d_mp.get(), n_mp * sizeof(float), hipMemcpyDeviceToHost, sg.s));
        MIL_HIP(hipStreamSynchronize(sg.s));
        return 0;
    });
}

namespace {

// One rotation series: resample into the rotated box and project along z -- one fused pass per angle (k::resample_mip_z;
// MIL_MIP3D_FUSED=0 runs the reference's two kernels, affineTransform then maxprojection, for comparison) -- and keep the
// plane: in device memory (d_out, one plane per angle) or, for the host-pointer entry points, copied out plane by plane like
// the reference (src/apifunc.cpp:610-618).
void mip_series(float *h_out, float *d_out, const float *d_img, Dims img, Dims rot_box, int axis, long long project_num,
                float project_step, hipStream_t s) {
    const size_t plane = (size_t)rot_box.nx * rot_box.ny;
    const char *e = getenv("MIL_MIP3D_FUSED");
    const bool fused = !(e && e[0] == '0');
    // The planes of a series form ONE stack -- the caller's (device-pointer form) or a scratch one that goes to the host in a
    // single copy -- zeroed by ONE memset: a memset per angle (36 fill kernels of ~5 us between kernels of ~65 us) was 4 - 7 % of
    // a series (round 6, profiles/r06/mip3d_grid.txt).
    DeviceBuffer<float> d_rot(fused ? 0 : (size_t)rot_box.count()), d_stack(d_out ? 0 : plane * (size_t)project_num);
    float *stack = d_out ? d_out : d_stack.get();
    std::vector<k::Affine12> mats((size_t)project_num), mats_t((size_t)project_num);
    // rotations about y: the wave-uniform kernel on the transposed volume (k::mip_about_y_by_transpose): one transpose of the
    // volume per series, the planes projected into a transposed stack, ONE transpose of that stack at the end
    const Dims img_t{img.ny, img.nx, img.nz}, box_t{rot_box.ny, rot_box.nx, rot_box.nz};
    std::vector<char> transposed((size_t)project_num, 0);  // per angle: the matrix has the form the transposed problem needs
    bool any_transposed = false;
    for (long long i = 0; i < project_num; ++i) {
        rot2matrix(mats[(size_t)i].m, project_step * (float)(int)i, img.nx, img.ny, img.nz, axis);
        if (fused && axis == 2 && k::mip_about_y_by_transpose(mats[(size_t)i], rot_box, img, mats_t[(size_t)i]))
            transposed[(size_t)i] = 1, any_transposed = true;
    }
    DeviceBuffer<float> d_img_t, d_stack_t;
    if (fused) {
        float *target = stack;
        if (any_transposed) {
            d_img_t.alloc((size_t)img.count());
            d_stack_t.alloc(plane * (size_t)project_num);
            k::transpose_xy(d_img_t.get(), d_img, img, s);
            target = d_stack_t.get();
        }
        MIL_HIP(hipMemsetAsync(target, 0, plane * (size_t)project_num * sizeof(float), s));
        for (long long i = 0; i < project_num; ++i) {
            if (transposed[(size_t)i]) k::resample_mip_z(target + plane * (size_t)i, d_img_t.get(), box_t, img_t, mats_t[(size_t)i], s, true);
            else if (!any_transposed) k::resample_mip_z(target + plane * (size_t)i, d_img, rot_box, img, mats[(size_t)i], s, true);
        }
        if (any_transposed) {
            // the planes of the other angles are zero in the transposed stack, so they arrive zeroed: projected in place now
            k::transpose_xy(stack, target, Dims{box_t.nx, box_t.ny, project_num}, s);
            for (long long i = 0; i < project_num; ++i)
                if (!transposed[(size_t)i]) k::resample_mip_z(stack + plane * (size_t)i, d_img, rot_box, img, mats[(size_t)i], s, true);
        }
    } else {
        for (long long i = 0; i < project_num; ++i) {
            k::affine_resample(d_rot.get(), d_img, rot_box, img, mats[(size_t)i], s);
            k::max_projection(stack + plane * (size_t)i, d_rot.get(), rot_box, 1, s);
        }
    }
    if (h_out) MIL_HIP(hipMemcpyAsync(h_out, stack, plane * (size_t)project_num * sizeof(float), hipMemcpyDeviceToHost, s));
    MIL_HIP(hipStreamSynchronize(s));
    // the scratch buffers were only ever used on `s`: hand them back without a device-wide fence (another worker thread's
    // registration may be queued on its own stream of this device)
    d_rot.release_after(s);
    d_stack.release_after(s);
    d_img_t.release_after(s);
    d_stack_t.release_after(s);
}

// rotated box of mip3dgpu (src/apifunc.cpp:582-590): the rotation axis keeps its extent, the other two become the diagonal
Dims mip3d_box(Dims d, int axis) {
    if (axis == 1) {
        const long long rot = (long long)std::round(std::sqrt((double)(d.ny * d.ny + d.nz * d.nz)));
        return Dims{d.nx, rot, rot};
    }
    const long long rot = (long long)std::round(std::sqrt((double)(d.nx * d.nx + d.nz * d.nz)));
    return Dims{rot, d.ny, rot};
}

}  // namespace

extern "C" int mip3dgpu(float *h_MP, unsigned int *sizeMP, float *h_img, unsigned int *sizeImg, int rAxis,
                        long long int projectNum) {
    if (rAxis != 1 && rAxis != 2) return -1;  // reference src/apifunc.cpp:590-591
    return guarded("mip3dgpu", [&]() -> int {
        const Dims d{(long long)sizeImg[0], (long long)sizeImg[1], (long long)sizeImg[2]};
        const Dims box = mip3d_box(d, rAxis);
        const float step = (float)(3.14159 * 2 / (float)projectNum);  // :597
        StreamGuard sg;
        DeviceBuffer<float> d_img((size_t)d.count());
        MIL_HIP(hipMemcpyAsync(d_img.get(), h_img, (size_t)d.count() * sizeof(float), hipMemcpyHostToDevice, sg.s));
        mip_series(h_MP, nullptr, d_img.get(), d, box, rAxis, projectNum, step, sg.s);
        sizeMP[0] = (unsigned)box.nx; sizeMP[1] = (unsigned)box.ny; sizeMP[2] = (unsigned)projectNum;
        return 0;
    });
}

// Device-pointer form (include/mil_device.h): the stack stays in HBM -- what the process-per-GPU batch driver gathers over
// RCCL (microimagelib_amd/batch.py).  Same arithmetic, same planes as mip3dgpu.
extern "C" int mil_mip3d(float *d_out, const float *d_img, int nx, int ny, int nz, int rAxis, long long projectNum,
                         int *size3, void *stream) {
    if ((rAxis != 1 && rAxis != 2) || projectNum < 1) return -1;
    return guarded("mil_mip3d", [&]() -> int {
        const Dims d{(long long)nx, (long long)ny, (long long)nz};
        const Dims box = mip3d_box(d, rAxis);
        if (size3) { size3[0] = (int)box.nx; size3[1] = (int)box.ny; size3[2] = (int)projectNum; }
        if (!d_out) return 0;  // size query
        mip_series(nullptr, d_out, d_img, d, box, rAxis, projectNum, (float)(3.14159 * 2 / (float)projectNum),
                   (hipStream_t)stream);
        return 0;
    });
}

extern "C" int mp3dgpu(float *h_MP, unsigned int *sizeMP, float *h_img, unsigned int *sizeImg, bool flagXaxis,
                       bool flagYaxis, int projectNum) {
    if (!flagXaxis && !flagYaxis) return -1;  // reference src/apifunc.cpp:528-529
    return guarded("mp3dgpu", [&]() -> int {
        const Dims d{(long long)sizeImg[0], (long long)sizeImg[1], (long long)sizeImg[2]};
        const long long rot_x = (long long)std::round(std::sqrt((double)(d.nx * d.nx + d.nz * d.nz)));
        const long long rot_y = (long long)std::round(std::sqrt((double)(d.ny * d.ny + d.nz * d.nz)));
        const float step = (float)(3.14159 * 2 / projectNum);  // :513
        StreamGuard sg;
        DeviceBuffer<float> d_img((size_t)d.count());
        MIL_HIP(hipMemcpyAsync(d_img.get(), h_img, (size_t)d.count() * sizeof(float), hipMemcpyHostToDevice, sg.s));
        if (flagXaxis) {
            mip_series(h_MP, nullptr, d_img.get(), d, Dims{d.nx, rot_y, rot_y}, 1, projectNum, step, sg.s);
            sizeMP[0] = d.nx; sizeMP[1] = rot_y; sizeMP[2] = projectNum;
        }
        if (flagYaxis) {
            const size_t y_start = (size_t)d.nx * rot_y * projectNum;  // :553 (offset applies even without X)
            mip_series(h_MP + y_start, nullptr, d_img.get(), d, Dims{rot_x, d.ny, rot_x}, 2, projectNum, step, sg.s);
            sizeMP[3] = rot_x; sizeMP[4] = d.ny; sizeMP[5] = projectNum;
        }
        return 0;
    });
}

extern "C" int atrans3dgpu_16bit(unsigned short *h_reg, float *iTmx, unsigned short *h_img2, unsigned int *imSize1,
                                 unsigned int *imSize2, int deviceNum) {
    // 16-bit path of the reference (src/api_reg.cpp:86-112): convert through float32, resample,
    // store with C truncation.  (Upstream's tex16 is never given a linear filter --
    // src/api_subfunc.cu:909-919 -- so its own result is a nearest-texel fetch; callers in the
    // reference tree never use this entry.  Out of the hot path.)
    const size_t n_out = (size_t)imSize1[0] * imSize1[1] * imSize1[2];
    const size_t n_in = (size_t)imSize2[0] * imSize2[1] * imSize2[2];
    std::vector<float> fin(n_in), fout(n_out);
    for (size_t i = 0; i < n_in; ++i) fin[i] = (float)h_img2[i];
    const int status = atrans3dgpu(fout.data(), iTmx, fin.data(), imSize1, imSize2, deviceNum);
    if (status != 0) return status;
    for (size_t i = 0; i < n_out; ++i) h_reg[i] = (unsigned short)(long long)fout[i];
    return 0;
}
