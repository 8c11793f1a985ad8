/*
 * mil_device.h -- thin device-pointer C ABI underneath libapi.h.
 *
 * libapi.h keeps the reference's host-pointer contract (every call uploads, computes,
 * downloads).  A host that keeps its volumes resident in HBM -- a time-lapse batch
 * driver, the benchmark, the parity tests -- binds these instead: plain pointers and
 * sizes, a HIP stream passed as `void *` (NULL = the engine's own stream), `int`
 * status (0 = ok).  Each entry names the reference interface it replaces
 * (paths relative to the reference tree).
 *
 * Volumes are float32, x fastest: idx = x + nx*(y + ny*z).  Half spectra are
 * interleaved complex64, dims [nz][ny][nx/2+1], unnormalised transforms.
 */
#ifndef MICROIMAGELIB_AMD_MIL_DEVICE_H
#define MICROIMAGELIB_AMD_MIL_DEVICE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library / device ---------------------------------------------------- */
const char *mil_version(void);
int mil_device_count(void);
int mil_set_device(int device);               /* cudaSetDevice(deviceNum), src/api_decon.cpp:102 */
/* cudaMemGetInfo of `device` (src/api_decon.cpp:103, queryDevice) plus what the library's pool holds free on it -- memory a
 * caller may count as available.  Does not change the calling thread's device. */
int mil_dev_mem_info(int device, size_t *free_bytes, size_t *total_bytes);
int mil_snap_transform_size(int n);            /* snapTransformSize, src/api_subfunc.cu:57-87 */
/* Interpolation weights of every resample / cost kernel: 0 = exact float32 (default), 8 = rounded to 8 fractional bits,
 * emulating the CUDA texture unit the reference samples through (include/cukernel.cuh:510-522, 545-546; rounding to nearest
 * is an assumption, the CUDA guide documents only the format).  A comparison aid for data produced by the CUDA build;
 * process-wide; any other value returns control to MIL_TEX_WEIGHT_BITS.  Returns the setting in force. */
int mil_set_texture_weight_bits(int bits);
/* Device buffers freed inside the library are kept for the next call of the same size (the reference re-allocates ~2 GB per
 * libapi.h call), and decon_singleview / decon_dualview keep their engine + OTFs per (device, FFT box); this returns all of
 * it to the driver (entries a concurrent call is using are left alone).  Returns the number of pool bytes released.
 * MIL_POOL_MB caps the pool per device (default min(6144 MB, 1/16 of the device), least recently freed blocks go first). */
size_t mil_release_cached_memory(void);
/* Seconds of this THREAD's last decon_singleview / decon_dualview call (GPU modes), which deconRecords[10] has no slots for:
 * out5 = {engine + device buffers, OTFs (0 when cached), host->device upload + pad + initial estimate, RL loop,
 * crop + device->host download}. */
int mil_decon_last_call_split(double *out5);
size_t mil_cached_memory_bytes(void);             /* bytes the pool holds right now (all devices) */

/* ---- RL element-wise kernels (in place) ----------------------------------- */
int mil_rl_ratio(float *d_T, const float *d_A, size_t n, void *stream);               /* div3Dgpu  src/api_subfunc.cu:333 */
int mil_rl_update(float *d_E, const float *d_T, size_t n, float eps, void *stream);   /* multi3Dgpu :281 + maxvalue3Dgpu :477 */
int mil_cmul(void *d_S, const void *d_OTF, size_t n_complex, void *stream);           /* multicomplex3Dgpu :305 */
int mil_sum_f64(const float *d_in, size_t n, double *h_out, void *stream);            /* sum3Dgpu :386 */

/* ---- geometry ------------------------------------------------------------- */
/* padPSFgpu :1691 (+ multivaluegpu scale, + flipgpu :1680 when flip != 0); psf dims <= box dims */
int mil_pad_psf(float *d_out, const float *d_psf, int fx, int fy, int fz, int px, int py, int pz, float scale,
                int flip, void *stream);
int mil_pad_stack(float *d_out, const float *d_in, int fx, int fy, int fz, int sx, int sy, int sz, void *stream);  /* padstackgpu :1713 */
int mil_crop_stack(float *d_out, const float *d_in, int sx, int sy, int sz, int fx, int fy, int fz, void *stream); /* cropgpu :1736 */
int mil_align_size(float *d_out, const float *d_in, int ox, int oy, int oz, int ix, int iy, int iz, void *stream); /* alignsize3Dgpu :1779 */
int mil_flip3(float *d_out, const float *d_in, int nx, int ny, int nz, void *stream);                             /* flipgpu :1680 */
int mil_rot_y90(float *d_out, const float *d_in, int nx, int ny, int nz, int direction, void *stream);            /* rotbyyaxis :528 */
int mil_max_projection(float *d_out, const float *d_in, int nx, int ny, int nz, int direction, void *stream);     /* maxprojection :490 */
/* Projections (mil_max_projection, mil_mip3d, mil_resample_mip_z) assume finite, NaN-free volumes: the reference's sequential
 * `a = a > v ? a : v` is order dependent on NaN (a NaN wipes the running maximum and survives only as a column's last sample),
 * which the segmented maxima here do not reproduce. */
/* mip3dgpu (src/apifunc.cpp:576-644) on device memory: projectNum rotations about x (rAxis 1) or y (2), each resampled into the
 * rotated box and projected along z; d_out receives [projectNum][size3[1]][size3[0]] floats.  d_out == NULL only reports
 * size3 = {width, height, projectNum}.  Synchronises `stream` before it returns. */
int mil_mip3d(float *d_out, const float *d_img, int nx, int ny, int nz, int rAxis, long long projectNum, int *size3,
              void *stream);
/* One angle of that series with the caller's matrix: the source resampled into a box of (bx, by, bz) by m12
 * (affinetransformkernel, include/cukernel.cuh:501-524) and projected along the box's z (maxprojectionkernel :396-418,
 * accumulator starting at 0) in ONE pass -- d_plane receives [by][bx] floats, the box itself is never written.  Bit-identical
 * to mil_affine_resample + mil_max_projection(direction 1) on finite data. */
int mil_resample_mip_z(float *d_plane, const float *d_src, int bx, int by, int bz, int sx, int sy, int sz, const float *m12,
                       void *stream);

/* ---- whole stages on device-resident volumes ----------------------------------
 * The libapi.h calls without their two PCIe trips, for hosts that keep a time point in HBM from the TIFF reader to the TIFF
 * writer (cli/spim_fusion_batch_main.cpp).  Same kernels in the same order as the host-pointer entries: the same bits. */
/* device / pinned host memory, streams and transfers, so that such a host needs no HIP header (sizes in bytes) */
int mil_dev_alloc(void **d_ptr, size_t bytes);     /* from the library's per-device pool */
void mil_dev_free(void *d_ptr);
int mil_host_alloc(void **h_ptr, size_t bytes);    /* pinned */
void mil_host_free(void *h_ptr);
int mil_stream_create(void **stream);              /* non-blocking */
void mil_stream_destroy(void *stream);             /* waits for the work queued on it */
int mil_stream_sync(void *stream);
int mil_upload(void *d_dst, const void *h_src, size_t bytes, void *stream);     /* asynchronous on `stream` */
int mil_download(void *h_dst, const void *d_src, size_t bytes, void *stream);
/* 16-bit stacks: readtifstack's widening (float)v (src/apifunc.cpp:166-183) and writetifstack's C cast (unsigned short)f
 * (:255: truncation toward zero, low 16 bits; values no long long holds, and NaN, give 0) -- on the device, so that 16-bit
 * files cross PCIe as 16 bits */
int mil_u16_to_f32(float *d_out, const unsigned short *d_in, size_t n, void *stream);
int mil_f32_to_u16(unsigned short *d_out, const float *d_in, size_t n, void *stream);
/* imresize3d (src/apifunc.cpp:429-446): output (sx, sy, sz) from input (sx2, sy2, sz2) */
int mil_imresize3d(float *d_out, const float *d_in, long long sx, long long sy, long long sz, long long sx2, long long sy2,
                   long long sz2, void *stream);
/* reg3d (src/api_reg.cpp:264-607; libapi.h:29-31) without gpuMemMode / deviceNum: all five regChoice paths, the centred
 * align of a differently sized source, records[11] as there.  d_reg (imSize1 floats) must not overlap the inputs.  Returns
 * when d_reg is complete. */
int mil_reg3d(float *d_reg, float *iTmx, const float *d_img1, const float *d_img2, const unsigned int *imSize1,
              const unsigned int *imSize2, int regChoice, int affMethod, int flagTmx, float FTOL, int itLimit, int verbose,
              float *records, void *stream);
/* decon_singleview / decon_dualview (src/api_decon.cpp:53-331, 333-704; libapi.h:42-46), GPU mode: images and result are
 * device pointers of imSize floats (the images are only read), the PSFs host pointers; the cached engine (per device and FFT
 * box) runs on its own stream ordered after `stream`; returns when d_decon is complete.  deconRecords[10] as there. */
int mil_decon_singleview(float *d_decon, const float *d_img, const unsigned int *imSize, const float *h_psf,
                         const unsigned int *psfSize, int flagConstInitial, int itNumForDecon, int verbose, float *deconRecords,
                         int flagUnmatch, const float *h_psf_bp, void *stream);
int mil_decon_dualview(float *d_decon, const float *d_img1, const float *d_img2, const unsigned int *imSize, const float *h_psf1,
                       const float *h_psf2, const unsigned int *psfSize, int flagConstInitial, int itNumForDecon, int verbose,
                       float *deconRecords, int flagUnmatch, const float *h_psf_bp1, const float *h_psf_bp2, void *stream);

/* ---- deconvolution engine -------------------------------------------------- */
typedef struct mil_decon mil_decon;
/* Transform backends of the engine.  FUSED: hand-written LDS-tiled transforms with the OTF
 * product, RL ratio and estimate update fused in (every snapTransformSize box up to 1024 per axis:
 * powers of two from 16 and multiples of 64 from 192; x also 2048);
 * ROCFFT: rocFFT plans + separate element-wise kernels (any box).  AUTO picks FUSED when the
 * box allows it.  OTF / spectrum buffers are opaque: `mil_decon_spectrum_count` complex64
 * elements, layout private to the backend (mil_decon_otf_to_natural converts). */
#define MIL_BACKEND_AUTO 0
#define MIL_BACKEND_ROCFFT 1
#define MIL_BACKEND_FUSED 2
/* FFT box (nx,ny,nz): plans / tables + scratch allocated once (replaces the per-call
 * cufftPlan3d / cudaMalloc of decon_*_OTF1, src/api_subfunc.cu:3376-3396). */
int mil_decon_create(mil_decon **out, int nx, int ny, int nz, void *stream);
int mil_decon_create_backend(mil_decon **out, int nx, int ny, int nz, void *stream, int backend);
void mil_decon_destroy(mil_decon *e);
int mil_decon_backend(const mil_decon *e);
/* Fused backend: which kernel sequence a half view update runs (csrc/fft2p.hip): 0 = y | z-conv | y^-1 | x (four
 * passes over the spectrum), 3 = yz-conv | xy (two passes; nx, ny multiples of 64 from 192 / 128, nz a multiple of 64 with (ny / 64) x nz <= 4096), 1 / 2 = mixed forms kept for tests.
 * Chosen at creation: MIL_FX_TWOPASS if set, else 3 where the box allows it. */
int mil_decon_two_pass(const mil_decon *e);
/* Host-side query (no device needed): does the fused backend have the two-pass kernels for this FFT box, and how wide are the
 * column groups its OTFs are tiled in then (2 / 4 / 8 / 16; 16 also for every box on the classic sequence)?  Returns 1 / 0. */
int mil_fx_two_pass_box(int nx, int ny, int nz, int *otf_cols);
size_t mil_decon_real_count(const mil_decon *e);
size_t mil_decon_spectrum_count(const mil_decon *e);
/* d_out = C2R( R2C(d_in) * otf ), unnormalised: cufftExecR2C + multicomplex3Dgpu + cufftExecC2R,
 * src/api_subfunc.cu:3406-3408 */
int mil_decon_convolve(mil_decon *e, float *d_out, const float *d_in, const void *d_otf);
/* engine OTF -> natural [nz][ny][nx/2+1] complex64 (the cufftExecR2C output order, :3298-3299) */
int mil_decon_otf_to_natural(mil_decon *e, void *d_natural, const void *d_otf);
/* host only: freq_of_pos[p] = frequency index held at position p after the in-place mixed-radix
 * (8,..,8,{4,2},{3,5,7,11,13}) forward transform of length n (prime factors <= 13) */
int mil_fft_digit_reversal(int n, unsigned short *freq_of_pos);
/* genOTFgpu :3269 (flip != 0 applies flipgpu first: matched back projector) */
int mil_decon_gen_otf(mil_decon *e, void *d_otf, const float *d_psf, int px, int py, int pz, int flip);
int mil_decon_init_single(mil_decon *e, float *d_A, float *d_E, int const_init);             /* :3380-3388 */
int mil_decon_init_dual(mil_decon *e, float *d_A, float *d_B, float *d_E, int const_init);   /* :3606-3618 */
/* the RL loops, decon_singleview_OTF1 :3404-3416 / decon_dualview_OTF1 :3634-3660; asynchronous */
int mil_decon_run_single(mil_decon *e, float *d_E, const float *d_A, const void *d_otf, const void *d_otf_bp,
                         int iters);
int mil_decon_run_dual(mil_decon *e, float *d_E, const float *d_A, const float *d_B, const void *d_otf1,
                       const void *d_otf2, const void *d_otf_bp1, const void *d_otf_bp2, int iters);
int mil_decon_sync(mil_decon *e);
/* HIP-event timers bracketing kernel families on the engine's stream.
 * which (ROCFFT backend): 0 rl_update, 1 rl_ratio, 2 cmul, 3 rocFFT executes;
 * which (FUSED backend):  0 x pass + estimate update, 1 x pass + ratio, 2 z convolution, 3 y passes. */
/* enable: 0 off | 1 bracket every launch | n > 1 bracket every n-th launch (an event pair costs stream time) */
int mil_decon_timer_enable(mil_decon *e, int which, int enable);
int mil_decon_timer_read(mil_decon *e, int which, double *total_ms, size_t *launches); /* syncs, then resets */

/* ---- registration ---------------------------------------------------------- */
/* affineTransform + affinetransformkernel, src/api_subfunc.cu:943, include/cukernel.cuh:501 */
int mil_affine_resample(float *d_out, const float *d_src, int ox, int oy, int oz, int sx, int sy, int sz,
                        const float *m12, void *stream);
/* corrkernel + 2x sumgpu1D, include/cukernel.cuh:526, src/api_subfunc.cu:954-988: h_out = {sum s*s, sum s*t} */
int mil_corr_sums(const float *d_target, const float *d_src, int tx, int ty, int tz, int sx, int sy, int sz,
                  const float *m12, double *h_out, void *stream);
/* The same two sums with the source held as a copy with a replicated one-voxel border ([sz+2][sy+2][sx+2]: the
 * texture unit's clamp addressing materialised in memory; kept both plain and as (row y, row y+1) pairs) -- the form
 * mil_reg3d_affine evaluates ~10^3 times per registration.  Bit-identical to mil_corr_sums.  create fails (-1) when the volumes are beyond the kernel's 32-bit
 * offset range (mil_padded_source_eligible == 0); the caller then stays with mil_corr_sums. */
typedef struct mil_padded_source mil_padded_source;
int mil_padded_source_eligible(int tx, int ty, int tz, int sx, int sy, int sz);
int mil_padded_source_create(mil_padded_source **out, const float *d_src, int sx, int sy, int sz, void *stream);
void mil_padded_source_destroy(mil_padded_source *p);
int mil_corr_sums_padded(const float *d_target, const mil_padded_source *src, int tx, int ty, int tz,
                         const float *m12, double *h_out, void *stream);
/* test / profiling aid (process-wide): pin the padded-source kernel -- 0 plain layout + plane reuse, 1 row-pair layout,
 * 2 row-pair layout + plane reuse, 3 / 4 / 5 the same with two voxels per lane (target widths that are multiples of 128;
 * otherwise the one-voxel form runs), 6 plain layout, two voxels per lane, marching along y with row reuse; 7 / 9 / 10 the
 * CHUNKED (persistent) forms of 3 / 5 / 6 -- the grid is what the device holds at once, one publish per workgroup -- which
 * is what runs by default (8 is accepted and runs the tiled kernel 4; MIL_CORR_CHUNK=0 runs the tiled forms everywhere);
 * -1 (default) chooses per matrix.  Kernels 0-5 return the same bits; kernels 6-10 sum the same float32 voxel values in
 * another order (equal to ~1e-15 relative, bit-reproducible from run to run on one device). */
int mil_corr_padded_force_kernel(int kernel);
/* diagnostics (process-wide): release / acquire fences around the cost kernels' completion counters (1 / 0; -1: back to
 * MIL_CORR_FENCES, which is read once), and the workgroups per compute unit of the chunked cost kernels (0: built-in) */
int mil_corr_set_fences(int on);
int mil_corr_set_chunk_wgs_per_cu(int n);
/* host logic only (no GPU needed): the LAYOUT the per-matrix choice takes for m12 -- 0 plain + plane reuse, 1 row pairs,
 * 2 row pairs + plane reuse (the one-voxel kernels; what runs when the target width is not a multiple of 128) */
int mil_corr_padded_pick_kernel(const float *m12);
/* the same choice as a KERNEL number (mil_corr_padded_force_kernel's numbering, tiled forms): wide != 0 for target widths
 * that are multiples of 128 -- 3 / 5 (two voxels per lane), 1 (row pairs, both planes gathered: one voxel per lane also
 * there), or 6 (march along y).  The thresholds are the constants kPlainMaxRate, kPairsReuseMaxRate, kMarchYMaxRate and
 * kMarchYMaxSpread of csrc/corr_padded.hip (fitted on timed trajectory matrices, profiles/r04/reg_cost_pick.txt): kernel 6
 * where the expected re-gather rate along y is below kMarchYMaxRate AND below the rate along z AND the 64 lanes of a wave
 * stay within kMarchYMaxSpread source rows + planes; wide == 0: as above */
int mil_corr_padded_pick_kernel2(const float *m12, int wide);
/* host logic only: which kernel mil_affine_resample takes for an output of (ox, oy, oz) from a source of (sx, sy, sz) --
 * 2 the marching kernel (csrc/resample_march.hip: two voxels per lane, plane / row reuse; sources below 2 GiB with planes
 * below 2^22 voxels), 1 one voxel per thread with 32-bit offsets (planes below 2^24 voxels), 0 one voxel per thread with
 * 64-bit address arithmetic (anything larger, sx < 2, or the texture-weight emulation).  All three return the same bits. */
int mil_affine_resample_path(int ox, int oy, int oz, int sx, int sy, int sz);
/* cost-kernel launches of this thread's last mil_reg3d_affine (records[5] counts evaluations like the reference; repeats of
 * a point just visited are answered from the last results, same value, no launch) */
int mil_reg3d_last_cost_launches(void);
/* Audit aid: what this thread's next mil_reg3d_affine / reg3d calls evaluate, in order.  begin(capacity > 0) starts a fresh
 * trace of up to `capacity` evaluations, begin(0) stops it.  read copies min(max_entries, stored) entries of 13 floats --
 * the matrix of that evaluation (the reference's global affCoef, src/api_subfunc.cu:2378-2383) and the cost handed to
 * powell() -- and returns the number of evaluations seen since begin. */
int mil_reg3d_trace_begin(int capacity);
int mil_reg3d_trace_read(float *out13, int max_entries);
/* reg3d_affine1, src/api_subfunc.cu:2733-2994, on device-resident volumes of equal size */
int mil_reg3d_affine(float *d_reg, float *iTmx, const float *d_img1, const float *d_img2, int nx, int ny, int nz,
                     int affMethod, int flagTmx, float FTOL, int itLimit, int verbose, float *records,
                     void *stream);

/* regChoice 4 pre-registration: 2-D MIP shift search, src/api_reg.cpp:448-497 + reg2d_shiftalign1 / X1
 * (src/api_subfunc.cu:1860-2120).  shift3 = {x, y, z} translation in voxels (host). */
int mil_reg3d_mip_shift(float *shift3, const float *d_img1, const float *d_img2, int nx, int ny, int nz, void *stream);

/* regChoice 1 / 3 pre-registration: reg3d_phasor1, src/api_subfunc.cu:2465-2590.  shift3 = {x, y, z} (host),
 * source(p) ~ target(p - shift). */
int mil_reg3d_phasor_shift(long long *shift3, const float *d_img1, const float *d_img2, int nx, int ny, int nz,
                           void *stream);
/* circshiftgpu (wrap != 0) / imshiftgpu (zero fill), src/api_subfunc.cu:827-847: out[p] = in[p - d] */
int mil_shift_volume(float *d_out, const float *d_in, int nx, int ny, int nz, long long dx, long long dy, long long dz,
                     int wrap, void *stream);

/* ---- host-side pieces of the path (no GPU needed) --------------------------- */
/* powell(), src/api_powell.c:305-360.  0-based: p[n]; xi row-major with leading dimension ld,
 * column j = direction j.  f is called with a 0-based parameter vector. */
typedef float (*mil_cost_fn)(const float *x, void *user);
int mil_powell(float *p, float *xi, int ld, int n, float ftol, int *iter, float *fret, mil_cost_fn f, void *user,
               int *totalIt, int itLimit);
void mil_p2matrix(float *m12, const float *p12);                    /* p2matrix :557 (0-based p) */
void mil_matrix2p(float *p12, const float *m12);                    /* matrix2p :569 */
void mil_dof9tomatrix(float *m12, const float *p9, int dofNum);     /* dof9tomatrix :715 (0-based p) */
void mil_matrixmultiply(float *m, const float *m1, const float *m2);/* matrixmultiply :584 */
void mil_rot2matrix(float *m12, float theta, long long sx, long long sy, long long sz, int rotAxis); /* :626 */


/* Host 3-D real transforms of the gpuMemMode 0 path (reference: fftwf_plan_dft_r2c_3d / c2r_3d at
 * src/api_subfunc.cu:3334-3336; bundled mixed-radix implementation, csrc/cpu_fft.cpp).  HOST pointers; nx even.
 * inverse == 0: h_real [nz][ny][nx] -> h_spec [nz][ny][nx/2+1] complex64; inverse != 0: h_spec -> h_real, unnormalised,
 * h_spec destroyed.  threads <= 0: MIL_CPU_THREADS or 1. */
int mil_cpu_rfft3d(float *h_real, void *h_spec, int nx, int ny, int nz, int inverse, int threads);

#ifdef __cplusplus
}
#endif
#endif /* MICROIMAGELIB_AMD_MIL_DEVICE_H */
