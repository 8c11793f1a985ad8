// spimFusion -- dual-view fusion of one diSPIM pair: resample both views to isotropic voxels, rotate view B,
// register it onto view A, joint Richardson-Lucy deconvolution.  Same flags and defaults as the reference tool
// (src/spim_fusion.cpp:22-79, 93-134): -i1 -i2 -fp1 -fp2 -o | -pxx1 -pxy1 -pxz1 -pxx2 -pxy2 -pxz2 -imgrot |
// -oreg1 -oreg2 -itmx -otmx -regc -affm -ftol -itreg | -bp1 -bp2 -it -cON/-cOFF | -gm -dev -bit -verbON/-verbOFF -log.
// Pipeline order follows src/spim_fusion.cpp:563-672: imresize3d (A), imoperation3D + imresize3d (B), reg3d,
// decon_dualview.
#include <chrono>
#include <cmath>

#include "cli_common.h"

namespace {

unsigned int scaled(unsigned int n, float num, float den) { return (unsigned int)std::lround((float)n * num / den); }

cli::Volume resize_to(const cli::Volume &in, const unsigned int *size, int device) {
    if (in.size[0] == size[0] && in.size[1] == size[1] && in.size[2] == size[2]) return in;
    cli::Volume out;
    out.bits = in.bits;
    for (int i = 0; i < 3; ++i) out.size[i] = size[i];
    out.data.resize(out.count());
    (void)imresize3d(out.data.data(), const_cast<float *>(in.data.data()), size[0], size[1], size[2], in.size[0],
                     in.size[1], in.size[2], device);
    return out;
}

}  // namespace

int main(int argc, char **argv) {
    std::string img1 = "../Data/SPIMA_0_crop.tif", img2 = "../Data/SPIMB_0_crop.tif", psf1 = "../Data/PSFA.tif",
                psf2 = "../Data/PSFA.tif", out_path = "../Data/Decon_0.tif", reg1_path = "../Data/SPIMA_reg_0.tif",
                reg2_path = "../Data/SPIMB_reg_0.tif", itmx_path = "../Data/tmxIn.tmx", otmx_path = "../Data/tmxOut.tmx",
                bp1 = "../Data/PSFA_bp.tif", bp2 = "../Data/PSFB_bp.tif";
    float px1[3] = {0.1625f, 0.1625f, 1.0f}, px2[3] = {0.1625f, 0.1625f, 1.0f};
    int rotation = -1, reg_choice = 2, aff_method = 6, it_limit = 3000, iters = 10, device = 0, mem_mode = -1;
    float ftol = 0.0001f;
    bool save_reg1 = false, save_reg2 = false, have_itmx = false, want_otmx = false, unmatch = false, const_init = false,
         verbose = true, bit_from_input = true;
    unsigned int bits = 16;

    cli::Parser p("spimFusion", "Fusion (registration and joint deconvolution) of dual-view light-sheet images",
                  "-i1 <inputImageName1> -i2 <inputImageName2> -fp1 <psfImageName1> -fp2 <psfImageName2> -o <outputImageName>");
    p.add("-i1", "<filename>", "Input image filename 1 (SPIM A) (mandatory)", [&](const char *v) { img1 = v; });
    p.add("-i2", "<filename>", "Input image filename 2 (SPIM B) (mandatory)", [&](const char *v) { img2 = v; });
    p.add("-fp1", "<filename>", "PSF1 (forward projector 1) image filename (mandatory)", [&](const char *v) { psf1 = v; });
    p.add("-fp2", "<filename>", "PSF2 (forward projector 2) image filename (mandatory)", [&](const char *v) { psf2 = v; });
    p.add("-o", "<filename>", "Output filename of the deconvolved image (mandatory)", [&](const char *v) { out_path = v; });
    p.add("-pxx1", "<float>", "Pixel size X 1 (um) [0.1625]", [&](const char *v) { px1[0] = (float)atof(v); });
    p.add("-pxy1", "<float>", "Pixel size Y 1 (um) [0.1625]", [&](const char *v) { px1[1] = (float)atof(v); });
    p.add("-pxz1", "<float>", "Pixel size Z 1 (um) [1.0]", [&](const char *v) { px1[2] = (float)atof(v); });
    p.add("-pxx2", "<float>", "Pixel size X 2 (um) [0.1625]", [&](const char *v) { px2[0] = (float)atof(v); });
    p.add("-pxy2", "<float>", "Pixel size Y 2 (um) [0.1625]", [&](const char *v) { px2[1] = (float)atof(v); });
    p.add("-pxz2", "<float>", "Pixel size Z 2 (um) [1.0]", [&](const char *v) { px2[2] = (float)atof(v); });
    p.add("-imgrot", "<int>", "Image 2 rotation: 0 none | 1: 90 deg by Y-axis | -1: -90 deg by Y-axis [-1]",
          [&](const char *v) { rotation = atoi(v); });
    p.add("-oreg1", "<filename>", "Output filename of the registered image 1 [no output]",
          [&](const char *v) { reg1_path = v; save_reg1 = true; });
    p.add("-oreg2", "<filename>", "Output filename of the registered image 2 [no output]",
          [&](const char *v) { reg2_path = v; save_reg2 = true; });
    p.add("-itmx", "<filename>", "Input transformation matrix filename [identity matrix]",
          [&](const char *v) { itmx_path = v; have_itmx = true; });
    p.add("-otmx", "<filename>", "Output transformation matrix filename [no output]",
          [&](const char *v) { otmx_path = v; want_otmx = true; });
    p.add("-regc", "<int>", "0 apply matrix | 1 phasor | 2 affine | 3 phasor+affine | 4 2D MIP+affine [2]",
          [&](const char *v) { reg_choice = atoi(v); });
    p.add("-affm", "<int>", "0 none | 1 3 DOF | 2 6 | 3 7 | 4 9 | 5 12 | 6 6->12 | 7 3->6->9->12 [6]",
          [&](const char *v) { aff_method = atoi(v); });
    p.add("-ftol", "<float>", "Tolerance of the stop point [0.0001]", [&](const char *v) { ftol = (float)atof(v); });
    p.add("-itreg", "<int>", "Maximum iteration number for registration [3000]", [&](const char *v) { it_limit = atoi(v); });
    p.add("-bp1", "<filename>", "Backward projector 1 filename [flip of PSF1]", [&](const char *v) { bp1 = v; unmatch = true; });
    p.add("-bp2", "<filename>", "Backward projector 2 filename [flip of PSF2]", [&](const char *v) { bp2 = v; unmatch = true; });
    p.add("-it", "<int>", "Iteration number of the deconvolution [10]", [&](const char *v) { iters = atoi(v); });
    p.add("-cON", nullptr, "constant as initialization", [&](const char *) { const_init = true; });
    p.add("-cOFF", nullptr, "average of the two views as initialization [default]", [&](const char *) { const_init = false; });
    p.add("-gm", "<int>", "-1 automatic | 0 CPU (host RL, MIL_CPU_THREADS=n) | 1 efficient GPU | 2 memory-saved GPU [-1]",
          [&](const char *v) { mem_mode = atoi(v); });
    p.add("-dev", "<int>", "GPU device [0]", [&](const char *v) { device = atoi(v); });
    p.add("-bit", "<int>", "Output image bit: 16 or 32 [same as input image]",
          [&](const char *v) { bits = (unsigned)atoi(v); bit_from_input = false; });
    p.add("-verbON", nullptr, "verbose information on [default]", [&](const char *) { verbose = true; });
    p.add("-verbOFF", nullptr, "verbose information off", [&](const char *) { verbose = false; });
    p.add("-log", "<filename>", "Log filename (accepted, not used)", [&](const char *) {});
    const int pr = p.parse(argc, argv);
    if (pr != 0) return pr > 0 ? EXIT_SUCCESS : EXIT_FAILURE;
    if (!cli::mode_text(mem_mode)) {
        printf("\tWrong GPU mode setting, processing stopped !!!\n");
        return 1;
    }
    auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = std::chrono::steady_clock::now();

    // ---- preprocessing: isotropic resampling, rotation of view B (src/spim_fusion.cpp:326-352, 563-600)
    printf("... Preprocessing ...\n");
    cli::Volume a_in = cli::load_tiff(img1), b_in = cli::load_tiff(img2);
    if (a_in.bits != 16 && a_in.bits != 32) {
        printf("\tOnly 16-bit or 32-bit images are supported, processing stopped !!!\n");
        return 1;
    }
    const unsigned int in_bits = b_in.bits;  // the reference keeps the bit depth of the LAST file it probed
    if (bit_from_input) bits = in_bits;
    unsigned int size1[3] = {a_in.size[0], scaled(a_in.size[1], px1[1], px1[0]), scaled(a_in.size[2], px1[2], px1[0])};
    unsigned int scaled2[3] = {scaled(b_in.size[0], px2[0], px1[0]), scaled(b_in.size[1], px2[1], px1[0]),
                               scaled(b_in.size[2], px2[2], px1[0])};
    int op_choice = 0;
    unsigned int size2[3] = {scaled2[0], scaled2[1], scaled2[2]};
    if (rotation == 1 || rotation == -1) {  // +-90 degrees about Y swaps the x and z extents
        op_choice = rotation == 1 ? 1 : 2;
        size2[0] = scaled2[2];
        size2[2] = scaled2[0];
    }
    cli::Volume a = resize_to(a_in, size1, device);
    cli::Volume b_rot;
    if (op_choice == 0) {
        b_rot = b_in;
    } else {
        b_rot.bits = b_in.bits;
        b_rot.data.resize(b_in.count());
        (void)imoperation3D(b_rot.data.data(), b_rot.size, b_in.data.data(), b_in.size, op_choice, device);
    }
    cli::Volume b = resize_to(b_rot, size2, device);
    const auto t1 = std::chrono::steady_clock::now();
    printf("\tImage 1: %u x %u x %u, image 2: %u x %u x %u\n\tTime cost for  preprocessing: %2.3f s\n", a.size[0],
           a.size[1], a.size[2], b.size[0], b.size[1], b.size[2], sec(t0, t1));

    // ---- registration (src/spim_fusion.cpp:603-637)
    printf("... Registration ...\n");
    float tmx[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    if (have_itmx) {
        if (!fexists(itmx_path.c_str())) {
            printf("***** Input transformation matrix file does not exist: %s\n", itmx_path.c_str());
            return 1;
        }
        FILE *f = fopen(itmx_path.c_str(), "r");
        for (int i = 0; i < 12; ++i)
            if (fscanf(f, "%f", &tmx[i]) != 1) { printf("***** Malformed matrix file %s\n", itmx_path.c_str()); fclose(f); return 1; }
        fclose(f);
    }
    std::vector<float> b_reg(a.count(), 0.f);
    float reg_records[11] = {0}, decon_records[10] = {0};
    int status = reg3d(b_reg.data(), tmx, a.data.data(), b.data.data(), a.size, b.size, reg_choice, aff_method, have_itmx,
                       ftol, it_limit, device, mem_mode, verbose, reg_records);
    if (status != 0) printf("\tRegistration status: %d\n", status);
    if (want_otmx) {
        FILE *f = fopen(otmx_path.c_str(), "w");
        if (!f) { printf("***** Cannot write %s\n", otmx_path.c_str()); return 1; }
        for (int r = 0; r < 3; ++r) fprintf(f, "%f\t%f\t%f\t%f\t\n", tmx[4 * r], tmx[4 * r + 1], tmx[4 * r + 2], tmx[4 * r + 3]);
        fprintf(f, "%f\t%f\t%f\t%f\n", 0.0, 0.0, 0.0, 1.0);
        fclose(f);
    }
    if (save_reg1) writetifstack(const_cast<char *>(reg1_path.c_str()), a.data.data(), a.size, (unsigned short)in_bits);
    if (save_reg2) writetifstack(const_cast<char *>(reg2_path.c_str()), b_reg.data(), a.size, (unsigned short)in_bits);
    const auto t2 = std::chrono::steady_clock::now();
    printf("\tTime cost for  registration: %2.3f s\n", sec(t1, t2));

    // ---- joint deconvolution (src/spim_fusion.cpp:640-668)
    printf("... Deconvolution ...\n");
    cli::Volume pa = cli::load_tiff(psf1), pb = cli::load_tiff(psf2), ba, bb;
    if (!cli::same_size(pa, pb)) {
        printf("\tThe two PSFs don't have the same image size, processing stopped !!!\n");
        return 1;
    }
    if (unmatch) {
        ba = cli::load_tiff(bp1);
        bb = cli::load_tiff(bp2);
        if (!cli::same_size(pa, ba) || !cli::same_size(pa, bb)) {
            printf("\tForward projector and backward projector don't have the same image size, processing stopped !!!\n");
            return 1;
        }
    } else {
        ba.data.resize(pa.count());
        bb.data.resize(pa.count());
    }
    std::vector<float> decon(a.count(), 0.f);
    status = decon_dualview(decon.data(), a.data.data(), b_reg.data(), a.size, pa.data.data(), pb.data.data(), pa.size,
                            const_init, iters, device, mem_mode, verbose, decon_records, unmatch, ba.data.data(),
                            bb.data.data());
    if (status != 0) printf("\tDeconvolution status: %d\n", status);
    writetifstack(const_cast<char *>(out_path.c_str()), decon.data(), a.size, (unsigned short)bits);
    const auto t3 = std::chrono::steady_clock::now();
    printf("\tTime cost for  deconvolution: %2.3f s\n", sec(t2, t3));
    printf("\n=== Processing completed, time cost for  whole processing: %2.3f s\n", sec(t0, t3));
    return 0;
}
