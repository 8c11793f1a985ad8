// deconDualView -- joint dual-view Richardson-Lucy deconvolution of two registered TIFF stacks.
// Same flags and defaults as the reference tool (src/decon_dv.cpp): -i1 -i2 -fp1 -fp2 -o -bp1 -bp2 -it [10]
// -cON/-cOFF -gm -dev -bit -verbON/-verbOFF -log.
#include <chrono>

#include "cli_common.h"

int main(int argc, char **argv) {
    std::string img1 = "../Data/SPIMA_0_crop.tif", img2 = "../Data/SPIMB_0_crop.tif", psf1 = "../Data/PSFA.tif",
                psf2 = "../Data/PSFA.tif", bp1 = "../Data/PSFA_bp.tif", bp2 = "../Data/PSFB_bp.tif",
                out_path = "../Data/Decon_0.tif";
    bool const_init = false, verbose = true, unmatch = false, bit_from_input = true;
    int iters = 10, device = 0, mem_mode = -1;
    unsigned int bits = 0;

    cli::Parser p("deconDualView", "Joint deconvolution for dual-view 3D images",
                  "-i1 <inputImageName1> -i2 <inputImageName2> -fp1 <psfImageName1> -fp2 <psfImageName2> -o <outputImageName>");
    p.add("-i1", "<filename>", "Input image filename 1 (SPIM A) (mandatory)", [&](const char *v) { img1 = v; });
    p.add("-i2", "<filename>", "Input image filename 2 (SPIM B) (mandatory)", [&](const char *v) { img2 = v; });
    p.add("-fp1", "<filename>", "PSF1 (forward projector 1) image filename (mandatory)", [&](const char *v) { psf1 = v; });
    p.add("-fp2", "<filename>", "PSF2 (forward projector 2) image filename (mandatory)", [&](const char *v) { psf2 = v; });
    p.add("-o", "<filename>", "Output filename of the deconvolved image (mandatory)", [&](const char *v) { out_path = v; });
    p.add("-bp1", "<filename>", "Backward projector 1 filename [flip of PSF1]", [&](const char *v) { bp1 = v; unmatch = true; });
    p.add("-bp2", "<filename>", "Backward projector 2 filename [flip of PSF2]", [&](const char *v) { bp2 = v; unmatch = true; });
    p.add("-it", "<int>", "Iteration number of the deconvolution [10]", [&](const char *v) { iters = atoi(v); });
    p.add("-cON", nullptr, "constant as initialization", [&](const char *) { const_init = true; });
    p.add("-cOFF", nullptr, "average of the input images as initialization [default]", [&](const char *) { const_init = false; });
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
    const auto t0 = std::chrono::steady_clock::now();
    cli::Volume a = cli::load_tiff(img1), b = cli::load_tiff(img2), pa = cli::load_tiff(psf1), pb = cli::load_tiff(psf2);
    if (!cli::same_size(a, b)) {
        printf("\tThe two input images don't have the same image size, processing stopped !!!\n");
        return 1;
    }
    if (!cli::same_size(pa, pb)) {
        printf("\tThe two PSFs don't have the same image size, processing stopped !!!\n");
        return 1;
    }
    cli::Volume ba, bb;
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
    if (bit_from_input) bits = a.bits;
    printf("=== Deconvolution settings ...\n\tImage size %u x %u x %u, PSF size %u x %u x %u\n", a.size[0], a.size[1],
           a.size[2], pa.size[0], pa.size[1], pa.size[2]);
    printf("\tIterations: %d; processing: %s; device %d; initialization: %s; output %u bit\n", iters,
           cli::mode_text(mem_mode), device, const_init ? "constant" : "average of the two views", bits);

    std::vector<float> out(a.count(), 0.f);
    float records[20] = {0};
    const auto t1 = std::chrono::steady_clock::now();
    printf("=== Deconvolution starting ...\n");
    const int status = decon_dualview(out.data(), a.data.data(), b.data.data(), a.size, pa.data.data(), pb.data.data(),
                                      pa.size, const_init, iters, device, mem_mode, verbose, records, unmatch,
                                      ba.data.data(), bb.data.data());
    const auto t2 = std::chrono::steady_clock::now();
    printf("runStatus: %d\nGPU mode: %d\n", status, (int)records[0]);
    writetifstack(const_cast<char *>(out_path.c_str()), out.data(), a.size, (unsigned short)bits);
    const auto t3 = std::chrono::steady_clock::now();
    auto sec = [](auto x, auto y) { return std::chrono::duration<double>(y - x).count(); };
    printf("\n****Time cost for  image reading/writing: %2.3f s\n", sec(t0, t1) + sec(t2, t3));
    printf("\n****Time cost for  deconvolution: %2.3f s\n", sec(t1, t2));
    printf("\n****Time cost for  whole processing: %2.3f s\n", sec(t0, t3));
    return 0;
}
