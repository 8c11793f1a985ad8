// deconSingleView -- single-view Richardson-Lucy deconvolution of a TIFF stack.
// Same flags, defaults and outputs as the reference tool (src/decon_sv.cpp): -i -fp -o -bp -it -gm -dev
// -cON/-cOFF -bit -verbON/-verbOFF -log (accepted, ignored upstream too: src/decon_sv.cpp:129-132).
#include <chrono>

#include "cli_common.h"

int main(int argc, char **argv) {
    std::string img_path = "../Data/SPIMA_0_crop.tif", psf_path = "../Data/PSF.tif", bp_path = "../Data/PSF_bp.tif",
                out_path = "../Data/Decon_0.tif";
    bool const_init = false, verbose = true, unmatch = false, bit_from_input = true;
    int iters = 20, device = 0, mem_mode = -1;
    unsigned int bits = 0;

    cli::Parser p("deconSingleView", "Deconvolution for single-view 3D image",
                  "-i <inputImageName> -fp <psfImageName> -o <outputImageName>");
    p.add("-i", "<filename>", "Input image filename (mandatory)", [&](const char *v) { img_path = v; });
    p.add("-fp", "<filename>", "PSF (forward projector) image filename (mandatory)", [&](const char *v) { psf_path = v; });
    p.add("-o", "<filename>", "Output filename of the deconvolved image (mandatory)", [&](const char *v) { out_path = v; });
    p.add("-bp", "<filename>", "Backward projector filename [flip of PSF]", [&](const char *v) { bp_path = v; unmatch = true; });
    p.add("-it", "<int>", "Iteration number of the deconvolution [20]", [&](const char *v) { iters = atoi(v); });
    p.add("-gm", "<int>", "-1 automatic | 0 CPU (host RL, MIL_CPU_THREADS=n) | 1 efficient GPU | 2 memory-saved GPU [-1]",
          [&](const char *v) { mem_mode = atoi(v); });
    p.add("-dev", "<int>", "GPU device [0]", [&](const char *v) { device = atoi(v); });
    p.add("-cON", nullptr, "constant as initialization", [&](const char *) { const_init = true; });
    p.add("-cOFF", nullptr, "input image as initialization [default]", [&](const char *) { const_init = false; });
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
    printf("=== Deconvolution settings ...\n\tInput image: %s\n\tPSF: %s\n\tOutput: %s\n", img_path.c_str(),
           psf_path.c_str(), out_path.c_str());
    cli::Volume img = cli::load_tiff(img_path), psf = cli::load_tiff(psf_path), bp;
    if (unmatch) {
        bp = cli::load_tiff(bp_path);
        if (!cli::same_size(psf, bp)) {
            printf("\tForward projector and backward projector don't have the same image size, processing stopped !!!\n");
            return 1;
        }
    } else {
        bp.data.resize(psf.count());  // never read when flagUnmatch is false
    }
    if (bit_from_input) bits = img.bits;
    printf("\tInput image size %u x %u x %u\n\tPSF image size %u x %u x %u\n", img.size[0], img.size[1], img.size[2],
           psf.size[0], psf.size[1], psf.size[2]);
    printf("\tIterations: %d; processing: %s; device %d; initialization: %s; output %u bit\n", iters,
           cli::mode_text(mem_mode), device, const_init ? "constant" : "input image", bits);

    std::vector<float> out(img.count(), 0.f);
    float records[20] = {0};
    const auto t1 = std::chrono::steady_clock::now();
    printf("=== Deconvolution starting ...\n");
    const int status = decon_singleview(out.data(), img.data.data(), img.size, psf.data.data(), psf.size, const_init,
                                        iters, device, mem_mode, verbose, records, unmatch, bp.data.data());
    const auto t2 = std::chrono::steady_clock::now();
    printf("runStatus: %d\nGPU mode: %d\n", status, (int)records[0]);
    writetifstack(const_cast<char *>(out_path.c_str()), out.data(), img.size, (unsigned short)bits);
    const auto t3 = std::chrono::steady_clock::now();
    auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    printf("\n****Time cost for  image reading/writing: %2.3f s\n", sec(t0, t1) + sec(t2, t3));
    printf("\n****Time cost for  deconvolution: %2.3f s\n", sec(t1, t2));
    printf("\n****Time cost for  whole processing: %2.3f s\n", sec(t0, t3));
    return 0;
}
