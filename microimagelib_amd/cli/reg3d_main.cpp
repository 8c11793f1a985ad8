// reg3D -- 3-D affine registration of a source TIFF stack onto a target stack.
// Same flags and defaults as the reference tool (src/reg3D.cpp): -t -s -o -itmx -otmx -regc [2] -affm [6, as the
// reference CODE does although its help text says 7: src/reg3D.cpp:33,74] -ftol [1e-4] -it [3000] -gm -dev -bit
// -verbON/-verbOFF -log.  Matrix files: 3 rows of 4 "%f\t" then "0 0 0 1" (src/reg3D.cpp:289-292, 316-326).
#include <chrono>

#include "cli_common.h"

int main(int argc, char **argv) {
    std::string tgt_path = "../Data/SPIMA_0_crop.tif", src_path = "../Data/SPIMA_0_crop2.tif",
                out_path = "../Data/SPIMA_0_reg2.tif", itmx_path = "../Data/tmxIn.tmx", otmx_path = "../Data/tmxOut.tmx";
    bool have_itmx = false, want_otmx = false, verbose = true, bit_from_input = true;
    int reg_choice = 2, aff_method = 6, it_limit = 3000, device = 0, mem_mode = -1;
    float ftol = 0.0001f;
    unsigned int bits = 0;

    cli::Parser p("reg3D", "3D image registration", "-t <targetImageName> -s <sourceImageName> -o <outputImageName>");
    p.add("-t", "<filename>", "Target image filename (fixed image) (mandatory)", [&](const char *v) { tgt_path = v; });
    p.add("-s", "<filename>", "Source image filename (moving image) (mandatory)", [&](const char *v) { src_path = v; });
    p.add("-o", "<filename>", "Output filename of the registered image (mandatory)", [&](const char *v) { out_path = v; });
    p.add("-itmx", "<filename>", "Input transformation matrix filename [identity matrix]",
          [&](const char *v) { itmx_path = v; have_itmx = true; });
    p.add("-otmx", "<filename>", "Output transformation matrix filename [no output]",
          [&](const char *v) { otmx_path = v; want_otmx = true; });
    p.add("-regc", "<int>", "0 apply matrix | 1 phasor | 2 affine | 3 phasor+affine | 4 2D MIP+affine [2]",
          [&](const char *v) { reg_choice = atoi(v); });
    p.add("-affm", "<int>", "0 none | 1 3 DOF | 2 6 | 3 7 | 4 9 | 5 12 | 6 6->12 | 7 3->6->9->12 [6]",
          [&](const char *v) { aff_method = atoi(v); });
    p.add("-ftol", "<float>", "Tolerance of the stop point [0.0001]", [&](const char *v) { ftol = (float)atof(v); });
    p.add("-it", "<int>", "Maximum iteration number [3000]", [&](const char *v) { it_limit = atoi(v); });
    p.add("-gm", "<int>", "-1 automatic | 0 CPU (not supported, as upstream) | 1 | 2 GPU [-1]",
          [&](const char *v) { mem_mode = atoi(v); });
    p.add("-dev", "<int>", "GPU device [0]", [&](const char *v) { device = atoi(v); });
    p.add("-bit", "<int>", "Output image bit: 16 or 32 [same as input image]",
          [&](const char *v) { bits = (unsigned)atoi(v); bit_from_input = false; });
    p.add("-verbON", nullptr, "verbose information on [default]", [&](const char *) { verbose = true; });
    p.add("-verbOFF", nullptr, "verbose information off", [&](const char *) { verbose = false; });
    p.add("-log", "<filename>", "Log filename (accepted, not used)", [&](const char *) {});
    const int pr = p.parse(argc, argv);
    if (pr != 0) return pr > 0 ? EXIT_SUCCESS : EXIT_FAILURE;

    const auto t0 = std::chrono::steady_clock::now();
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
    cli::Volume tgt = cli::load_tiff(tgt_path), src = cli::load_tiff(src_path);
    if (bit_from_input) bits = tgt.bits;
    printf("=== Registration settings ...\n\tTarget %s (%u x %u x %u)\n\tSource %s (%u x %u x %u)\n", tgt_path.c_str(),
           tgt.size[0], tgt.size[1], tgt.size[2], src_path.c_str(), src.size[0], src.size[1], src.size[2]);
    printf("\tregChoice %d, affMethod %d, FTOL %g, itLimit %d, input matrix: %s, device %d, output %u bit\n", reg_choice,
           aff_method, ftol, it_limit, have_itmx ? "yes" : "no", device, bits);

    std::vector<float> out(tgt.count(), 0.f);
    float records[11] = {0};
    const auto t1 = std::chrono::steady_clock::now();
    printf("=== Registration starting ...\n");
    const int status = reg3d(out.data(), tmx, tgt.data.data(), src.data.data(), tgt.size, src.size, reg_choice,
                             aff_method, have_itmx, ftol, it_limit, device, mem_mode, verbose, records);
    const auto t2 = std::chrono::steady_clock::now();
    printf("runStatus: %d\nGPU mode: %d\n", status, (int)records[0]);
    if (want_otmx) {
        FILE *f = fopen(otmx_path.c_str(), "w");
        if (!f) { printf("***** Cannot write %s\n", otmx_path.c_str()); return 1; }
        for (int r = 0; r < 3; ++r) fprintf(f, "%f\t%f\t%f\t%f\t\n", tmx[4 * r], tmx[4 * r + 1], tmx[4 * r + 2], tmx[4 * r + 3]);
        fprintf(f, "%f\t%f\t%f\t%f\n", 0.0, 0.0, 0.0, 1.0);
        fclose(f);
    }
    writetifstack(const_cast<char *>(out_path.c_str()), out.data(), tgt.size, (unsigned short)bits);
    const auto t3 = std::chrono::steady_clock::now();
    auto sec = [](auto x, auto y) { return std::chrono::duration<double>(y - x).count(); };
    printf("\n****Time cost for  image reading/writing: %2.3f s\n", sec(t0, t1) + sec(t2, t3));
    printf("\n****Time cost for  registration: %2.3f s\n", sec(t1, t2));
    printf("\n****Time cost for  whole processing: %2.3f s\n", sec(t0, t3));
    return 0;
}
