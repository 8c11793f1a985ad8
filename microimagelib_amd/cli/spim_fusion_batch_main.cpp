// spimFusionBatch -- dual-view fusion of a time-lapse: for every time point resample / rotate, register view B
// onto view A according to the registration mode, joint Richardson-Lucy deconvolution, projections.
// Same 34 (or 36) positional arguments, folder layout, matrix files and log as the reference tool
// (src/spim_fusion_batch.cpp:58-113, 135-190, 262-296, 613-966) -- "set same with the ImageJ diSPIMFusion plugin".
//
// Multi-GPU: argument 34 may be a comma-separated device list ("0,1,2,3"), or MIL_DEVICES names
// one; the default is the reference's single device.  One host WORKER THREAD per listed device -- its own HIP context,
// streams, engine, OTFs and buffers inside libapi.so, which serialises nothing across devices -- and the time points are
// dealt round-robin over the workers.  No volume ever crosses devices:
//   regMode 0 / 3   time points are independent;
//   regMode 1       the test time point is registered first (on the first device), then as regMode 0;
//   regMode 2       the starting matrix of time point i is the running average of the previous results
//                   (src/spim_fusion_batch.cpp:786-808): the chain state -- 24 floats -- is handed from the worker of
//                   time point i-1 to the worker of i, which has its stacks loaded and preprocessed by then; it registers,
//                   hands the state on and only then deconvolves.
// Results, file names and matrix files are those of the single-device run bit for bit; ProcessingLog.txt is written in
// time-point order whichever worker finishes first.  A device may be listed twice ("0,0"): two workers then share it --
// which is also what a single device number gets when the device's free memory holds two time points (else one worker;
// MIL_BATCH_WORKERS_PER_DEVICE=1: one worker, the reference's strictly sequential order).
//
// A time point stays in HBM from the TIFF reader to the TIFF writer (round 5): the stacks are uploaded once -- 16-bit files
// as 16 bits, widened to float on the device -- and resize / rotate, registration, deconvolution and the projections run on
// device pointers (include/mil_device.h: mil_imresize3d, mil_rot_y90, mil_reg3d, mil_decon_dualview, mil_max_projection,
// mil_mip3d); what comes back is what is written: the deconvolved stack (as 16 bits when that is the output depth), the
// projections, the matrix.  The host-pointer path of round 4 -- the calls the reference's main makes, ~10 PCIe trips of
// 268 MB per time point -- is kept behind MIL_BATCH_PATH=host for comparisons.  Same kernels in the same order: every output
// file is the host-pointer path's (and the reference main's) byte for byte (tests/test_gpu_cli.py).
//
// Unlike the reference the file I/O does not sit in the GPU's way: each worker reads (and converts to float) its next time
// point's two stacks on a helper thread while this one is registered and deconvolved, and writes the deconvolved stack
// behind while the projections and the next time point proceed -- at 512 x 512 x 256 the TIFF work is as long as the GPU
// work.
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <future>
#include <map>
#include <mutex>
#include <thread>
#include <utility>

#include "cli_common.h"
#include "../../include/mil_device.h"

namespace {

void usage(bool full) {
    printf("\nspimFusionBatch: Dual-view fusion (registration and joint deconvolution) for diSPIM images in batch mode\n");
    printf("\nUsage:\tspimFusionBatch [34 or 36 mandatory arguments]\n");
    if (!full) {
        printf("\nUse command for more details:\n\tspimFusionBatch -help or spimFusionBatch -h\n");
        return;
    }
    static const char *lines[] = {
        " 1: <path>      Output directory",
        " 2: <path>      Input image 1 (SPIM A) directory",
        " 3: <path>      Input image 2 (SPIM B) directory",
        " 4: <string>    Input image 1 base name",
        " 5: <string>    Input image 2 base name",
        " 6: <int>       Input image index - start",
        " 7: <int>       Input image index - end",
        " 8: <int>       Input image index - interval",
        " 9: <int>       Input image index - test (used when argument 16 is 1)",
        "10-12: <float>  Pixel size X, Y, Z of image 1 (um)",
        "13-15: <float>  Pixel size X, Y, Z of image 2 (um)",
        "16: <int>       Registration mode: 0 apply input matrix | 1 register the test image, apply to all |",
        "                2 dependently, from the previous results | 3 independently",
        "17: <int>       Image 2 rotation: 0 none | 1: 90 deg by Y-axis | -1: -90 deg by Y-axis",
        "18: <int>       Initial matrix: 0 identity | 1 input matrix file | 2 3D phase translation | 3 2D max projections",
        "19: <filename>  Input matrix file (any string unless argument 18 is 1)",
        "20: <float>     Registration tolerance",
        "21: <int>       Maximum iteration number for registration",
        "22, 23: <int>   Save registered image 1 / 2 (0 | 1)",
        "24, 25: <file>  PSF1 / PSF2 (forward projectors)",
        "26: <int>       Iteration number for deconvolution",
        "27-29: <int>    Save max projection of the decon image: X, Y, Z projection (0 | 1)",
        "30, 31: <int>   Save 3D max projection: X-axis, Y-axis (0 | 1)",
        "32: <int>       Bit of output images (16 or 32)",
        "33: <int>       Query GPU information before processing (0 | 1)",
        "34: <int>       GPU device; or a comma-separated list (\"0,1,2\": one worker thread per entry, time points dealt",
        "                round-robin; a device may be listed more than once).  A single device runs two workers when its",
        "                free memory holds two time points, else one; MIL_BATCH_WORKERS_PER_DEVICE=1..8 forces the number",
        "35, 36: <file>  (optional) backward projectors 1 / 2 [flip of the PSFs]",
    };
    printf("\tOnly 16-bit or 32-bit standard TIFF images are currently supported.\n\n");
    for (const char *l : lines) printf("\t%s\n", l);
    printf("\nArguments are set same with the ImageJ diSPIMFusion plugin.\n");
}

unsigned int scaled(unsigned int n, float num, float den) { return (unsigned int)std::lround((float)n * num / den); }

// ProcessingLog.txt + stdout.  A worker collects the text of its time point and hands it over when the time point is
// done; the texts are written in time-point order whichever worker finishes first (with one worker: as they come).
struct Log {
    std::string path;
    std::mutex mu;
    std::map<int, std::string> waiting;  // time-point index -> text, not yet written
    int next = 0;
    void emit(const std::string &text) {
        fputs(text.c_str(), stdout);
        fflush(stdout);
        if (FILE *f = fopen(path.c_str(), "a")) {
            fputs(text.c_str(), f);
            fclose(f);
        }
    }
    void direct(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        std::lock_guard<std::mutex> lock(mu);
        emit(buf);
    }
    void publish(int index, std::string text) {
        std::lock_guard<std::mutex> lock(mu);
        waiting[index] = std::move(text);
        for (auto it = waiting.find(next); it != waiting.end(); it = waiting.find(next)) {
            emit(it->second);
            waiting.erase(it);
            ++next;
        }
    }
    void flush() {  // after a failure: whatever was collected, in time-point order, gaps and all
        std::lock_guard<std::mutex> lock(mu);
        for (auto &kv : waiting) emit(kv.second);
        waiting.clear();
    }
};

// text of one time point
struct Lines {
    std::string text;
    void operator()(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        text += buf;
    }
};

std::vector<int> parse_devices(const char *arg34) {
    std::string list = arg34;
    if (list.find(',') == std::string::npos)
        if (const char *env = getenv("MIL_DEVICES")) list = env;
    std::vector<int> out;
    size_t pos = 0;
    while (pos <= list.size()) {
        const size_t comma = list.find(',', pos);
        const std::string item = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        if (!item.empty()) out.push_back(atoi(item.c_str()));
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    if (out.empty()) out.push_back(atoi(arg34));
    return out;
}

void write_tmx(const std::string &path, const float *m) {
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return;
    for (int r = 0; r < 3; ++r) fprintf(f, "%f\t%f\t%f\t%f\t\n", m[4 * r], m[4 * r + 1], m[4 * r + 2], m[4 * r + 3]);
    fprintf(f, "%f\t%f\t%f\t%f\n", 0.0, 0.0, 0.0, 1.0);
    fclose(f);
}

}  // namespace

int main(int argc, char **argv) {
    if (argc == 1) { usage(false); return EXIT_SUCCESS; }
    if (argc == 2) { usage(!strcmp(argv[1], "-h") || !strcmp(argv[1], "-help")); return EXIT_SUCCESS; }
    if (argc != 35 && argc != 37) {
        printf("Arguments do NOT match! Please input exactly 34 or 36 arguments...\nFor more information, use option -help or -h.\n");
        return 0;
    }
    const std::string out_dir = argv[1], dir1 = argv[2], dir2 = argv[3], base1 = argv[4], base2 = argv[5];
    const int n_start = atoi(argv[6]), n_end = atoi(argv[7]), n_step = atoi(argv[8]) > 0 ? atoi(argv[8]) : 1, n_test = atoi(argv[9]);
    const float px1[3] = {(float)atof(argv[10]), (float)atof(argv[11]), (float)atof(argv[12])};
    const float px2[3] = {(float)atof(argv[13]), (float)atof(argv[14]), (float)atof(argv[15])};
    int reg_mode = atoi(argv[16]);
    const int rotation = atoi(argv[17]), initial = atoi(argv[18]);
    const std::string itmx_path = argv[19];
    const float ftol = (float)atof(argv[20]);
    const int it_limit = atoi(argv[21]);
    const bool save_reg1 = atoi(argv[22]) != 0, save_reg2 = atoi(argv[23]) != 0;
    const std::string psf1_path = argv[24], psf2_path = argv[25];
    const int iters = atoi(argv[26]);
    const bool save_xproj = atoi(argv[27]) != 0, save_yproj = atoi(argv[28]) != 0, save_zproj = atoi(argv[29]) != 0;
    const bool save_3dx = atoi(argv[30]) != 0, save_3dy = atoi(argv[31]) != 0;
    const unsigned short out_bits = (unsigned short)atoi(argv[32]);
    const bool query = atoi(argv[33]) != 0;
    std::vector<int> devices = parse_devices(argv[34]);
    const int device = devices[0];
    const bool unmatch = argc == 37;
    if (query) queryDevice();

    // ---- output folders (src/spim_fusion_batch.cpp:262-296); directory arguments end with '/'
    const std::string decon_dir = out_dir + "Decon/", tmx_dir = out_dir + "TMX/", reg1_dir = out_dir + "RegA/",
                      reg2_dir = out_dir + "RegB/", mp_xy = decon_dir + "MP_ZProj/", mp_yz = decon_dir + "MP_XProj/",
                      mp_zx = decon_dir + "MP_YProj/", mp_3dx = decon_dir + "MP_3D_Xaxis/", mp_3dy = decon_dir + "MP_3D_Yaxis/";
    mkdir(out_dir.c_str(), 0755);
    mkdir(decon_dir.c_str(), 0755);
    mkdir(tmx_dir.c_str(), 0755);
    if (save_reg1) mkdir(reg1_dir.c_str(), 0755);
    if (save_reg2) mkdir(reg2_dir.c_str(), 0755);
    if (save_zproj) mkdir(mp_xy.c_str(), 0755);
    if (save_xproj) mkdir(mp_yz.c_str(), 0755);
    if (save_yproj) mkdir(mp_zx.c_str(), 0755);
    if (save_3dx) mkdir(mp_3dx.c_str(), 0755);
    if (save_3dy) mkdir(mp_3dy.c_str(), 0755);
    Log log;
    log.path = out_dir + "ProcessingLog.txt";

    // ---- geometry from the first time point (:400-452)
    auto name_of = [](const std::string &dir, const std::string &base, int n) { return dir + base + std::to_string(n) + ".tif"; };
    const int n_probe = reg_mode == 1 ? n_test : n_start;
    unsigned int in1[3], in2[3];
    const unsigned int in_bits = gettifinfo(const_cast<char *>(name_of(dir1, base1, n_probe).c_str()), in1);
    (void)gettifinfo(const_cast<char *>(name_of(dir2, base2, n_probe).c_str()), in2);
    unsigned int size[3] = {in1[0], scaled(in1[1], px1[1], px1[0]), scaled(in1[2], px1[2], px1[0])};
    unsigned int sc2[3] = {scaled(in2[0], px2[0], px1[0]), scaled(in2[1], px2[1], px1[0]), scaled(in2[2], px2[2], px1[0])};
    unsigned int size2[3] = {sc2[0], sc2[1], sc2[2]};
    int op_choice = 0;
    if (rotation == 1 || rotation == -1) {
        op_choice = rotation == 1 ? 1 : 2;
        size2[0] = sc2[2];
        size2[2] = sc2[0];
    }
    const long long sx = size[0], sy = size[1], sz = size[2];
    const size_t total = (size_t)sx * sy * sz;

    // ---- workers.  One device named once: TWO workers share it when its memory holds two time points
    // (MIL_BATCH_WORKERS_PER_DEVICE = 1 .. 8 forces the number).  A registration is a chain of ~2 000 dependent cost
    // evaluations with a host round trip between any two, and the readers / writers sit between the GPU stages of a time
    // point: a second time point in flight fills those gaps (3.05 -> 3.25 time points/s at 512 x 512 x 256,
    // profiles/r05/bench_cli_batch.json).  Results do not depend on the number of workers; ProcessingLog.txt is written in
    // time-point order either way.  What a worker keeps resident: ~12 stage volumes, the registration's border-padded copies
    // of the source (3 volumes) and target, and a dual-view engine with its OTFs on the FFT box (~10 volumes): 28 float
    // volumes is a generous bound.  A device that does not have twice that free -- a smaller part, very large stacks, a
    // device shared with another process -- runs ONE worker, like the reference's sequential main, instead of ending the
    // batch when the second worker's allocation fails.
    if (devices.size() == 1) {
        const char *per = getenv("MIL_BATCH_WORKERS_PER_DEVICE");
        int n_workers = per ? atoi(per) : 2;
        if (!per) {
            const size_t per_worker = 28 * total * sizeof(float);
            size_t free_b = 0, total_b = 0;
            if (mil_dev_mem_info(devices[0], &free_b, &total_b) != 0 || free_b < 2 * per_worker) n_workers = 1;
        }
        for (int i = 1; i < n_workers && i < 8; ++i) devices.push_back(devices[0]);
    }

    cli::Volume psf1 = cli::load_tiff(psf1_path), psf2 = cli::load_tiff(psf2_path), bp1, bp2;
    if (!cli::same_size(psf1, psf2)) { printf("\tThe two PSFs don't have the same image size, processing stopped !!!\n"); return 1; }
    if (unmatch) { bp1 = cli::load_tiff(argv[35]); bp2 = cli::load_tiff(argv[36]); }
    else { bp1.data.resize(psf1.count()); bp2.data.resize(psf1.count()); }

    // ---- registration set-up (:530-561): initial matrix choice -> regChoice; the batch hard-codes affMethod 6
    int reg_choice = 2;
    bool have_itmx = false;
    switch (initial) {
    case 1: have_itmx = true; break;
    case 2: reg_choice = 3; break;
    case 3: reg_choice = 4; break;
    default: break;
    }
    if (reg_mode < 0 || reg_mode > 3) { printf("\tWrong registration mode %d, processing stopped !!!\n", reg_mode); return 1; }
    const int aff_method = 6;
    const float cost_bar = 0.1f;
    float tmx0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    if (have_itmx) {
        FILE *f = fopen(itmx_path.c_str(), "r");
        if (!f) { printf("***** Input transformation matrix file does not exist: %s\n", itmx_path.c_str()); return 1; }
        for (int i = 0; i < 12; ++i)
            if (fscanf(f, "%f", &tmx0[i]) != 1) { printf("***** Malformed matrix file %s\n", itmx_path.c_str()); fclose(f); return 1; }
        fclose(f);
    }

    const auto t_batch = std::chrono::steady_clock::now();
    auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    log.direct("\n=== spimFusionBatch: images %d .. %d (step %d), size %lld x %lld x %lld, regMode %d, %d decon iterations\n",
               n_start, n_end, n_step, sx, sy, sz, reg_mode, iters);
    if (devices.size() == 1) {
        log.direct("=== 1 worker on GPU device %d\n", devices[0]);
    } else {
        std::string list;
        for (size_t i = 0; i < devices.size(); ++i) list += (i ? "," : "") + std::to_string(devices[i]);
        log.direct("=== %zu workers on GPU devices %s, time points dealt round-robin\n", devices.size(), list.c_str());
    }

    typedef std::pair<cli::Volume, cli::Volume> Pair;
    auto load_pair = [&](int n) { return Pair(cli::load_tiff(name_of(dir1, base1, n)), cli::load_tiff(name_of(dir2, base2, n))); };

    // What one registration needs besides the volumes; the fall-back ladder of the reference (:714-843).
    struct RegState {
        float tmx[12], initial_tmx[12];
        bool have_itmx;
    };
    // Preprocessing of one time point (:666-711).  No copies: a stack that needs no resampling / rotation is used where
    // it was read.
    struct Prepared {
        Pair in;
        cli::Volume a_res, b_rot_own, b_res;
        cli::Volume *a = nullptr, *b = nullptr;
    };
    // index of the first time point that failed (INT_MAX: none).  The batch stops WHERE THE REFERENCE STOPS: time points before
    // it are completed by whichever worker owns them, time points after it are not started.  One in flight when the failure
    // happens is abandoned at its next stage boundary -- before its matrix file if it is still registering, before its
    // deconvolved stack if it is deconvolving; a later time point that had already written its matrix (or, past the
    // deconvolution, its stack and projections) keeps what it wrote.
    std::atomic<int> fail_at{INT_MAX};
    auto preprocess = [&](Prepared &p, int dev) -> bool {
        cli::Volume &a_in = p.in.first, &b_in = p.in.second;
        if (a_in.size[0] != in1[0] || a_in.size[1] != in1[1] || a_in.size[2] != in1[2]) { printf("\t Input image 1 size does not match !!!\n"); return false; }
        if (b_in.size[0] != in2[0] || b_in.size[1] != in2[1] || b_in.size[2] != in2[2]) { printf("\t Input image 2 size does not match !!!\n"); return false; }
        p.a = &a_in;
        if (a_in.size[0] != size[0] || a_in.size[1] != size[1] || a_in.size[2] != size[2]) {
            p.a_res.bits = a_in.bits;
            p.a_res.data.resize(total);
            for (int i = 0; i < 3; ++i) p.a_res.size[i] = size[i];
            (void)imresize3d(p.a_res.data.data(), a_in.data.data(), sx, sy, sz, a_in.size[0], a_in.size[1], a_in.size[2], dev);
            p.a = &p.a_res;
        }
        cli::Volume *b_rot_ptr = &b_in;
        if (op_choice != 0) {
            p.b_rot_own.bits = b_in.bits;
            p.b_rot_own.data.resize(b_in.count());
            (void)imoperation3D(p.b_rot_own.data.data(), p.b_rot_own.size, b_in.data.data(), b_in.size, op_choice, dev);
            b_rot_ptr = &p.b_rot_own;
        }
        cli::Volume &b_rot = *b_rot_ptr;
        p.b = &b_rot;
        if (b_rot.size[0] != size2[0] || b_rot.size[1] != size2[1] || b_rot.size[2] != size2[2]) {
            p.b_res.bits = b_rot.bits;
            p.b_res.data.resize((size_t)size2[0] * size2[1] * size2[2]);
            for (int i = 0; i < 3; ++i) p.b_res.size[i] = size2[i];
            (void)imresize3d(p.b_res.data.data(), b_rot.data.data(), size2[0], size2[1], size2[2], b_rot.size[0], b_rot.size[1],
                             b_rot.size[2], dev);
            p.b = &p.b_res;
        }
        return true;
    };
    // One registration attempt = reg(tmx in / out, rec, regChoice, use the input matrix) -> status; the ladder around it.
    typedef std::function<int(float *, float *, int, bool)> RegCall;
    auto host_reg = [&](Prepared &p, std::vector<float> &b_reg, int dev) -> RegCall {
        return [&p, &b_reg, dev, aff_method, ftol, it_limit](float *tmx, float *rec, int choice, bool use_input) {
            return reg3d(b_reg.data(), tmx, p.a->data.data(), p.b->data.data(), p.a->size, p.b->size, choice, aff_method, use_input,
                         ftol, it_limit, dev, -1, true, rec);
        };
    };
    auto bad = [&](const RegState &st, const float *rec) {
        return !checkmatrix(const_cast<float *>(st.tmx), sx, sy, sz) || rec[3] < cost_bar;
    };
    // Returns the status of the library call that failed (0: all calls succeeded).  A failed call leaves tmx / b_reg / rec
    // untouched, so nothing below may use them: the caller stops the batch.
    auto ladder = [&](RegState &st, const RegCall &reg, float *rec, int choice_now, Lines &out) -> int {
        // first attempt, other scheme without the input matrix, then the input matrix itself
        if (int status = reg(st.tmx, rec, choice_now, st.have_itmx)) return status;
        if (bad(st, rec)) {
            out("\n\t... Attempt failed: transformation matrix problematic or cost function value %f < threshold %2.2f\n"
                "\n\t... Change scheme and redo the registration!!!\n", rec[3], cost_bar);
            if (int status = reg(st.tmx, rec, choice_now == 4 ? 2 : 4, false)) return status;
            if (bad(st, rec) && st.have_itmx) {
                out("\n\t... Attempt failed again: use input transformation matrix!!!\n");
                std::memcpy(st.tmx, st.initial_tmx, sizeof st.tmx);
                if (int status = reg(st.tmx, rec, 0, true)) return status;
            }
        }
        return 0;
    };

    // ---- regMode 1: register on the test image first (on the first device), then apply that matrix to every image
    // (:719-751).  Nothing is written for this pass.
    RegState base;
    std::memcpy(base.tmx, tmx0, sizeof tmx0);
    std::memcpy(base.initial_tmx, tmx0, sizeof tmx0);
    base.have_itmx = have_itmx;
    if (reg_mode == 1) {
        Lines out;
        out("\n*** Image time point number: %d \n", n_test);
        Prepared p;
        p.in = load_pair(n_test);
        if (!preprocess(p, device)) return 1;
        std::vector<float> b_reg(total);
        float rec[11] = {0};
        const int status = ladder(base, host_reg(p, b_reg, device), rec, reg_choice, out);
        log.direct("%s", out.text.c_str());
        if (status != 0) { printf("\t... registration of the test image failed (status %d), processing stopped !!!\n", status); return 1; }
        reg_mode = 0;
        base.have_itmx = true;
        std::memcpy(base.initial_tmx, base.tmx, sizeof base.tmx);
    }

    std::vector<int> points;
    for (int n = n_start; n <= n_end; n += n_step) points.push_back(n);

    // ---- regMode 2: the chain state travels from the worker of time point i-1 to the worker of i
    struct Chain {
        std::mutex mu;
        std::condition_variable cv;
        int next = 0;  // index of the time point whose registration may start
        float weighted[12], previous[12];
        bool have_itmx;
    } chain;
    std::memcpy(chain.weighted, tmx0, sizeof tmx0);
    std::memcpy(chain.previous, tmx0, sizeof tmx0);
    chain.have_itmx = have_itmx;

    // index of the first time point that failed is fail_at; a failed worker must not leave the others waiting for the chain
    auto stop = [&](int index) {
        int seen = fail_at.load();
        while (index < seen && !fail_at.compare_exchange_weak(seen, index)) {}
        { std::lock_guard<std::mutex> lock(chain.mu); }  // a waiter is either before its predicate test or asleep
        chain.cv.notify_all();
    };
    // A time point given up because an EARLIER one failed: its text so far plus one line goes to the log, so that every index a
    // worker took up has an entry (the ordered log never holds later texts back for an index nobody will publish).
    auto abandoned = [&](size_t i, Lines &out) {
        out("\t... abandoned: an earlier time point failed, nothing more of this one is written\n");
        log.publish((int)i, std::move(out.text));
    };
    // The registration of time point i according to the mode (:714-843), through `reg` (host or device pointers).
    // 0: registered; kAbandoned: an earlier time point failed, nothing was done; anything else: the status of the library
    // call that failed (libapi.h returns 1 or -1).
    constexpr int kAbandoned = INT_MIN;
    auto register_step = [&](size_t i, RegState &st, const RegCall &reg, float *rec, Lines &out) -> int {
        int reg_status = 0;
        switch (reg_mode) {
        case 0: reg_status = reg(st.tmx, rec, 0, st.have_itmx); break;
        case 2: {
            std::unique_lock<std::mutex> lock(chain.mu);
            chain.cv.wait(lock, [&] { return fail_at.load() < (int)i || chain.next >= (int)i; });
            if (fail_at.load() < (int)i || chain.next != (int)i) return kAbandoned;  // an earlier time point failed: the chain ends there
            lock.unlock();  // the state is this worker's alone until it advances `next`
            if (i == 0) {
                reg_status = ladder(st, reg, rec, reg_choice, out);
                if (reg_status == 0) std::memcpy(chain.weighted, st.tmx, sizeof st.tmx);
            } else {  // start from the running average, fall back to the previous matrix (:786-808)
                std::memcpy(st.tmx, chain.weighted, sizeof st.tmx);
                reg_status = reg(st.tmx, rec, 2, true);
                if (reg_status == 0 && bad(st, rec)) {
                    out("\n\t... Attempt failed: use previous transformation matrix!!!\n");
                    std::memcpy(st.tmx, chain.previous, sizeof st.tmx);
                    reg_status = reg(st.tmx, rec, 0, true);
                }
                if (reg_status == 0)
                    for (int j = 0; j < 12; ++j) chain.weighted[j] = (float)(0.8 * chain.weighted[j] + 0.2 * st.tmx[j]);  // double arithmetic, :805-807
            }
            if (reg_status != 0) break;  // the chain does NOT advance from a failed registration: stop() wakes the others
            std::memcpy(chain.previous, st.tmx, sizeof st.tmx);
            lock.lock();
            chain.next = (int)i + 1;
            lock.unlock();
            chain.cv.notify_all();
            break;
        }
        case 3: reg_status = ladder(st, reg, rec, reg_choice, out); break;
        }
        return reg_status;
    };

    auto host_worker = [&](int w) {
        const int dev = devices[(size_t)w];
        const int n_workers = (int)devices.size();
        std::vector<float> b_reg(total), decon(total), decon_out(total), mp((size_t)(sx * sy + sy * sz + sz * sx));
        float reg_records[11] = {0}, decon_records[10] = {0};
        std::future<Pair> ahead;
        size_t ahead_i = 0;
        std::future<void> behind;
        for (size_t i = (size_t)w; i < points.size() && (int)i < fail_at.load(); i += (size_t)n_workers) {
            const int n = points[i];
            Lines out;
            out("\n*** Image time point number: %d \n", n);
            const auto t0 = std::chrono::steady_clock::now();

            // ---- stacks: read ahead by this worker's previous time point, or read now
            Prepared p;
            if (ahead.valid() && ahead_i == i) {
                p.in = ahead.get();
            } else {
                // A missing stack ends the batch HERE, in order: the library's readers exit() the process on an unreadable file
                // (like the reference's), which from one of several worker threads would cut the other workers' deconvolutions
                // and half-written stacks off and drop the buffered log.  Checked first, the failure goes through stop(): the
                // others finish their time point, the log is flushed, main returns 1.
                const std::string fa = name_of(dir1, base1, n), fb = name_of(dir2, base2, n);
                const bool have_a = fexists(const_cast<char *>(fa.c_str())), have_b = fexists(const_cast<char *>(fb.c_str()));
                if (!have_a || !have_b) {
                    out("\t... input stack %s does not exist, processing stopped !!!\n", (have_a ? fb : fa).c_str());
                    fprintf(stderr, "spimFusionBatch: %s does not exist\n", (have_a ? fb : fa).c_str());
                    log.publish((int)i, std::move(out.text));
                    stop((int)i);
                    break;
                }
                p.in = load_pair(n);
            }
            {   // this worker's next time point: only when both files are there -- a missing file must stop the batch where
                // the reference stops it, after the earlier time points' results are written
                const size_t ni = i + (size_t)n_workers;
                if (ni < points.size() && fexists(const_cast<char *>(name_of(dir1, base1, points[ni]).c_str())) &&
                    fexists(const_cast<char *>(name_of(dir2, base2, points[ni]).c_str()))) {
                    ahead_i = ni;
                    ahead = std::async(std::launch::async, load_pair, points[ni]);
                }
            }
            if (!preprocess(p, dev)) { out("\t Input image size does not match !!!\n"); log.publish((int)i, std::move(out.text)); stop((int)i); break; }
            cli::Volume &a = *p.a;
            const auto t1 = std::chrono::steady_clock::now();

            // ---- registration (:714-843)
            RegState st = base;
            const int reg_status = register_step(i, st, host_reg(p, b_reg, dev), reg_records, out);
            if (reg_status == kAbandoned) { abandoned(i, out); break; }
            if (reg_status != 0) {  // a library call failed (its message is on stderr): never write results made from stale buffers
                out("\t... registration failed (status %d), processing stopped !!!\n", reg_status);
                log.publish((int)i, std::move(out.text));
                stop((int)i);
                break;
            }
            if (fail_at.load() < (int)i) { abandoned(i, out); break; }  // an earlier time point failed while this one registered: write nothing
            out("\t... initial cost function value: %f\n\t... final cost function value: %f\n\t... registration time cost: %2.3f s\n",
                reg_records[1], reg_records[3], reg_records[7]);
            const std::string ns = std::to_string(n);
            write_tmx(tmx_dir + "Matrix_" + ns + ".tmx", st.tmx);  // always saved (:859-868)
            if (save_reg1) writetifstack(const_cast<char *>((reg1_dir + base1 + "reg_" + ns + ".tif").c_str()), a.data.data(), size, (unsigned short)in_bits);
            if (save_reg2) writetifstack(const_cast<char *>((reg2_dir + base2 + "reg_" + ns + ".tif").c_str()), b_reg.data(), size, (unsigned short)in_bits);
            const auto t2 = std::chrono::steady_clock::now();

            // ---- joint deconvolution (:874-911)
            if (const int status = decon_dualview(decon.data(), a.data.data(), b_reg.data(), size, psf1.data.data(), psf2.data.data(),
                                                  psf1.size, false, iters, dev, -1, true, decon_records, unmatch, bp1.data.data(),
                                                  bp2.data.data())) {
                out("\t... deconvolution failed (status %d), processing stopped !!!\n", status);
                log.publish((int)i, std::move(out.text));
                stop((int)i);
                break;
            }
            if (fail_at.load() < (int)i) { abandoned(i, out); break; }  // an earlier time point failed meanwhile: nothing more of this one is written
            out("\t... Deconvolution time cost: %2.3f s\n\t... GPU free memory after deconvolution is %.0f MBites\n", decon_records[9],
                decon_records[5]);
            if (behind.valid()) behind.get();  // the previous stack has left decon_out
            decon_out.swap(decon);             // projections below read decon_out; decon is scratch again
            {
                const std::string path = decon_dir + "Decon_" + ns + ".tif";
                const float *data = decon_out.data();
                unsigned int wsize[3] = {size[0], size[1], size[2]};
                const unsigned short bits = out_bits;
                behind = std::async(std::launch::async, [path, data, wsize, bits]() mutable {
                    writetifstack(const_cast<char *>(path.c_str()), const_cast<float *>(data), wsize, bits);
                });
            }
            const auto t3 = std::chrono::steady_clock::now();

            // ---- projections (:913-943)
            if (save_zproj || save_xproj || save_yproj) {
                unsigned int mp_size[6];
                (void)mp2dgpu(mp.data(), mp_size, decon_out.data(), size, save_zproj, save_xproj, save_yproj);
                unsigned int s2[3] = {0, 0, 1};
                if (save_zproj) { s2[0] = mp_size[0]; s2[1] = mp_size[1]; writetifstack(const_cast<char *>((mp_xy + "MP_XY_" + ns + ".tif").c_str()), mp.data(), s2, out_bits); }
                if (save_xproj) { s2[0] = mp_size[2]; s2[1] = mp_size[3]; writetifstack(const_cast<char *>((mp_yz + "MP_YZ_" + ns + ".tif").c_str()), mp.data() + sx * sy, s2, out_bits); }
                if (save_yproj) { s2[0] = mp_size[4]; s2[1] = mp_size[5]; writetifstack(const_cast<char *>((mp_zx + "MP_ZX_" + ns + ".tif").c_str()), mp.data() + sx * sy + sy * sz, s2, out_bits); }
            }
            for (int axis = 1; axis <= 2; ++axis) {
                if (!(axis == 1 ? save_3dx : save_3dy)) continue;
                const long long project_num = 36;
                const long long rot = (long long)std::llround(std::sqrt(axis == 1 ? (double)(sy * sy + sz * sz) : (double)(sx * sx + sz * sz)));
                std::vector<float> mp3d((size_t)((axis == 1 ? sx : sy) * rot * project_num));
                unsigned int s3[3];
                (void)mip3dgpu(mp3d.data(), s3, decon_out.data(), size, axis, project_num);
                writetifstack(const_cast<char *>(((axis == 1 ? mp_3dx + "MP_3D_Xaxis_" : mp_3dy + "MP_3D_Yaxis_") + ns + ".tif").c_str()),
                              mp3d.data(), s3, out_bits);
            }
            const auto t4 = std::chrono::steady_clock::now();
            out("\tTime cost: preprocessing %2.3f s, registration %2.3f s, deconvolution %2.3f s, projections %2.3f s\n", sec(t0, t1),
                sec(t1, t2), sec(t2, t3), sec(t3, t4));
            log.publish((int)i, std::move(out.text));
        }
        if (ahead.valid()) ahead.wait();
        if (behind.valid()) behind.get();
    };

    // ---- the same time point with everything in HBM between the readers and the writers --------------------------------
    struct DevBuf {  // device memory from the library's pool
        void *p = nullptr;
        size_t bytes = 0;
        ~DevBuf() { if (p) mil_dev_free(p); }
        bool need(size_t b) {
            if (b <= bytes) return true;
            if (p) mil_dev_free(p);
            p = nullptr;
            bytes = 0;
            if (mil_dev_alloc(&p, b) != 0) return false;
            bytes = b;
            return true;
        }
        float *f() const { return (float *)p; }
    };
    struct PinBuf {  // pinned host memory: what the TIFF codec reads into and writes from
        void *p = nullptr;
        size_t bytes = 0;
        ~PinBuf() { if (p) mil_host_free(p); }
        bool need(size_t b) {
            if (b <= bytes) return true;
            if (p) mil_host_free(p);
            p = nullptr;
            bytes = 0;
            if (mil_host_alloc(&p, b) != 0) return false;
            bytes = b;
            return true;
        }
    };
    struct RawStack {  // one stack as the file holds it: 16-bit samples stay 16 bits until they are on the device
        PinBuf host;
        unsigned int size[3] = {0, 0, 0};
        unsigned int bits = 0;
        size_t count() const { return (size_t)size[0] * size[1] * size[2]; }
    };
    auto read_raw = [&](const std::string &path, RawStack &r) {
        r.bits = gettifinfo(const_cast<char *>(path.c_str()), r.size);  // exits on a missing file, like the reference
        if (!r.host.need(r.count() * (r.bits == 16 ? 2 : 4))) { fprintf(stderr, "spimFusionBatch: no pinned memory for %s\n", path.c_str()); exit(1); }
        if (r.bits == 16) readtifstack_16to16((unsigned short *)r.host.p, const_cast<char *>(path.c_str()), r.size);
        else readtifstack((float *)r.host.p, const_cast<char *>(path.c_str()), r.size);
    };

    auto device_worker = [&](int w) {
        const int dev = devices[(size_t)w];
        const int n_workers = (int)devices.size();
        if (mil_set_device(dev) != 0) { stop(w); return; }
        void *stream = nullptr;
        if (mil_stream_create(&stream) != 0) { stop(w); return; }
        RawStack raw[2][2];  // [slot][view]: the slot of time point k is k's turn modulo 2 -- one is read while the other is used
        DevBuf d_raw_a, d_raw_b, d_in_a, d_in_b, d_a_res, d_b_rot, d_b_res, d_b_reg, d_decon, d_out16, d_mp, d_mip;
        PinBuf h_out;
        std::vector<float> mp((size_t)(sx * sy + sy * sz + sz * sx)), scratch;
        float reg_records[11] = {0}, decon_records[10] = {0};
        std::future<void> ahead;
        size_t ahead_i = 0;
        int turn = 0;
        std::future<void> behind;
        bool fatal = false;
        auto ok = [&](int status, const char *what, size_t i, Lines &out) {  // a library call on the device path
            if (status == 0) return true;
            out("\t... %s failed (status %d), processing stopped !!!\n", what, status);
            log.publish((int)i, std::move(out.text));
            stop((int)i);
            fatal = true;
            return false;
        };
        for (size_t i = (size_t)w; i < points.size() && (int)i < fail_at.load() && !fatal; i += (size_t)n_workers, turn ^= 1) {
            const int n = points[i];
            Lines out;
            out("\n*** Image time point number: %d \n", n);
            const auto t0 = std::chrono::steady_clock::now();
            RawStack &ra = raw[turn][0], &rb = raw[turn][1];
            if (ahead.valid() && ahead_i == i) {
                ahead.get();
            } else {
                const std::string fa = name_of(dir1, base1, n), fb = name_of(dir2, base2, n);
                const bool have_a = fexists(const_cast<char *>(fa.c_str())), have_b = fexists(const_cast<char *>(fb.c_str()));
                if (!have_a || !have_b) {  // ends the batch HERE, in order (see the host-pointer worker)
                    out("\t... input stack %s does not exist, processing stopped !!!\n", (have_a ? fb : fa).c_str());
                    fprintf(stderr, "spimFusionBatch: %s does not exist\n", (have_a ? fb : fa).c_str());
                    log.publish((int)i, std::move(out.text));
                    stop((int)i);
                    break;
                }
                read_raw(fa, ra);
                read_raw(fb, rb);
            }
            {   // this worker's next time point, into the other slot, while this one is on the GPU
                const size_t ni = i + (size_t)n_workers;
                if (ni < points.size() && fexists(const_cast<char *>(name_of(dir1, base1, points[ni]).c_str())) &&
                    fexists(const_cast<char *>(name_of(dir2, base2, points[ni]).c_str()))) {
                    ahead_i = ni;
                    RawStack *na = &raw[turn ^ 1][0], *nb = &raw[turn ^ 1][1];
                    const std::string fa = name_of(dir1, base1, points[ni]), fb = name_of(dir2, base2, points[ni]);
                    ahead = std::async(std::launch::async, [&read_raw, na, nb, fa, fb] { read_raw(fa, *na); read_raw(fb, *nb); });
                }
            }
            if (ra.size[0] != in1[0] || ra.size[1] != in1[1] || ra.size[2] != in1[2]) { printf("\t Input image 1 size does not match !!!\n"); out("\t Input image 1 size does not match !!!\n"); log.publish((int)i, std::move(out.text)); stop((int)i); break; }
            if (rb.size[0] != in2[0] || rb.size[1] != in2[1] || rb.size[2] != in2[2]) { printf("\t Input image 2 size does not match !!!\n"); out("\t Input image 2 size does not match !!!\n"); log.publish((int)i, std::move(out.text)); stop((int)i); break; }

            // ---- upload (16-bit files as 16 bits) and widen on the device: (float)v like readtifstack (src/apifunc.cpp:166-183)
            auto to_device = [&](RawStack &r, DevBuf &d_raw, DevBuf &d_in) -> float * {
                const size_t cnt = r.count();
                if (r.bits == 16) {
                    if (!d_raw.need(cnt * 2) || !d_in.need(cnt * 4)) return nullptr;
                    if (mil_upload(d_raw.p, r.host.p, cnt * 2, stream) != 0) return nullptr;
                    if (mil_u16_to_f32(d_in.f(), (const unsigned short *)d_raw.p, cnt, stream) != 0) return nullptr;
                    return d_in.f();
                }
                if (!d_in.need(cnt * 4)) return nullptr;
                if (r.bits == 32 && mil_upload(d_in.p, r.host.p, cnt * 4, stream) != 0) return nullptr;
                return d_in.f();  // other depths: the reader left the buffer alone, like the reference's
            };
            float *d_a = to_device(ra, d_raw_a, d_in_a), *d_b = to_device(rb, d_raw_b, d_in_b);
            if (!ok(d_a && d_b ? 0 : -1, "upload of the input stacks", i, out)) break;
            // ---- preprocessing (:666-711) on device pointers
            if (in1[0] != size[0] || in1[1] != size[1] || in1[2] != size[2]) {
                if (!ok(d_a_res.need(total * 4) ? mil_imresize3d(d_a_res.f(), d_a, sx, sy, sz, in1[0], in1[1], in1[2], stream) : -1, "imresize3d", i, out)) break;
                d_a = d_a_res.f();
            }
            unsigned int rot_size[3] = {in2[0], in2[1], in2[2]};
            if (op_choice != 0) {
                if (!ok(d_b_rot.need(rb.count() * 4) ? mil_rot_y90(d_b_rot.f(), d_b, (int)in2[0], (int)in2[1], (int)in2[2], op_choice == 1 ? 1 : -1, stream) : -1,
                        "imoperation3D", i, out)) break;
                d_b = d_b_rot.f();
                rot_size[0] = in2[2];
                rot_size[2] = in2[0];
            }
            if (rot_size[0] != size2[0] || rot_size[1] != size2[1] || rot_size[2] != size2[2]) {
                if (!ok(d_b_res.need((size_t)size2[0] * size2[1] * size2[2] * 4)
                            ? mil_imresize3d(d_b_res.f(), d_b, size2[0], size2[1], size2[2], rot_size[0], rot_size[1], rot_size[2], stream) : -1,
                        "imresize3d", i, out)) break;
                d_b = d_b_res.f();
            }
            if (!ok(d_b_reg.need(total * 4) && d_decon.need(total * 4) ? mil_stream_sync(stream) : -1, "preprocessing", i, out)) break;
            const auto t1 = std::chrono::steady_clock::now();

            // ---- registration (:714-843)
            RegState st = base;
            const float *reg_a = d_a, *reg_b = d_b;
            const RegCall dev_reg = [&, reg_a, reg_b](float *tmx, float *rec, int choice, bool use_input) {
                return mil_reg3d(d_b_reg.f(), tmx, reg_a, reg_b, size, size2, choice, aff_method, use_input ? 1 : 0, ftol, it_limit, 1, rec, stream);
            };
            const int reg_status = register_step(i, st, dev_reg, reg_records, out);
            if (reg_status == kAbandoned) { abandoned(i, out); break; }
            if (!ok(reg_status, "registration", i, out)) break;
            if (fail_at.load() < (int)i) { abandoned(i, out); break; }  // an earlier time point failed while this one registered: write nothing
            out("\t... initial cost function value: %f\n\t... final cost function value: %f\n\t... registration time cost: %2.3f s\n",
                reg_records[1], reg_records[3], reg_records[7]);
            const std::string ns = std::to_string(n);
            write_tmx(tmx_dir + "Matrix_" + ns + ".tmx", st.tmx);  // always saved (:859-868)
            auto save_volume = [&](const std::string &path, const float *d_vol) {
                scratch.resize(total);
                if (mil_download(scratch.data(), d_vol, total * 4, stream) != 0 || mil_stream_sync(stream) != 0) return false;
                writetifstack(const_cast<char *>(path.c_str()), scratch.data(), size, (unsigned short)in_bits);
                return true;
            };
            if (save_reg1 && !ok(save_volume(reg1_dir + base1 + "reg_" + ns + ".tif", d_a) ? 0 : -1, "download", i, out)) break;
            if (save_reg2 && !ok(save_volume(reg2_dir + base2 + "reg_" + ns + ".tif", d_b_reg.f()) ? 0 : -1, "download", i, out)) break;
            const auto t2 = std::chrono::steady_clock::now();

            // ---- joint deconvolution (:874-911)
            if (!ok(mil_decon_dualview(d_decon.f(), d_a, d_b_reg.f(), size, psf1.data.data(), psf2.data.data(), psf1.size, 0, iters, 1,
                                       decon_records, unmatch ? 1 : 0, bp1.data.data(), bp2.data.data(), stream),
                    "deconvolution", i, out)) break;
            if (fail_at.load() < (int)i) { abandoned(i, out); break; }  // an earlier time point failed meanwhile: nothing of this one is written
            out("\t... Deconvolution time cost: %2.3f s\n\t... GPU free memory after deconvolution is %.0f MBites\n", decon_records[9],
                decon_records[5]);
            if (behind.valid()) behind.get();  // the previous stack has left h_out
            {
                const std::string path = decon_dir + "Decon_" + ns + ".tif";
                const bool as16 = out_bits == 16;
                int status = h_out.need(total * (as16 ? 2 : 4)) ? 0 : -1;
                if (status == 0 && as16)  // (unsigned short)f on the device, as writetifstack does on the host (:255)
                    status = d_out16.need(total * 2) ? mil_f32_to_u16((unsigned short *)d_out16.p, d_decon.f(), total, stream) : -1;
                if (status == 0) status = mil_download(h_out.p, as16 ? d_out16.p : d_decon.p, total * (as16 ? 2 : 4), stream);
                if (status == 0) status = mil_stream_sync(stream);
                if (!ok(status, "download of the deconvolved stack", i, out)) break;
                void *data = h_out.p;
                unsigned int wsize[3] = {size[0], size[1], size[2]};
                const unsigned short bits = out_bits;
                behind = std::async(std::launch::async, [path, data, wsize, bits, as16]() mutable {
                    if (as16) writetifstack_16to16(const_cast<char *>(path.c_str()), (unsigned short *)data, wsize);
                    else writetifstack(const_cast<char *>(path.c_str()), (float *)data, wsize, bits);
                });
            }
            const auto t3 = std::chrono::steady_clock::now();

            // ---- projections (:913-943) of the stack in HBM
            if (save_zproj || save_xproj || save_yproj) {  // mp2dgpu (src/apifunc.cpp:485-505; the Y projection hangs on the Z flag there)
                const size_t n_mp = mp.size();
                int status = d_mp.need(n_mp * 4) ? 0 : -1;
                std::fill(mp.begin(), mp.end(), 0.0f);
                if (status == 0) status = mil_upload(d_mp.p, mp.data(), n_mp * 4, stream);
                if (status == 0 && save_zproj) status = mil_max_projection(d_mp.f(), d_decon.f(), (int)sx, (int)sy, (int)sz, 1, stream);
                if (status == 0 && save_xproj) status = mil_max_projection(d_mp.f() + sx * sy, d_decon.f(), (int)sx, (int)sy, (int)sz, 3, stream);
                if (status == 0 && save_zproj) status = mil_max_projection(d_mp.f() + sx * sy + sy * sz, d_decon.f(), (int)sx, (int)sy, (int)sz, 2, stream);
                if (status == 0) status = mil_download(mp.data(), d_mp.p, n_mp * 4, stream);
                if (status == 0) status = mil_stream_sync(stream);
                if (!ok(status, "max projections", i, out)) break;
                unsigned int s2[3] = {0, 0, 1};
                if (save_zproj) { s2[0] = (unsigned)sx; s2[1] = (unsigned)sy; writetifstack(const_cast<char *>((mp_xy + "MP_XY_" + ns + ".tif").c_str()), mp.data(), s2, out_bits); }
                if (save_xproj) { s2[0] = (unsigned)sy; s2[1] = (unsigned)sz; writetifstack(const_cast<char *>((mp_yz + "MP_YZ_" + ns + ".tif").c_str()), mp.data() + sx * sy, s2, out_bits); }
                if (save_yproj) { s2[0] = (unsigned)sz; s2[1] = (unsigned)sx; writetifstack(const_cast<char *>((mp_zx + "MP_ZX_" + ns + ".tif").c_str()), mp.data() + sx * sy + sy * sz, s2, out_bits); }
            }
            for (int axis = 1; axis <= 2 && !fatal; ++axis) {
                if (!(axis == 1 ? save_3dx : save_3dy)) continue;
                const long long project_num = 36;
                int s3i[3] = {0, 0, 0};
                int status = mil_mip3d(nullptr, d_decon.f(), (int)sx, (int)sy, (int)sz, axis, project_num, s3i, stream);
                const size_t cnt = (size_t)s3i[0] * s3i[1] * s3i[2];
                if (status == 0) status = d_mip.need(cnt * 4) ? mil_mip3d(d_mip.f(), d_decon.f(), (int)sx, (int)sy, (int)sz, axis, project_num, s3i, stream) : -1;
                scratch.resize(cnt);
                if (status == 0) status = mil_download(scratch.data(), d_mip.p, cnt * 4, stream);
                if (status == 0) status = mil_stream_sync(stream);
                if (!ok(status, "3-D max projections", i, out)) break;
                unsigned int s3[3] = {(unsigned)s3i[0], (unsigned)s3i[1], (unsigned)s3i[2]};
                writetifstack(const_cast<char *>(((axis == 1 ? mp_3dx + "MP_3D_Xaxis_" : mp_3dy + "MP_3D_Yaxis_") + ns + ".tif").c_str()),
                              scratch.data(), s3, out_bits);
            }
            if (fatal) break;
            const auto t4 = std::chrono::steady_clock::now();
            out("\tTime cost: preprocessing %2.3f s, registration %2.3f s, deconvolution %2.3f s, projections %2.3f s\n", sec(t0, t1),
                sec(t1, t2), sec(t2, t3), sec(t3, t4));
            log.publish((int)i, std::move(out.text));
        }
        if (ahead.valid()) ahead.wait();
        if (behind.valid()) behind.get();
        (void)mil_stream_sync(stream);
        mil_stream_destroy(stream);
    };

    const char *path_env = getenv("MIL_BATCH_PATH");
    const bool host_path = path_env && !strcmp(path_env, "host");
    auto worker = [&](int w) {
        if (host_path) host_worker(w);
        else device_worker(w);
    };

    if (devices.size() == 1) {
        worker(0);
    } else {
        std::vector<std::thread> threads;
        for (int w = 0; w < (int)devices.size(); ++w) threads.emplace_back(worker, w);
        for (auto &t : threads) t.join();
    }
    if (fail_at.load() != INT_MAX) { log.flush(); return 1; }
    log.direct("\n=== Batch processing completed, whole time cost: %2.3f s\n", sec(t_batch, std::chrono::steady_clock::now()));
    return 0;
}
