// The gpuMemMode 0 host path (csrc/cpu_fft.cpp, csrc/cpu_decon.cpp) under AddressSanitizer + UBSan: ragged boxes (pad /
// crop), odd radices, a PSF larger than the box, dual view, several thread counts.  Built and run by
// tests/test_host_sanitize.py; exit code 0 = clean and self-consistent (threads do not change a bit).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../microimagelib_amd/csrc/cpu_decon.h"
#include "../../microimagelib_amd/csrc/cpu_fft.h"

using namespace mil;

static int snap(int n) {  // snapTransformSize, src/api_subfunc.cu:57-87
    n = (n % 16) ? n - n % 16 + 16 : n;
    int hi = 31;
    while (!(n & (1u << hi))) --hi;
    if ((1 << hi) == n) return n;
    if ((2 << hi) <= 128) return 2 << hi;
    return (n % 64) ? n - n % 64 + 64 : n;
}

int main() {
    const int cases[][6] = {{20, 12, 9, 5, 5, 5}, {50, 36, 20, 9, 7, 3}, {16, 16, 16, 21, 5, 5}, {70, 18, 200, 7, 7, 7}};
    for (auto &c : cases) {
        const Dims im{c[0], c[1], c[2]}, psf{c[3], c[4], c[5]};
        const Dims box{snap(c[0]), snap(c[1]), snap(c[2])};
        std::vector<float> a((size_t)im.count()), b(a.size()), p((size_t)psf.count()), q(p.size());
        for (size_t i = 0; i < a.size(); ++i) {
            a[i] = 100.f + (float)((i * 2654435761u) % 997);
            b[i] = 100.f + (float)((i * 40503u) % 991);
        }
        for (size_t i = 0; i < p.size(); ++i) {
            p[i] = 1.f + (float)(i % 7);
            q[i] = 1.f + (float)(i % 5);
        }
        for (int dual = 0; dual < 2; ++dual) {
            std::vector<float> ref;
            for (int threads : {1, 3, 8}) {
                std::vector<float> out(a.size(), -1.f);
                cpu::DeconCpuArgs args;
                args.out = out.data();
                args.img[0] = a.data(); args.img[1] = b.data();
                args.psf[0] = p.data(); args.psf[1] = q.data();
                args.im = im; args.psf_dims = psf; args.box = box;
                args.dual = dual != 0; args.const_init = dual != 0; args.iterations = 3; args.threads = threads;
                if (cpu::decon(args) != 0) return 1;
                for (float v : out)
                    if (!(v >= 0.01f) || !std::isfinite(v)) { printf("bad value %g\n", v); return 1; }
                if (ref.empty()) ref = out;
                else if (std::memcmp(ref.data(), out.data(), out.size() * sizeof(float)) != 0) {
                    printf("threads change the result (%d threads)\n", threads);
                    return 1;
                }
            }
        }
    }
    printf("cpu decon clean\n");
    return 0;
}
