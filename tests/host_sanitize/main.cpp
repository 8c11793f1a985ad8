// Host-only pieces of libapi.so under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool):
// the TIFF codec (round trips at ragged sizes, 16- and 32-bit) and the Powell optimiser.  Built and run by
// tests/test_host_sanitize.py; exit code 0 = clean.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/libapi.h"
#include "../../microimagelib_amd/csrc/powell.h"

int main(int argc, char **argv) {
    if (argc > 3 && std::string(argv[2]) == "read") {
        // malformed-input mode: size the buffer from gettifinfo like every caller does, then read.  The library must
        // refuse the file (exit 1 with a TIFF read error) -- never write past the buffer (ASan would report it).
        unsigned sz[3] = {0, 0, 0};
        gettifinfo(argv[3], sz);
        std::vector<float> buf((size_t)sz[0] * sz[1] * sz[2]);
        readtifstack(buf.data(), argv[3], sz);
        printf("read %u x %u x %u\n", sz[0], sz[1], sz[2]);
        return 0;
    }
    const std::string path = std::string(argc > 1 ? argv[1] : "/tmp") + "/sanitize_roundtrip.tif";
    const unsigned sizes[][3] = {{7, 5, 3}, {1, 1, 1}, {33, 2, 1}, {16, 16, 9}};
    for (auto &sz : sizes) {
        unsigned s[3] = {sz[0], sz[1], sz[2]};
        const size_t n = (size_t)s[0] * s[1] * s[2];
        std::vector<float> a(n), b(n);
        for (size_t i = 0; i < n; ++i) a[i] = (float)((i * 37) % 1000);
        for (unsigned short bits : {(unsigned short)32, (unsigned short)16}) {
            writetifstack(const_cast<char *>(path.c_str()), a.data(), s, bits);
            unsigned got[3] = {0, 0, 0};
            if (gettifinfo(const_cast<char *>(path.c_str()), got) != bits || got[0] != s[0] || got[1] != s[1] || got[2] != s[2]) {
                printf("TIFF info mismatch\n");
                return 1;
            }
            readtifstack(b.data(), const_cast<char *>(path.c_str()), got);
            if (memcmp(a.data(), b.data(), n * sizeof(float)) != 0) {
                printf("TIFF data mismatch (%u bit)\n", bits);
                return 1;
            }
            if (!fexists(const_cast<char *>(path.c_str()))) return 1;
        }
    }
    remove(path.c_str());
    char *c = concat(3, (char *)"ab", (char *)"", (char *)"cde");
    if (strcmp(c, "abcde") != 0) return 1;
    free(c);

    // Powell on a 12-parameter cost in the range the registration uses (-1 .. 0)
    int evals = 0;
    mil::CostFn cost = [&](const float *x) {
        ++evals;
        float s = 0;
        for (int i = 0; i < 12; ++i) s += (x[i] - 0.1f * i) * (x[i] - 0.1f * i) * (1 + i);
        return -std::exp(-0.05f * s);
    };
    float p[12] = {0}, xi[144] = {0};
    for (int i = 0; i < 12; ++i) xi[i * 12 + i] = 1.f;
    const mil::PowellResult r = mil::powell_minimize(p, xi, 12, 12, 1e-6f, cost, &evals, 3000);
    printf("powell: fret %g after %d evaluations, p[11] = %g\n", r.fret, evals, p[11]);
    return std::fabs(p[11] - 1.1f) < 1e-2f ? 0 : 1;
}
