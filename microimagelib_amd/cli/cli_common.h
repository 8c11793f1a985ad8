// Shared pieces of the command-line front ends (deconSingleView, deconDualView, reg3D, checkGPUDevice).
// The programs accept the same named flags, defaults and file formats as the reference's tools
// (src/decon_sv.cpp:25-39,72-133; src/decon_dv.cpp:27-45; src/reg3D.cpp:24-45,73-76) and only call libapi.h.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/libapi.h"

namespace cli {

struct Option {
    const char *flag;       // "-it"
    const char *value;      // "<int>" or nullptr for a switch
    const char *help;
    std::function<void(const char *)> set;  // receives the value (or nullptr for a switch)
};

class Parser {
public:
    Parser(std::string app, std::string title, std::string usage)
        : app_(std::move(app)), title_(std::move(title)), usage_(std::move(usage)) {}
    void add(const char *flag, const char *value, const char *help, std::function<void(const char *)> set) {
        opts_.push_back(Option{flag, value, help, std::move(set)});
    }
    void brief() const {
        printf("\n%s: %s\n\nUsage:\t%s %s [OPTIONS]\n", app_.c_str(), title_.c_str(), app_.c_str(), usage_.c_str());
        printf("\nUse command for more details:\n\t%s -help or %s -h\n", app_.c_str(), app_.c_str());
    }
    void full() const {
        printf("\n%s: %s\n\nUsage:\t%s %s [OPTIONS]\n", app_.c_str(), title_.c_str(), app_.c_str(), usage_.c_str());
        printf("\tOnly 16-bit or 32-bit standard TIFF images are currently supported.\n\n= = = [OPTIONS] = = =\n");
        for (const Option &o : opts_) {
            std::string lhs = std::string(o.flag) + (o.value ? std::string(" ") + o.value : std::string());
            printf("\t%-22s %s\n", lhs.c_str(), o.help);
        }
    }
    // returns 0 to continue, 1 when help was printed, -1 on a malformed command line
    int parse(int argc, char **argv) const {
        if (argc == 1) { brief(); return 1; }
        for (int i = 1; i < argc; ++i) {
            if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "-help")) { full(); return 1; }
            bool known = false;
            for (const Option &o : opts_) {
                if (strcmp(argv[i], o.flag)) continue;
                known = true;
                if (o.value) {
                    if (i + 1 >= argc) { fprintf(stderr, "%s: option %s needs a value\n", app_.c_str(), o.flag); return -1; }
                    o.set(argv[++i]);
                } else {
                    o.set(nullptr);
                }
                break;
            }
            (void)known;  // unknown words are ignored, like the reference's parsers
        }
        return 0;
    }

private:
    std::string app_, title_, usage_;
    std::vector<Option> opts_;
};

inline const char *mode_text(int gpuMemMode) {
    switch (gpuMemMode) {
    case -1: return "automatically setting";
    case 0: return "CPU";
    case 1: return "efficient GPU";
    case 2: return "memory-saved GPU";
    default: return nullptr;
    }
}

struct Volume {
    std::vector<float> data;
    unsigned int size[3] = {0, 0, 0};
    unsigned int bits = 0;
    size_t count() const { return (size_t)size[0] * size[1] * size[2]; }
};

inline Volume load_tiff(const std::string &path) {
    Volume v;
    v.bits = gettifinfo(const_cast<char *>(path.c_str()), v.size);  // exits on a missing file, like the reference
    v.data.resize(v.count());
    readtifstack(v.data.data(), const_cast<char *>(path.c_str()), v.size);
    return v;
}

inline bool same_size(const Volume &a, const Volume &b) {
    return a.size[0] == b.size[0] && a.size[1] == b.size[1] && a.size[2] == b.size[2];
}

}  // namespace cli
