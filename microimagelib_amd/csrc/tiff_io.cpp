// Self-contained baseline-TIFF stack codec behind the libapi.h file functions.
//
// The reference goes through libtiff (src/apifunc.cpp:116-326); this image has no libtiff
// headers, and the on-disk format the reference's tools exchange is narrow: classic TIFF,
// one IFD per slice, 1 sample per pixel, uncompressed strips, 16-bit unsigned or 32-bit
// IEEE float.  The reader below accepts exactly that family (either byte order, any number
// of strips per slice); the writer emits what the reference's writetifstack emits (one strip
// per slice, ROWSPERSTRIP = height, MINISBLACK, ORIENTATION_TOPLEFT, PLANARCONFIG_SEPARATE).
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/libapi.h"

namespace {

struct Slice {
    uint32_t width = 0, height = 0;
    uint16_t bits = 0, compression = 1, sample_format = 1, samples = 1;
    bool has_sample_format = false;
    std::vector<uint64_t> strip_offsets, strip_bytes;
};

class TiffReader {
public:
    explicit TiffReader(const char *path) : f_(fopen(path, "rb")), path_(path) {
        if (!f_) fail("cannot open");
        unsigned char h[8];
        if (fread(h, 1, 8, f_) != 8) fail("truncated header");
        if (h[0] == 'I' && h[1] == 'I') big_endian_ = false;
        else if (h[0] == 'M' && h[1] == 'M') big_endian_ = true;
        else fail("not a TIFF file");
        if (u16(h + 2) != 42) fail("unsupported TIFF flavour (only classic TIFF, magic 42)");
        if (fseeko(f_, 0, SEEK_END) != 0) fail("seek failed");
        file_size_ = (uint64_t)ftello(f_);
        uint64_t ifd = u32(h + 4);
        while (ifd != 0) {
            slices_.push_back(read_ifd(ifd, &ifd));
            if (slices_.size() > (1u << 24)) fail("IFD chain too long");
        }
        if (slices_.empty()) fail("no image directories");
        // The callers size their buffers from slice 0 (gettifinfo): every directory must describe the same kind of
        // slice -- a thumbnail / pyramid level, a slice of another depth or a corrupt directory would otherwise be
        // written past the end of the caller's buffer.
        const Slice &s0 = slices_[0];
        if (s0.width == 0 || s0.height == 0) fail("empty slice");
        if (s0.samples != 1) fail("only 1 sample per pixel is supported");
        if (s0.compression != 1) fail("compressed TIFF is not supported");
        if (s0.bits == 32 && s0.has_sample_format && s0.sample_format != 3)
            fail("32-bit integer samples are not supported (expected IEEE float)");
        for (const Slice &s : slices_)
            if (s.width != s0.width || s.height != s0.height || s.bits != s0.bits || s.samples != s0.samples ||
                s.compression != s0.compression || s.sample_format != s0.sample_format)
                fail("slices differ in size, depth or sample format (thumbnail / pyramid directories are not supported)");
    }
    ~TiffReader() {
        if (f_) fclose(f_);
    }
    const std::vector<Slice> &slices() const { return slices_; }
    bool big_endian() const { return big_endian_; }

    // raw bytes of slice i (all strips concatenated) into dst (width*height*bits/8 bytes)
    void read_slice(size_t i, unsigned char *dst) {
        const Slice &s = slices_[i];  // homogeneous with slice 0, uncompressed, 1 sample: checked at open
        const uint64_t want = (uint64_t)s.width * s.height * (s.bits / 8);
        uint64_t got = 0;
        for (size_t k = 0; k < s.strip_offsets.size() && got < want; ++k) {
            uint64_t nb = k < s.strip_bytes.size() ? s.strip_bytes[k] : want - got;
            if (nb > want - got) nb = want - got;
            if (fseeko(f_, (off_t)s.strip_offsets[k], SEEK_SET) != 0) fail("seek failed");
            if (fread(dst + got, 1, nb, f_) != nb) fail("truncated strip");
            got += nb;
        }
        if (got != want) fail("slice data shorter than width*height");
    }

    [[noreturn]] void fail(const char *why) const {
        fprintf(stderr, "*** TIFF read error (%s): %s\n", why, path_.c_str());
        exit(1);  // the reference's I/O layer exits on unreadable input (src/apifunc.cpp:117-120)
    }

private:
    uint16_t u16(const unsigned char *p) const {
        return big_endian_ ? (uint16_t)((p[0] << 8) | p[1]) : (uint16_t)((p[1] << 8) | p[0]);
    }
    uint32_t u32(const unsigned char *p) const {
        return big_endian_ ? ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]
                           : ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
    }

    std::vector<uint64_t> values(const unsigned char *entry) {
        const uint16_t type = u16(entry + 2);
        const uint32_t count = u32(entry + 4);
        const size_t sz = (type == 3) ? 2 : (type == 4) ? 4 : (type == 1 || type == 2) ? 1 : 0;
        std::vector<uint64_t> out;
        if (sz == 0 || count == 0) return out;
        if ((uint64_t)count * sz > file_size_) fail("tag data larger than the file");
        std::vector<unsigned char> buf((size_t)count * sz);
        if ((size_t)count * sz <= 4) {
            memcpy(buf.data(), entry + 8, buf.size());
        } else {
            if (fseeko(f_, (off_t)u32(entry + 8), SEEK_SET) != 0) fail("seek failed");
            if (fread(buf.data(), 1, buf.size(), f_) != buf.size()) fail("truncated tag data");
        }
        out.reserve(count);
        for (uint32_t i = 0; i < count; ++i) {
            const unsigned char *p = buf.data() + (size_t)i * sz;
            out.push_back(sz == 2 ? u16(p) : sz == 4 ? u32(p) : p[0]);
        }
        return out;
    }

    Slice read_ifd(uint64_t off, uint64_t *next) {
        if (fseeko(f_, (off_t)off, SEEK_SET) != 0) fail("seek failed");
        unsigned char nb[2];
        if (fread(nb, 1, 2, f_) != 2) fail("truncated IFD");
        const uint16_t n = u16(nb);
        std::vector<unsigned char> ent((size_t)n * 12 + 4);
        if (fread(ent.data(), 1, ent.size(), f_) != ent.size()) fail("truncated IFD");
        Slice s;
        for (uint16_t i = 0; i < n; ++i) {
            const unsigned char *e = ent.data() + (size_t)i * 12;
            const uint16_t tag = u16(e);
            if (tag != 256 && tag != 257 && tag != 258 && tag != 259 && tag != 273 && tag != 277 && tag != 279 &&
                tag != 339)
                continue;
            std::vector<uint64_t> v = values(e);
            if (v.empty()) continue;
            switch (tag) {
            case 256: s.width = (uint32_t)v[0]; break;
            case 257: s.height = (uint32_t)v[0]; break;
            case 258: s.bits = (uint16_t)v[0]; break;
            case 259: s.compression = (uint16_t)v[0]; break;
            case 273: s.strip_offsets = v; break;
            case 277: s.samples = (uint16_t)v[0]; break;
            case 279: s.strip_bytes = v; break;
            case 339: s.sample_format = (uint16_t)v[0]; s.has_sample_format = true; break;
            }
        }
        *next = u32(ent.data() + (size_t)n * 12);
        return s;
    }

    FILE *f_;
    std::string path_;
    bool big_endian_ = false;
    uint64_t file_size_ = 0;
    std::vector<Slice> slices_;
};

void swap_bytes(unsigned char *p, size_t count, size_t width) {
    for (size_t i = 0; i < count; ++i, p += width)
        for (size_t a = 0, b = width - 1; a < b; ++a, --b) {
            unsigned char t = p[a];
            p[a] = p[b];
            p[b] = t;
        }
}

class TiffWriter {
public:
    explicit TiffWriter(const char *path) : f_(fopen(path, "wb")), path_(path) {
        if (!f_) {
            fprintf(stderr, "*** Failed to create file!!! Please check the directory: %s\n", path);
            exit(1);  // reference src/apifunc.cpp:248-251
        }
        const unsigned char hdr[8] = {'I', 'I', 42, 0, 0, 0, 0, 0};
        put(hdr, 8);
        link_pos_ = 4;
        pos_ = 8;
    }
    ~TiffWriter() { close(); }
    // flushes and closes; a short write (full disk) is an error, not a silently truncated stack
    void close() {
        if (!f_) return;
        FILE *f = f_;
        f_ = nullptr;
        if (fclose(f) != 0) fail();
    }
    void add_slice(const void *data, uint32_t width, uint32_t height, uint16_t bits, bool ieee_float) {
        const uint64_t nbytes = (uint64_t)width * height * (bits / 8);
        if (pos_ + nbytes + 256 > 0xFFFFFFFFull) {
            fprintf(stderr, "*** TIFF write error: stack exceeds the 4 GiB classic-TIFF limit\n");
            exit(1);
        }
        const uint32_t data_off = (uint32_t)pos_;
        put(data, nbytes);
        pos_ += nbytes;
        if (pos_ & 1) {
            const unsigned char zero = 0;
            put(&zero, 1);
            ++pos_;
        }
        struct Entry { uint16_t tag, type; uint32_t count, value; };
        std::vector<Entry> e = {
            {256, 4, 1, width}, {257, 4, 1, height}, {258, 3, 1, bits}, {259, 3, 1, 1},
            {262, 3, 1, 1},     {273, 4, 1, data_off}, {274, 3, 1, 1},  {277, 3, 1, 1},
            {278, 4, 1, height}, {279, 4, 1, (uint32_t)nbytes}, {284, 3, 1, 2},
        };
        if (ieee_float) e.push_back({339, 3, 1, 3});
        const uint32_t ifd_off = (uint32_t)pos_;
        patch32(link_pos_, ifd_off);
        put16((uint16_t)e.size());
        for (const Entry &x : e) {
            put16(x.tag);
            put16(x.type);
            put32(x.count);
            put32(x.value);  // SHORT values sit left-justified: little-endian low half first
        }
        link_pos_ = pos_;
        put32(0);
    }

private:
    [[noreturn]] void fail() const {
        fprintf(stderr, "*** TIFF write error (disk full or I/O error): %s\n", path_.c_str());
        exit(1);
    }
    void put(const void *data, size_t n) {
        if (fwrite(data, 1, n, f_) != n) fail();
    }
    void put16(uint16_t v) {
        unsigned char b[2] = {(unsigned char)(v & 255), (unsigned char)(v >> 8)};
        put(b, 2);
        pos_ += 2;
    }
    void put32(uint32_t v) {
        unsigned char b[4] = {(unsigned char)(v & 255), (unsigned char)((v >> 8) & 255),
                              (unsigned char)((v >> 16) & 255), (unsigned char)(v >> 24)};
        put(b, 4);
        pos_ += 4;
    }
    void patch32(uint64_t where, uint32_t v) {
        if (fseeko(f_, (off_t)where, SEEK_SET) != 0) fail();
        unsigned char b[4] = {(unsigned char)(v & 255), (unsigned char)((v >> 8) & 255),
                              (unsigned char)((v >> 16) & 255), (unsigned char)(v >> 24)};
        put(b, 4);
        if (fseeko(f_, (off_t)pos_, SEEK_SET) != 0) fail();
    }
    FILE *f_;
    std::string path_;
    uint64_t pos_ = 0, link_pos_ = 0;
};

bool host_is_big_endian() {
    const uint16_t one = 1;
    return *(const unsigned char *)&one == 0;
}

void require_file(const char *path, const char *what) {
    if (!fexists(path)) {
        fprintf(stderr, "*** %s File does not exist: %s\n", what, path);
        exit(1);
    }
}

}  // namespace

// ---- libapi.h -----------------------------------------------------------------
extern "C" char *concat(int count, ...) {  // reference src/apifunc.cpp:53-80: caller frees
    va_list ap;
    size_t len = 1;
    va_start(ap, count);
    for (int i = 0; i < count; ++i) len += strlen(va_arg(ap, char *));
    va_end(ap);
    char *merged = (char *)calloc(len, 1);
    size_t at = 0;
    va_start(ap, count);
    for (int i = 0; i < count; ++i) {
        const char *s = va_arg(ap, char *);
        const size_t n = strlen(s);
        memcpy(merged + at, s, n);
        at += n;
    }
    va_end(ap);
    return merged;
}

extern "C" bool fexists(const char *filename) {
    FILE *f = fopen(filename, "r");
    if (!f) return false;
    fclose(f);
    return true;
}

extern "C" unsigned short gettifinfo(char tifdir[], unsigned int *tifSize) {
    require_file(tifdir, "");
    TiffReader r(tifdir);
    tifSize[0] = r.slices()[0].width;
    tifSize[1] = r.slices()[0].height;
    tifSize[2] = (unsigned int)r.slices().size();
    return r.slices()[0].bits;
}

extern "C" void readtifstack(float *h_Image, char *tifdir, unsigned int *imsize) {
    require_file(tifdir, "Failed to read image!!!");
    TiffReader r(tifdir);
    const Slice &s0 = r.slices()[0];
    imsize[0] = s0.width;
    imsize[1] = s0.height;
    imsize[2] = (unsigned int)r.slices().size();
    const size_t plane = (size_t)s0.width * s0.height;
    const bool swap = r.big_endian() != host_is_big_endian();
    if (s0.bits == 16) {  // uint16 -> float (reference :166-183)
        std::vector<uint16_t> buf(plane);
        for (size_t n = 0; n < r.slices().size(); ++n) {
            r.read_slice(n, (unsigned char *)buf.data());
            if (swap) swap_bytes((unsigned char *)buf.data(), plane, 2);
            float *dst = h_Image + n * plane;
            for (size_t i = 0; i < plane; ++i) dst[i] = (float)buf[i];
        }
    } else if (s0.bits == 32) {  // float32 passthrough (:184-195)
        for (size_t n = 0; n < r.slices().size(); ++n) {
            r.read_slice(n, (unsigned char *)(h_Image + n * plane));
            if (swap) swap_bytes((unsigned char *)(h_Image + n * plane), plane, 4);
        }
    }
    // other depths: silently nothing, like the reference
}

extern "C" void readtifstack_16to16(unsigned short *h_Image, char *tifdir, unsigned int *imsize) {
    require_file(tifdir, "Failed to read image!!!");
    TiffReader r(tifdir);
    const Slice &s0 = r.slices()[0];
    imsize[0] = s0.width;
    imsize[1] = s0.height;
    imsize[2] = (unsigned int)r.slices().size();
    if (s0.bits != 16) {
        printf("Image bit per sample is not supported, please set input image as 16 bit!!!\n\n");
        return;
    }
    const size_t plane = (size_t)s0.width * s0.height;
    const bool swap = r.big_endian() != host_is_big_endian();
    for (size_t n = 0; n < r.slices().size(); ++n) {
        r.read_slice(n, (unsigned char *)(h_Image + n * plane));
        if (swap) swap_bytes((unsigned char *)(h_Image + n * plane), plane, 2);
    }
}

extern "C" void writetifstack(char *tifdir, float *h_Image, unsigned int *imsize, unsigned short bitPerSample) {
    const size_t plane = (size_t)imsize[0] * imsize[1];
    TiffWriter w(tifdir);
    if (bitPerSample == 16) {
        std::vector<uint16_t> buf(plane);
        for (unsigned int n = 0; n < imsize[2]; ++n) {
            const float *src = h_Image + (size_t)n * plane;
            for (size_t i = 0; i < plane; ++i) buf[i] = (uint16_t)(long long)src[i];  // C-cast truncation (:255)
            w.add_slice(buf.data(), imsize[0], imsize[1], 16, false);
        }
    } else if (bitPerSample == 32) {
        for (unsigned int n = 0; n < imsize[2]; ++n)
            w.add_slice(h_Image + (size_t)n * plane, imsize[0], imsize[1], 32, true);
    } else {
        printf("Image bit per sample is not supported, please set bitPerPample to 16 or 32 !!!\n\n");
    }
    w.close();
}

extern "C" void writetifstack_16to16(char *tifdir, unsigned short *h_Image, unsigned int *imsize) {
    const size_t plane = (size_t)imsize[0] * imsize[1];
    TiffWriter w(tifdir);
    for (unsigned int n = 0; n < imsize[2]; ++n)
        w.add_slice(h_Image + (size_t)n * plane, imsize[0], imsize[1], 16, false);
    w.close();
}
