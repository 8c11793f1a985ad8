// Self-contained TIFF stack codec behind the libapi.h file functions.
//
// The reference goes through libtiff (src/apifunc.cpp:116-326: TIFFOpen + TIFFReadScanline per row); this image has no
// libtiff headers.  The reader accepts what that call sequence accepts for the stacks microscopes and ImageJ / Fiji /
// Python write: classic TIFF and BigTIFF (magic 43, 64-bit offsets -- stacks beyond 4 GiB are routine), either byte
// order, one IFD per slice, 1 sample per pixel, 16-bit unsigned or 32-bit IEEE float, any number of STRIPS per slice,
// strips uncompressed or compressed with PackBits (32773), LZW (5) or Deflate (8 / 32946, zlib), with the horizontal
// (2) or floating-point (3) predictor.  Tiled files are refused -- TIFFReadScanline refuses them too.  The writer emits
// what the reference's writetifstack emits (one uncompressed strip per slice, ROWSPERSTRIP = height, MINISBLACK,
// ORIENTATION_TOPLEFT, PLANARCONFIG_SEPARATE) and switches to BigTIFF by itself when the stack would not fit 4 GiB
// (libtiff's classic writer fails there).
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <zlib.h>

#include "../../include/libapi.h"

namespace {

struct Slice {
    uint32_t width = 0, height = 0;
    uint16_t bits = 0, compression = 1, sample_format = 1, samples = 1, predictor = 1;
    uint32_t rows_per_strip = 0;  // 0: not given = the whole slice
    bool has_sample_format = false, tiled = false;
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
        const uint16_t magic = u16(h + 2);
        if (magic != 42 && magic != 43) fail("not a TIFF file (magic is neither 42 nor 43)");
        big_ = magic == 43;
        if (fseeko(f_, 0, SEEK_END) != 0) fail("seek failed");
        file_size_ = (uint64_t)ftello(f_);
        uint64_t ifd;
        if (big_) {  // BigTIFF: offset size 8, zero, then the 64-bit offset of the first directory
            unsigned char h2[8];
            if (u16(h + 4) != 8 || u16(h + 6) != 0) fail("malformed BigTIFF header");
            if (fseeko(f_, 8, SEEK_SET) != 0 || fread(h2, 1, 8, f_) != 8) fail("truncated header");
            ifd = u64(h2);
        } else {
            ifd = u32(h + 4);
        }
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
        if (s0.tiled) fail("tiled TIFF is not supported (the reference reads scanlines, which tiled files do not have)");
        if (s0.compression != 1 && s0.compression != 5 && s0.compression != 8 && s0.compression != 32946 &&
            s0.compression != 32773)
            fail("unsupported compression (none, PackBits, LZW and Deflate are read)");
        if (s0.predictor != 1 && s0.predictor != 2 && s0.predictor != 3) fail("unsupported predictor");
        if (s0.predictor == 3 && !(s0.bits == 32 && (!s0.has_sample_format || s0.sample_format == 3)))
            fail("floating-point predictor on non-float samples");
        if (s0.bits != 16 && s0.bits != 32 && s0.compression != 1) fail("compressed slices of this depth are not supported");
        {   // a slice the file cannot possibly hold (corrupt extents) must not size the caller's buffers
            const uint64_t slice_bytes = (uint64_t)s0.width * s0.height * (s0.bits / 8 ? s0.bits / 8 : 1);
            const uint64_t ratio = s0.compression == 1 ? 1 : 4096;  // no codec read here expands further
            if (slice_bytes / ratio > file_size_) fail("slice extents larger than the file can hold");
        }
        if (s0.bits == 32 && s0.has_sample_format && s0.sample_format != 3)
            fail("32-bit integer samples are not supported (expected IEEE float)");
        for (const Slice &s : slices_)
            if (s.width != s0.width || s.height != s0.height || s.bits != s0.bits || s.samples != s0.samples ||
                s.compression != s0.compression || s.sample_format != s0.sample_format || s.predictor != s0.predictor ||
                s.tiled != s0.tiled)
                fail("slices differ in size, depth or sample format (thumbnail / pyramid directories are not supported)");
    }
    ~TiffReader() {
        if (f_) fclose(f_);
    }
    const std::vector<Slice> &slices() const { return slices_; }
    bool big_endian() const { return big_endian_; }

    // decoded bytes of slice i (all strips, in file byte order) into dst (width*height*bits/8 bytes)
    void read_slice(size_t i, unsigned char *dst) {
        const Slice &s = slices_[i];  // homogeneous with slice 0, 1 sample: checked at open
        const uint64_t row_bytes = (uint64_t)s.width * (s.bits / 8);
        const uint64_t want = row_bytes * s.height;
        if (s.compression == 1) {  // strips are the rows in order: read what is there, up to the slice
            uint64_t got = 0;
            for (size_t k = 0; k < s.strip_offsets.size() && got < want; ++k) {
                uint64_t nb = k < s.strip_bytes.size() ? s.strip_bytes[k] : want - got;
                if (nb > want - got) nb = want - got;
                if (fseeko(f_, (off_t)s.strip_offsets[k], SEEK_SET) != 0) fail("seek failed");
                if (fread(dst + got, 1, nb, f_) != nb) fail("truncated strip");
                got += nb;
            }
            if (got != want) fail("slice data shorter than width*height");
        } else {
            const uint64_t rps = (s.rows_per_strip == 0 || s.rows_per_strip > s.height) ? s.height : s.rows_per_strip;
            const uint64_t n_strips = (s.height + rps - 1) / rps;
            if (s.strip_offsets.size() < n_strips || s.strip_bytes.size() < n_strips) fail("fewer strips than the slice needs");
            for (uint64_t k = 0; k < n_strips; ++k) {
                const uint64_t rows = (k + 1 == n_strips) ? s.height - k * rps : rps;
                const uint64_t out_bytes = rows * row_bytes;
                const uint64_t nb = s.strip_bytes[k];
                if (nb > file_size_ || s.strip_offsets[k] > file_size_ - nb) fail("strip outside the file");
                packed_.resize((size_t)nb);
                if (fseeko(f_, (off_t)s.strip_offsets[k], SEEK_SET) != 0) fail("seek failed");
                if (nb && fread(packed_.data(), 1, (size_t)nb, f_) != nb) fail("truncated strip");
                unsigned char *out = dst + k * rps * row_bytes;
                uint64_t got = 0;
                switch (s.compression) {
                case 32773: got = unpack_bits(packed_.data(), (size_t)nb, out, (size_t)out_bytes); break;
                case 5: got = unpack_lzw(packed_.data(), (size_t)nb, out, (size_t)out_bytes); break;
                default: got = unpack_deflate(packed_.data(), (size_t)nb, out, (size_t)out_bytes); break;
                }
                if (got != out_bytes) fail("compressed strip decodes to fewer bytes than its rows need");
            }
        }
        if (s.predictor == 2) undo_horizontal(dst, s);
        else if (s.predictor == 3) undo_float_predictor(dst, s);
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

    uint64_t u64(const unsigned char *p) const {
        return big_endian_ ? ((uint64_t)u32(p) << 32) | u32(p + 4) : ((uint64_t)u32(p + 4) << 32) | u32(p);
    }

    // values of one directory entry: classic 12 bytes (count 32 bit, 4 value bytes), BigTIFF 20 (count 64, 8 value bytes)
    std::vector<uint64_t> values(const unsigned char *entry) {
        const uint16_t type = u16(entry + 2);
        const uint64_t count = big_ ? u64(entry + 4) : u32(entry + 4);
        const unsigned char *inl = entry + (big_ ? 12 : 8);
        const size_t inl_bytes = big_ ? 8 : 4;
        const size_t sz = (type == 3) ? 2 : (type == 4) ? 4 : (type == 16 || type == 18) ? 8 : (type == 1 || type == 2) ? 1 : 0;
        std::vector<uint64_t> out;
        if (sz == 0 || count == 0) return out;
        if (count > file_size_ || count * sz > file_size_) fail("tag data larger than the file");
        std::vector<unsigned char> buf((size_t)(count * sz));
        if (count * sz <= inl_bytes) {
            memcpy(buf.data(), inl, buf.size());
        } else {
            const uint64_t where = big_ ? u64(inl) : u32(inl);
            if (where > file_size_ - buf.size()) fail("tag data outside the file");
            if (fseeko(f_, (off_t)where, SEEK_SET) != 0) fail("seek failed");
            if (fread(buf.data(), 1, buf.size(), f_) != buf.size()) fail("truncated tag data");
        }
        out.reserve((size_t)count);
        for (uint64_t i = 0; i < count; ++i) {
            const unsigned char *p = buf.data() + (size_t)i * sz;
            out.push_back(sz == 2 ? u16(p) : sz == 4 ? u32(p) : sz == 8 ? u64(p) : p[0]);
        }
        return out;
    }

    Slice read_ifd(uint64_t off, uint64_t *next) {
        const size_t esz = big_ ? 20 : 12, csz = big_ ? 8 : 2, nsz = big_ ? 8 : 4;
        if (off > file_size_ || file_size_ - off < csz) fail("directory outside the file");
        if (fseeko(f_, (off_t)off, SEEK_SET) != 0) fail("seek failed");
        unsigned char nb[8];
        if (fread(nb, 1, csz, f_) != csz) fail("truncated IFD");
        const uint64_t n = big_ ? u64(nb) : u16(nb);
        if (n > 65535 || n * esz + nsz > file_size_) fail("implausible directory entry count");
        std::vector<unsigned char> ent((size_t)n * esz + nsz);
        if (fread(ent.data(), 1, ent.size(), f_) != ent.size()) fail("truncated IFD");
        Slice s;
        for (uint64_t i = 0; i < n; ++i) {
            const unsigned char *e = ent.data() + (size_t)i * esz;
            const uint16_t tag = u16(e);
            if (tag != 256 && tag != 257 && tag != 258 && tag != 259 && tag != 273 && tag != 277 && tag != 278 && tag != 279 &&
                tag != 317 && tag != 322 && tag != 324 && tag != 339)
                continue;
            if (tag == 322 || tag == 324) {  // TileWidth / TileOffsets
                s.tiled = true;
                continue;
            }
            std::vector<uint64_t> v = values(e);
            if (v.empty()) continue;
            switch (tag) {
            case 256: s.width = (uint32_t)v[0]; break;
            case 257: s.height = (uint32_t)v[0]; break;
            case 258: s.bits = (uint16_t)v[0]; break;
            case 259: s.compression = (uint16_t)v[0]; break;
            case 273: s.strip_offsets = v; break;
            case 277: s.samples = (uint16_t)v[0]; break;
            case 278: s.rows_per_strip = (uint32_t)(v[0] > 0xFFFFFFFFull ? 0xFFFFFFFFull : v[0]); break;
            case 279: s.strip_bytes = v; break;
            case 317: s.predictor = (uint16_t)v[0]; break;
            case 339: s.sample_format = (uint16_t)v[0]; s.has_sample_format = true; break;
            }
        }
        *next = big_ ? u64(ent.data() + (size_t)n * esz) : u32(ent.data() + (size_t)n * esz);
        return s;
    }

    // ---- strip decoders: each returns the number of bytes produced (at most `cap`), never writes beyond dst + cap
    static size_t unpack_bits(const unsigned char *src, size_t n, unsigned char *dst, size_t cap) {
        size_t i = 0, o = 0;
        while (i < n && o < cap) {
            const int c = (signed char)src[i++];
            if (c >= 0) {  // c + 1 literal bytes
                size_t run = (size_t)c + 1;
                if (run > n - i) run = n - i;
                if (run > cap - o) run = cap - o;
                memcpy(dst + o, src + i, run);
                i += (size_t)c + 1;
                o += run;
            } else if (c != -128) {  // 1 - c copies of the next byte
                if (i >= n) break;
                size_t run = (size_t)(1 - c);
                if (run > cap - o) run = cap - o;
                memset(dst + o, src[i++], run);
                o += run;
            }
        }
        return o;
    }

    // TIFF LZW: codes MSB first, 9 to 12 bits, Clear = 256, EndOfInformation = 257, width grows one code early
    static size_t unpack_lzw(const unsigned char *src, size_t n, unsigned char *dst, size_t cap) {
        struct Node { uint16_t prev; unsigned char ch; uint16_t len; };
        static thread_local std::vector<Node> table;
        table.resize(4096);
        for (unsigned c = 0; c < 256; ++c) table[c] = Node{0xFFFF, (unsigned char)c, 1};
        unsigned next = 258, width = 9;
        int prev = -1;
        uint64_t acc = 0;
        unsigned bits = 0;
        size_t i = 0, o = 0;
        while (o < cap) {
            while (bits < width && i < n) {
                acc = (acc << 8) | src[i++];
                bits += 8;
            }
            if (bits < width) break;
            const unsigned code = (unsigned)((acc >> (bits - width)) & ((1u << width) - 1));
            bits -= width;
            if (code == 257) break;
            if (code == 256) {
                next = 258;
                width = 9;
                prev = -1;
                continue;
            }
            unsigned entry;
            if (prev < 0) {
                if (code >= 256) return o;  // corrupt: the first code after a clear must be a literal
                entry = code;
            } else if (code < next) {
                entry = code;
                if (next < 4096) {  // prev + first character of this entry
                    unsigned c = code;
                    while (table[c].prev != 0xFFFF) c = table[c].prev;
                    table[next] = Node{(uint16_t)prev, table[c].ch, (uint16_t)(table[(unsigned)prev].len + 1)};
                    ++next;
                }
            } else if (code == next && next < 4096) {  // KwKwK: prev + first character of prev
                unsigned c = (unsigned)prev;
                while (table[c].prev != 0xFFFF) c = table[c].prev;
                table[next] = Node{(uint16_t)prev, table[c].ch, (uint16_t)(table[(unsigned)prev].len + 1)};
                entry = next++;
            } else {
                return o;  // corrupt stream
            }
            const size_t len = table[entry].len;
            size_t room = cap - o;
            // write the string backwards, clipped to the room that is left
            unsigned c = entry;
            for (size_t k = len; k-- > 0;) {
                if (k < room) dst[o + k] = table[c].ch;
                c = table[c].prev;
            }
            o += len < room ? len : room;
            prev = (int)entry;
            if (next + 1 >= (1u << width) && width < 12) ++width;  // "early change"
        }
        return o;
    }

    static size_t unpack_deflate(const unsigned char *src, size_t n, unsigned char *dst, size_t cap) {
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit(&z) != Z_OK) return 0;
        z.next_in = const_cast<unsigned char *>(src);
        z.avail_in = (uInt)(n > 0xFFFFFFFFu ? 0xFFFFFFFFu : n);
        z.next_out = dst;
        z.avail_out = (uInt)(cap > 0xFFFFFFFFu ? 0xFFFFFFFFu : cap);
        const int rc = inflate(&z, Z_FINISH);
        const size_t produced = (size_t)z.total_out;
        inflateEnd(&z);
        return (rc == Z_STREAM_END || rc == Z_OK || rc == Z_BUF_ERROR) ? produced : 0;
    }

    // Predictor 2: every sample but the first of a row is stored as the difference to its left neighbour, in the file's
    // byte order (integer arithmetic modulo the sample width; float samples are treated as 32-bit integers, as libtiff does)
    void undo_horizontal(unsigned char *dst, const Slice &s) const {
        const size_t bytes = s.bits / 8;
        for (uint32_t y = 0; y < s.height; ++y) {
            unsigned char *row = dst + (size_t)y * s.width * bytes;
            if (bytes == 2) {
                uint16_t acc = u16(row);
                for (uint32_t x = 1; x < s.width; ++x) {
                    acc = (uint16_t)(acc + u16(row + 2 * (size_t)x));
                    put16(row + 2 * (size_t)x, acc);
                }
            } else {
                uint32_t acc = u32(row);
                for (uint32_t x = 1; x < s.width; ++x) {
                    acc += u32(row + 4 * (size_t)x);
                    put32(row + 4 * (size_t)x, acc);
                }
            }
        }
    }
    // Predictor 3 (TIFF Technical Note 3): per row the bytes of the floats are stored planar, most significant byte plane
    // first, and byte-wise differenced over the whole row.  Result in the file's byte order.
    void undo_float_predictor(unsigned char *dst, const Slice &s) const {
        const size_t w = s.width, n = w * 4;
        std::vector<unsigned char> tmp(n);
        for (uint32_t y = 0; y < s.height; ++y) {
            unsigned char *row = dst + (size_t)y * n;
            for (size_t k = 1; k < n; ++k) row[k] = (unsigned char)(row[k] + row[k - 1]);
            memcpy(tmp.data(), row, n);
            for (size_t x = 0; x < w; ++x)
                for (size_t b = 0; b < 4; ++b)  // plane b holds byte b counted from the most significant one
                    row[4 * x + (big_endian_ ? b : 3 - b)] = tmp[b * w + x];
        }
    }
    void put16(unsigned char *p, uint16_t v) const {
        if (big_endian_) { p[0] = (unsigned char)(v >> 8); p[1] = (unsigned char)v; }
        else { p[1] = (unsigned char)(v >> 8); p[0] = (unsigned char)v; }
    }
    void put32(unsigned char *p, uint32_t v) const {
        if (big_endian_) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }
        else { p[3] = (unsigned char)(v >> 24); p[2] = (unsigned char)(v >> 16); p[1] = (unsigned char)(v >> 8); p[0] = (unsigned char)v; }
    }

    FILE *f_;
    std::string path_;
    bool big_endian_ = false, big_ = false;
    uint64_t file_size_ = 0;
    std::vector<Slice> slices_;
    std::vector<unsigned char> packed_;  // one compressed strip
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
    // big: BigTIFF (magic 43, 64-bit offsets) -- chosen by the caller when the stack would not fit classic TIFF's 4 GiB
    TiffWriter(const char *path, bool big) : f_(fopen(path, "wb")), path_(path), big_(big) {
        if (!f_) {
            fprintf(stderr, "*** Failed to create file!!! Please check the directory: %s\n", path);
            exit(1);  // reference src/apifunc.cpp:248-251
        }
        if (big_) {
            const unsigned char hdr[16] = {'I', 'I', 43, 0, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            put(hdr, 16);
            link_pos_ = 8;
            pos_ = 16;
        } else {
            const unsigned char hdr[8] = {'I', 'I', 42, 0, 0, 0, 0, 0};
            put(hdr, 8);
            link_pos_ = 4;
            pos_ = 8;
        }
    }
    ~TiffWriter() { close(); }
    // whether `slices` slices of width x height x bits need BigTIFF (data + directories beyond 4 GiB - 1 MiB)
    static bool needs_big(uint64_t width, uint64_t height, uint64_t slices, uint64_t bits) {
        return width * height * (bits / 8) * slices + slices * 512 + 16 > 0xFFF00000ull;
    }
    // flushes and closes; a short write (full disk) is an error, not a silently truncated stack
    void close() {
        if (!f_) return;
        FILE *f = f_;
        f_ = nullptr;
        if (fclose(f) != 0) fail();
    }
    void add_slice(const void *data, uint32_t width, uint32_t height, uint16_t bits, bool ieee_float) {
        const uint64_t nbytes = (uint64_t)width * height * (bits / 8);
        if (!big_ && pos_ + nbytes + 256 > 0xFFFFFFFFull) {
            fprintf(stderr, "*** TIFF write error: stack exceeds the 4 GiB classic-TIFF limit\n");
            exit(1);
        }
        const uint64_t data_off = pos_;
        put(data, nbytes);
        pos_ += nbytes;
        if (pos_ & 1) {
            const unsigned char zero = 0;
            put(&zero, 1);
            ++pos_;
        }
        struct Entry { uint16_t tag, type; uint64_t value; };  // count is 1 throughout
        const uint16_t off_type = big_ ? 16 : 4;                // LONG8 / LONG
        std::vector<Entry> e = {
            {256, 4, width}, {257, 4, height}, {258, 3, bits}, {259, 3, 1},
            {262, 3, 1},     {273, off_type, data_off}, {274, 3, 1},  {277, 3, 1},
            {278, 4, height}, {279, off_type, nbytes}, {284, 3, 2},
        };
        if (ieee_float) e.push_back({339, 3, 3});
        const uint64_t ifd_off = pos_;
        patch_link(ifd_off);
        if (big_) put64(e.size());
        else put16((uint16_t)e.size());
        for (const Entry &x : e) {
            put16(x.tag);
            put16(x.type);
            if (big_) {
                put64(1);
                put64(x.value);  // values sit left-justified: little-endian low bytes first
            } else {
                put32(1);
                put32((uint32_t)x.value);
            }
        }
        link_pos_ = pos_;
        if (big_) put64(0);
        else put32(0);
    }

private:
    [[noreturn]] void fail() const {
        fprintf(stderr, "*** TIFF write error (disk full or I/O error): %s\n", path_.c_str());
        exit(1);
    }
    void put(const void *data, size_t n) {
        if (fwrite(data, 1, n, f_) != n) fail();
    }
    void put_le(uint64_t v, int n) {
        unsigned char b[8];
        for (int i = 0; i < n; ++i) b[i] = (unsigned char)(v >> (8 * i));
        put(b, (size_t)n);
        pos_ += (uint64_t)n;
    }
    void put16(uint16_t v) { put_le(v, 2); }
    void put32(uint32_t v) { put_le(v, 4); }
    void put64(uint64_t v) { put_le(v, 8); }
    void patch_link(uint64_t v) {
        if (fseeko(f_, (off_t)link_pos_, SEEK_SET) != 0) fail();
        const uint64_t keep = pos_;
        put_le(v, big_ ? 8 : 4);
        pos_ = keep;
        if (fseeko(f_, (off_t)pos_, SEEK_SET) != 0) fail();
    }
    FILE *f_;
    std::string path_;
    bool big_ = false;
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
    // MIL_TIFF_BIG=1 forces BigTIFF (tests); otherwise only when the stack does not fit classic TIFF
    const char *force = getenv("MIL_TIFF_BIG");
    TiffWriter w(tifdir, (force && force[0] == '1') || TiffWriter::needs_big(imsize[0], imsize[1], imsize[2], bitPerSample == 16 ? 16 : 32));
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
    const char *force = getenv("MIL_TIFF_BIG");
    TiffWriter w(tifdir, (force && force[0] == '1') || TiffWriter::needs_big(imsize[0], imsize[1], imsize[2], 16));
    for (unsigned int n = 0; n < imsize[2]; ++n)
        w.add_slice(h_Image + (size_t)n * plane, imsize[0], imsize[1], 16, false);
    w.close();
}
