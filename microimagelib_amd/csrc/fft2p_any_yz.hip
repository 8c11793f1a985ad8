// Two-pass path: the LDS-resident yz kernel for every (ny, nz) whose tile of T = ny / 64 rows x 2 .. 16 columns x nz fits the LDS
// (T x nz <= 4096; ny = 128 .. 1024, nz = 64 .. 1024, multiples of 64), any nx -- the powers of two included: fft2p.hip keeps
// only the register kernel of 512 x 256 --; and the partial classic y passes for the
// 64 k extents (see fft2p_any.h).  yz_lds_kernel<NY, NZ, 0, COLS, PAIR> of fft2p_dev.h: the row geometry (pitch, 16-column
// groups) is a run-time argument, one instantiation per (ny, nz).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "fft2p_any_yz.hip is written for gfx950 (160 KB of LDS per workgroup): build with --offload-arch=gfx950"
#endif
#include "fft2p_any.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

namespace mil {
namespace fx {

namespace {

#include "fft2p_dev.h"

#ifndef MIL_YZ_MAX_TILE
#define MIL_YZ_MAX_TILE 4096
#endif
constexpr int kMaxTile = MIL_YZ_MAX_TILE;  // T x nz: 2 columns x 4096 complex = 64 KB of LDS

// Columns per tile, by the size of one column's share T x nz (complex values; profiles/r05/twopass_any_variants.txt):
//   < 512   16 columns: whole 128-byte row segments, the workgroup still has enough work
//   else     8 columns if a CU can hold two such workgroups (T x nz up to ~1100), the two halves of a 16-column group paired
//            on one XCD
//   else     4 columns, four workgroups paired: 8 columns would leave ONE workgroup per CU (no overlap of its load /
//            transform / store phases): 384 x 256, 165 us with 8 columns, 132 us with 4
//   else     2 columns (T x nz beyond ~2400), eight workgroups paired: 16-byte row segments, and still 1.1 - 1.3 x the classic
//            sequence (512 x 512 x 512: 491 -> 549 it/s; 256 x 1024 x 1024: 204 -> 252)
constexpr int cols_for(int t, int nz) {
#ifdef MIL_YZ_FORCE_COLS_2048  // experiments: tile width at T x nz = 2048
    if (t * nz == 2048) return MIL_YZ_FORCE_COLS_2048;
#endif
    return t * nz < 512 ? 16 : 2 * yz_lds_bytes(t, nz, 8) <= 163840 ? 8 : 2 * yz_lds_bytes(t, nz, 4) <= 163840 ? 4 : 2;
}
// The OTF block is requested before the forward transform where the registers for it do not cost a workgroup per CU: tiles of
// which a CU holds two anyway (128 registers per thread: T x nz > 1706, and the 8-column tiles from T x nz ~ 600) and whose block
// is at most 8 float4 per thread.  512 x 256: 191 -> 173 us; 384 x 256 (three workgroups per CU, 80 registers): 126 -> 133 us, so not there.
constexpr bool pref_for(int t, int nz) {
    const int cols = cols_for(t, nz), wgs = 163840 / yz_lds_bytes(t, nz, cols);
    return wgs <= 2 && t * nz * (cols / 2) <= 8 * kYzThreads;
}

template <int T, int NZ>
void launch_yz(const YzArgs &a, const float2 *tw_z, int xs, hipStream_t s) {
    constexpr int COLS = cols_for(T, NZ);
    const dim3 grid((unsigned)(64 * (xs / COLS) + a.pair_blocks)), block(kYzThreads);
    hipLaunchKernelGGL((yz_lds_kernel<64 * T, NZ, 0, COLS, (COLS < kW), pref_for(T, NZ)>), grid, block, 0, s, a, tw_z);
}

// (T, nz = 64 J): T = 2 .. 16, J = 1 .. 16
template <int T, int... J>
bool dispatch_nz(int nz, const YzArgs &a, const float2 *tw_z, int xs, hipStream_t s, std::integer_sequence<int, J...>) {
    bool done = false;
    auto one = [&](auto j_tag) {
        constexpr int NZ = 64 * (decltype(j_tag)::value + 1);
        if constexpr (T * NZ <= kMaxTile) {
            if (nz == NZ) {
                launch_yz<T, NZ>(a, tw_z, xs, s);
                done = true;
            }
        }
    };
    (one(std::integral_constant<int, J>{}), ...);
    return done;
}
template <int... I>
bool dispatch_yz(int t, int nz, const YzArgs &a, const float2 *tw_z, int xs, hipStream_t s, std::integer_sequence<int, I...>) {
    bool done = false;
    ((t == 2 + I ? (done = dispatch_nz<2 + I>(nz, a, tw_z, xs, s, std::make_integer_sequence<int, 16>{})) : false), ...);
    return done;
}

template <int N>
void launch_part(const PartArgs &a, int part, unsigned blocks, hipStream_t s) {
    constexpr int NT = N >= 320 ? 512 : 256;
    const dim3 grid(blocks), block(NT);
    switch (part) {
        case 0: hipLaunchKernelGGL((col_part_kernel<N, NT, 0>), grid, block, 0, s, a); break;
        case 1: hipLaunchKernelGGL((col_part_kernel<N, NT, 1>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((col_part_kernel<N, NT, 2>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((col_part_kernel<N, NT, 3>), grid, block, 0, s, a); break;
    }
}
template <int... I>
bool dispatch_part(int ny, const PartArgs &a, int part, unsigned blocks, hipStream_t s, std::integer_sequence<int, I...>) {
    bool done = false;
    auto one = [&](auto i_tag) {
        constexpr int N = 64 * (decltype(i_tag)::value + 3);  // 192 .. 960
        if constexpr ((N & (N - 1)) != 0) {                   // the powers of two live in fft2p.hip
            if (ny == N) {
                launch_part<N>(a, part, blocks, s);
                done = true;
            }
        }
    };
    (one(std::integral_constant<int, I>{}), ...);
    return done;
}

}  // namespace

namespace any {

bool yz_covers(const Box &b) {
    if (b.ny % 64 || b.ny < 128 || b.ny > 1024 || b.nz % 64 || b.nz < 64 || b.nz > 1024) return false;
    if (b.xs != b.nx / 2 || b.xs % kW || b.pitch != b.xs + kW) return false;
    return (b.ny / 64) * b.nz <= kMaxTile;
}

int yz_cols(const Box &b) { return cols_for(b.ny / 64, b.nz); }

void yz_fill_tables(const float2 *tw_y_dev, int ny) { fill_tw_tables(tw_y_dev, ny, nullptr, 0); }

void yz_conv(const Box &b, float2 *S, const float2 *otf, const unsigned short *neg_y, const unsigned short *neg_z,
             const unsigned short *pairs, int n_pairs, const float2 *tw_z, hipStream_t s) {
    if (!yz_covers(b)) throw Error(-1, "two-pass path: yz kernel does not cover this box");
    YzArgs a;
    a.S = S; a.otf = otf; a.otf_q = otf + (size_t)b.xs * b.ny * b.nz;
    a.neg_y = neg_y; a.neg_z = neg_z; a.pairs = pairs;
    a.pair_blocks = n_pairs;
    a.stamps = nullptr;
    a.pitch = b.pitch; a.groups = b.xs / kW;
    // OTF loads with the stream hint from 96 MB per volume: at 64 MB the two OTFs of a view still survive in the Infinity Cache
    // from one iteration to the next (hinted there: -1 .. -3 % on 512 x 512 x 64, 512 x 256 x 128, 256^3; +1 % from 128 MB)
    a.otf_stream = (long long)b.nx * b.ny * b.nz * 4 >= 2 * kStreamVolumeBytes;
    if (const char *e = getenv("MIL_FX_STREAM")) a.otf_stream = e[0] == '1';
    if (!dispatch_yz(b.ny / 64, b.nz, a, tw_z, b.xs, s, std::make_integer_sequence<int, 15>{}))
        throw Error(-1, "two-pass path: no yz kernel for this (ny, nz)");
    MIL_HIP(hipGetLastError());
}

bool y_part_covers(int ny) { return ny % 64 == 0 && ny >= 192 && ny <= 960 && (ny & (ny - 1)) != 0; }

void y_part(const Box &b, float2 *S, const float2 *tw_y, float2 *col0, int part, hipStream_t s) {
    if (part < 0 || part > 3) throw Error(-1, "y_part: unknown part");
    PartArgs a;
    a.S = S; a.tw = tw_y; a.col0 = (part == 2) ? col0 : nullptr;
    a.row_stride = b.pitch; a.outer_stride = (long long)b.ny * b.pitch; a.groups = b.xs / kW;
    if (!dispatch_part(b.ny, a, part, (unsigned)(a.groups * b.nz), s, std::make_integer_sequence<int, 13>{}))
        throw Error(-1, "two-pass path: no partial y pass for this ny");
    MIL_HIP(hipGetLastError());
}

}  // namespace any
}  // namespace fx
}  // namespace mil
