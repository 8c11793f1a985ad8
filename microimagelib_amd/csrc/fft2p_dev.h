// Device side of the two-pass path (csrc/fft2p.hip: powers of two; csrc/fft2p_any.hip: every other box of the path).
// Included inside namespace mil::fx { namespace { ... } } by both files; each translation unit owns its copy of the
// constant tables below and fills it through fill_tw_tables().
#include "fft_dev.h"

constexpr int kNY = 512, kNZ = 256;  // the register yz kernel's y and z extents
constexpr long long kStreamVolumeBytes = 48ll << 20;  // volumes from here on run the STREAM instantiations (see Aux): measured
                                                      // +2 .. +7 % at 64 - 256 MB, +-1 % at 32 and at 512 MB (profiles/r05/nontemporal_streams.txt)
// NH = nx / 2 spectrum columns (template parameter of every kernel): row pitch NH + 16 (ConvPlan: columns + 16), NH / 16
// 16-column groups per row
template <int NH>
struct Geo {
    static_assert(NH % 32 == 0 && NH >= 96 && NH <= 512, "two-pass path: nx must be a multiple of 64 in 192 .. 1024");
    static constexpr int pitch = NH + kW, groups = NH / kW;
};

// Buffer addressing: one 32-bit lane offset + a scalar offset per access, so that the 64 accesses of a register tile
// share ONE address register instead of 64 pointer pairs.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// Streams and the 256 MB Infinity Cache.  AUX = the cache-policy bits of a buffer access (gfx942 / gfx950: 1 = sc0, 2 = nt,
// 16 = sc1).  An iteration moves ~3.5 GB through a memory-side cache of 256 MB; what profits from it is the SPECTRUM, written
// by one kernel and read back by the next (268 MB at 512 x 512 x 256).  Everything touched once per iteration -- the OTFs, the
// measured image, the estimate -- only pushes spectrum lines out, so in the STREAM instantiations of the register kernels those
// accesses carry the non-temporal hint: 512 x 512 x 256 on one box 1 240 it/s without, 1 295 with the OTF and image / estimate
// loads marked, 1 361 with the estimate's store marked too; marking the SPECTRUM instead loses 5-6 %
// (profiles/r05/nontemporal_streams.txt).  Boxes whose whole working set fits the cache (volumes of a few ten MB) lose 4-7 %
// with the hint -- their OTFs and images would have stayed -- so the host picks the instantiation by volume size
// (ConvPlan::stream_, MIL_FX_STREAM=0 / 1 forces it).
template <bool STREAM>
struct Aux {
    static constexpr int ld_spec = 0, st_spec = 0;
    static constexpr int ld_otf = STREAM ? 2 : 0, ld_real = STREAM ? 2 : 0, st_real = STREAM ? 2 : 0;
};
template <int AUX = 0>
__device__ __forceinline__ float2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2 x) {
    u32x2 v;
    v.x = __float_as_uint(x.x);
    v.y = __float_as_uint(x.y);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, AUX);
}
// Twiddles of the in-register stages: their index is uniform over a wave, and from constant memory that is a scalar
// load into SGPRs (through a kernel argument the compiler cannot prove the table read-only and fetches every twiddle
// into a vector register: +40 registers, spills).  Filled by ConvPlan::init_two_pass on the current device.
// One table per y extent: engines of different boxes may live on the same device (host threads), and each writes only
// its own table -- with the same values every time.
// One table per y extent ny = 64 T (T = 2 .. 16), back to back: engines of different boxes may live on the same device
// (host threads), and each writes only its own table -- with the same values every time.
constexpr int tw_y_offset(int T) { return 32 * T * (T - 1) - 64; }  // 64 * (2 + 3 + ... + (T - 1))
__constant__ float2 c_tw_y[tw_y_offset(17)];
__constant__ float2 c_tw_z[kNZ];
template <int NY>
__device__ __forceinline__ const float2 *tw_y_const() {
    static_assert(NY % 64 == 0 && NY >= 128 && NY <= 1024, "two-pass path: y extent");
    return c_tw_y + tw_y_offset(NY / 64);
}
// host: copy the plan's twiddles (device arrays of ny / nz entries) into this translation unit's tables
inline void fill_tw_tables(const float2 *tw_y_dev, int ny, const float2 *tw_z_dev, int nz) {
    std::vector<float2> t((size_t)ny);
    MIL_HIP(hipMemcpy(t.data(), tw_y_dev, t.size() * sizeof(float2), hipMemcpyDeviceToHost));
    MIL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tw_y), t.data(), t.size() * sizeof(float2), (size_t)tw_y_offset(ny / 64) * sizeof(float2)));
    if (tw_z_dev && nz == kNZ) {  // the register yz kernel reads its z twiddles from constant memory
        std::vector<float2> tz((size_t)kNZ);
        MIL_HIP(hipMemcpy(tz.data(), tw_z_dev, tz.size() * sizeof(float2), hipMemcpyDeviceToHost));
        MIL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tw_z), tz.data(), tz.size() * sizeof(float2)));
    }
}

// The y tail: the in-place transform of length T = NY / 64 over T adjacent rows, in the classic schedule (pick_radix) so
// that every frequency lands where the classic kernels leave it.  T <= 8 and the primes: one radix-T stage (natural
// order).  T = 16: radix 8 then radix 2, frequency f = k1 + 8 k2 at row 2 k1 + k2 (one dft16 + a permutation).  The
// other composites (6, 9, 10, 12, 14, 15 = R1 x M1 with M1 prime): the two classic stages on registers, twiddles
// w_T^(i k) = w_NY^(64 i k) from the constant table (compile-time indices: scalar loads).
// tail_forward: rows in natural order -> frequencies at their classic positions; tail_inverse: the reverse.
template <int T>
constexpr bool tail_two_stage() {
    return T != 16 && pick_radix(T) != T;
}
template <int T>
__device__ __forceinline__ int tail_pos(int f) {
    return T == 16 ? 2 * (f & 7) + (f >> 3) : f;
}
template <int T>
__device__ __forceinline__ void tail_forward(float2 *p, int stride) {
    float2 t[T];
#pragma unroll
    for (int k = 0; k < T; ++k) t[k] = p[k * stride];
    if constexpr (tail_two_stage<T>()) {
        constexpr int R1 = pick_radix(T), M1 = T / R1;
        const float2 *tw = tw_y_const<64 * T>();
#pragma unroll
        for (int i = 0; i < M1; ++i) {
            float2 v[R1];
#pragma unroll
            for (int j = 0; j < R1; ++j) v[j] = t[i + j * M1];
            dft_small<R1, -1>(v);
#pragma unroll
            for (int k = 1; k < R1; ++k)
                if (i > 0) v[k] = tw_mul<-1>(v[k], tw[64 * i * k]);
#pragma unroll
            for (int j = 0; j < R1; ++j) t[i + j * M1] = v[j];
        }
#pragma unroll
        for (int b = 0; b < R1; ++b) dft_small<M1, -1>(t + b * M1);
#pragma unroll
        for (int k = 0; k < T; ++k) p[k * stride] = t[k];
    } else {
        dft_small<T, -1>(t);
#pragma unroll
        for (int k = 0; k < T; ++k) p[tail_pos<T>(k) * stride] = t[k];
    }
}
template <int T>
__device__ __forceinline__ void tail_inverse(float2 *p, int stride) {
    float2 t[T];
    if constexpr (tail_two_stage<T>()) {
        constexpr int R1 = pick_radix(T), M1 = T / R1;
        const float2 *tw = tw_y_const<64 * T>();
#pragma unroll
        for (int k = 0; k < T; ++k) t[k] = p[k * stride];
#pragma unroll
        for (int b = 0; b < R1; ++b) dft_small<M1, +1>(t + b * M1);
#pragma unroll
        for (int i = 0; i < M1; ++i) {
            float2 v[R1];
#pragma unroll
            for (int j = 0; j < R1; ++j) v[j] = t[i + j * M1];
#pragma unroll
            for (int k = 1; k < R1; ++k)
                if (i > 0) v[k] = tw_mul<+1>(v[k], tw[64 * i * k]);
            dft_small<R1, +1>(v);
#pragma unroll
            for (int j = 0; j < R1; ++j) t[i + j * M1] = v[j];
        }
    } else {
#pragma unroll
        for (int k = 0; k < T; ++k) t[k] = p[tail_pos<T>(k) * stride];
        dft_small<T, +1>(t);
    }
#pragma unroll
    for (int k = 0; k < T; ++k) p[k * stride] = t[k];
}


// ---------------------------------------------------------------------------
// partial y passes (classic column tiles): head = the first two (radix-8) stages, tail = the sub-transforms of length T = N / 64
// ---------------------------------------------------------------------------
struct PartArgs {
    float2 *S;
    const float2 *tw;
    float2 *col0;  // forward tail: copy of the completed packed column, as the classic forward y pass writes it
    long long row_stride, outer_stride;
    int groups;
};

template <int N, int NT, int PART>
__global__ __launch_bounds__(NT) void col_part_kernel(PartArgs a) {
    constexpr int T = N / 64;
    __shared__ float2 tile[N * kWP];
    __shared__ float2 tw[N];
    const int g = blockIdx.x % a.groups;
    const long long outer = blockIdx.x / a.groups;
    float2 *base = a.S + outer * a.outer_stride + (long long)g * kW;
    for (int k = threadIdx.x; k < N; k += NT) tw[k] = a.tw[k];
    tile_load<N, NT>(tile, base, a.row_stride);
    lds_barrier();
    if constexpr (PART == 0) fft_dif<N, N, -1, kW, NT, kWP, T>(tile, tw);
    else if constexpr (PART == 1) fft_dit<N, N, +1, kW, NT, T>(tile, tw);
    else if constexpr (PART == 2) fft_dif<N, T, -1, kW, NT, kWP>(tile, tw);
    else fft_dit<N, T, +1, kW, NT>(tile, tw);
    tile_store<N, NT>(tile, base, a.row_stride);
    if constexpr (PART == 2) {
        if (g == 0 && a.col0)
            for (int k = threadIdx.x; k < N; k += NT) a.col0[outer * N + k] = tile[k * kWP];
    }
}

// ---------------------------------------------------------------------------
// yz_conv: y stage(8), z, OTF, z^-1, y stage(8)^-1 on 8 rows x 16 columns x 256 z held in registers
// ---------------------------------------------------------------------------
struct YzArgs {
    float2 *S;
    const float2 *otf;    // [row][column group][kz][columns of the calling kernel's tile] (ConvPlan::set_yz_otf_layout)
    const float2 *otf_q;  // [row][kz]
    const unsigned short *neg_y, *neg_z;
    const unsigned short *pairs;  // [pair][2]: row blocks holding ky and -ky (packed column)
    int pair_blocks;              // the first blocks do the packed column, one pair of row blocks each (short: they
                                  // start first so that the long tiles fill in behind them instead of waiting at the end)
    unsigned long long *stamps;   // timing experiments: 8 wall-clock stamps (100 MHz) per workgroup, or null
    int pitch, groups;            // row pitch (NH + 16) and 16-column groups per row, for the kernels built with NH = 0
    int otf_stream;               // yz_lds_kernel: read the OTF with the non-temporal hint (volumes far beyond the Infinity Cache)
};

// NH = 0: the row geometry comes from the arguments (one instantiation per (ny, nz) serves every nx)
template <int NH>
__device__ __forceinline__ int yz_pitch(const YzArgs &a) {
    if constexpr (NH == 0) return a.pitch;
    else return Geo<NH>::pitch;
}
template <int NH>
__device__ __forceinline__ int yz_groups(const YzArgs &a) {
    if constexpr (NH == 0) return a.groups;
    else return Geo<NH>::groups;
}

#ifdef MIL_FX_STAMPS  // timing experiments: wall-clock stamps at the phase boundaries of one workgroup (tools/twopass_stamps.py)
#define MIL_STAMP(i)                                                                   \
    do {                                                                               \
        if (a.stamps && threadIdx.x == 0) a.stamps[(blockIdx.x % 2048) * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define MIL_STAMP(i)
#endif

constexpr int kYzThreads = 512;
constexpr int kQuarter = 16;             // registers exchanged at a time (a quarter of the thread's 64)
constexpr int kPark = 23;                // values of the waiting quarters parked in the rest of the 160 KB

// value of the neighbouring lane (lane ^ 1)
__device__ __forceinline__ float2 swap_pair(float2 a) {
    const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(a.x), 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]
    const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(a.y), 0xB1, 0xF, 0xF, false);
    return make_float2(__int_as_float(x), __int_as_float(y));
}

// ---- packed DC/Nyquist column: G = F0 + i FN needs conj(G(-ky, -kz)), which lives in the row block of -ky.
// One workgroup per pair of row blocks, 2 x T rows x NZ z of column 0 in LDS as tile[z][2 T + 1].  `tw`: w_NZ^k, any address
// space the 512 threads may index freely.
// OC: columns per group of the OTF tiling (= the calling kernel's tile width): column 0 of group 0 is element 0 of [kz][OC].
template <int NY, int NZ, int NH, int OC, int NT = 512>
__device__ __forceinline__ void packed_column_block(const YzArgs &a, float2 *tile, const float2 *tw, int pair) {
    constexpr int T = NY / 64, C = 2 * T;  // rows per block = columns of the tile
    constexpr int PW = C, PWP = PW + 1;    // tile[z][PW + 1]
    const int kPitch = yz_pitch<NH>(a), kGroups = yz_groups<NH>(a);
    const int tid = threadIdx.x;
    const int blk[2] = {a.pairs[2 * pair], a.pairs[2 * pair + 1]};
    for (int i = tid; i < NZ * C; i += NT) {
        const int c = i % C, z = i / C;
        tile[z * PWP + c] = a.S[((long long)z * NY + T * blk[c / T] + (c % T)) * kPitch];
    }
    lds_barrier();
    for (int i = tid; i < NZ * 2; i += NT) tail_forward<T>(tile + (i >> 1) * PWP + T * (i & 1), 1);
    lds_barrier();
    fft_dif<NZ, NZ, -1, PW, NT, PWP>(tile, tw);
    constexpr int PER = (NZ * C + NT - 1) / NT;
    float2 r[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = i * NT + tid, c = e % C, p = e / C;
        if (e < NZ * C) {
            const int row = T * blk[c / T] + (c % T);
            const int prow = a.neg_y[row];
            const int pc = ((prow / T) == blk[0] ? 0 : T) + (prow % T);
            const float2 gk = tile[p * PWP + c], gp = tile[a.neg_z[p] * PWP + pc];
            const float2 pv = a.otf[(((long long)row * kGroups * (kW / OC)) * NZ + p) * OC];
            const float2 qv = a.otf_q[(long long)row * NZ + p];
            r[i] = c_add(c_mul(gk, pv), c_mul(qv, make_float2(gp.x, -gp.y)));
        }
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = i * NT + tid;
        if (e < NZ * C) tile[(e / C) * PWP + (e % C)] = r[i];
    }
    lds_barrier();
    fft_dit<NZ, NZ, +1, PW, NT>(tile, tw);
    for (int i = tid; i < NZ * 2; i += NT) tail_inverse<T>(tile + (i >> 1) * PWP + T * (i & 1), 1);
    lds_barrier();
    const int ncol = (blk[0] == blk[1]) ? T : C;
    for (int i = tid; i < NZ * C; i += NT) {
        const int c = i % C, z = i / C;
        if (c < ncol) a.S[((long long)z * NY + T * blk[c / T] + (c % T)) * kPitch] = tile[z * PWP + c];
    }
}

// COLS = 16: 512 threads, one workgroup per CU (the 256 KB tile).  COLS = 8: 256 threads and half the tile, TWO workgroups
// per CU whose load / exchange / store phases overlap; the two halves of every 128-byte row segment go to workgroups t and
// t + 8 -- the same XCD, a few dispatches apart -- so that HBM still sees whole lines (unpaired, the 64-byte segments cost
// what the overlap gains: round 2).
template <int NH, int COLS, bool STREAM>
__global__ __launch_bounds__(32 * COLS, 2) void yz_conv_kernel(YzArgs a) {
    static_assert(COLS == 16 || COLS == 8, "yz_conv_kernel: 16- or 8-column tiles");
    constexpr int THREADS = 32 * COLS;           // COLS columns x 8 rows x 4 z-residues
    constexpr int kXchg = 4 * 8 * COLS + COLS;   // float2 between consecutive exchanged registers: 4 z-residues x 8 rows x COLS
                                                 // columns (+ COLS: the register slots a wave reads at once fall on different bank halves)
    constexpr int kPitch = Geo<NH>::pitch;
    constexpr long long kSpecBytes = (long long)kPitch * kNY * kNZ * 8, kOtfBytes = (long long)NH * kNY * kNZ * 8;
    // exchange buffer of one quarter (66 KB) + parking space (92 KB): registers are the scarce resource of this kernel,
    // every spilled one is HBM traffic
    __shared__ float2 lds[kQuarter * kXchg + kPark * THREADS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float2 *twz = c_tw_z;  // in-register stages; the packed-column path below uses the same table through LDS-style code
    if ((int)blockIdx.x >= a.pair_blocks) {
        const int tile_id = blockIdx.x - a.pair_blocks;
        int g = tile_id % (NH / COLS), b = tile_id / (NH / COLS);  // g: COLS-column group
        if constexpr (COLS == 8) {  // XCD pairing of the two halves of a 16-column group (see yz_lds_kernel)
            const int tt = tile_id % 16, q = tile_id / 16;
            g = 2 * ((q % (NH / 128)) * 8 + (tt % 8)) + tt / 8;
            b = q / (NH / 128);
        }
        // ownership 1: all z = za + 4m (m = j2 + 8 j1) of one (row j, column)
        const int col = lane & (COLS - 1), za = wave & 3, j = (lane / COLS) + (64 / COLS) * (wave >> 2);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.S, kSpecBytes), ro = make_rsrc(a.otf, kOtfBytes);
        const unsigned voff = (unsigned)(j * kPitch + col) * 8u;
        const unsigned soff0 = (unsigned)((za * kNY + 8 * b) * kPitch + g * COLS) * 8u;  // + m * zstep
        constexpr unsigned zstep = 4u * kNY * kPitch * 8u;
        // ownership 2 (per quarter q: ka = 2q, 2q + 1): a PAIR of lanes shares the 8 rows x 4 z-residues of one (register
        // slot kk = kb + 8 (ka - 2q), column); lane sp of the pair holds rows sp, sp + 2, sp + 4, sp + 6 going in and the
        // row frequencies 4 sp .. 4 sp + 3 at the OTF product (the radix-8 row butterfly split over the pair)
        const int sp = tid & 1, scol = (tid >> 1) & (COLS - 1), skk = tid / (2 * COLS);
        // OTF: [ky][COLS-column group][kz][COLS] -- this workgroup's block of a row is kNZ * COLS contiguous values
        constexpr int kOGroups = NH / COLS;
        const unsigned ooff0 = (unsigned)((((8 * b) * kOGroups + g) * kNZ) * COLS) * 8u;  // wave-uniform part (scalar offset)
        const unsigned ovoff = (unsigned)((4 * sp * kOGroups * kNZ + 4 * (skk & 7) + 32 * (skk >> 3)) * COLS + scol) * 8u;  // row 4 sp, position 4 kb + 32 (ka - 2q)
        float2 v[64], o[2][16];
        auto load_otf = [&](int q) {  // position kc + 4 kb + 32 ka of row frequency 4 sp + k
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
                    o[q & 1][kc * 4 + k] = buf_load2<Aux<STREAM>::ld_otf>(ro, ovoff, ooff0 + (unsigned)((k * kOGroups * kNZ + 64 * q + kc) * COLS) * 8u);
        };
        MIL_STAMP(0);
#pragma unroll
        for (int m = 0; m < 64; ++m) v[m] = buf_load2<Aux<STREAM>::ld_spec>(rs, voff, soff0 + m * zstep);
        load_otf(0);
        load_otf(1);
        // ---- z forward, stages L = 256 (over j1) and L = 32 (over j2), in registers
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) {
            float2 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = v[j2 + 8 * k];
            dft8<-1>(t);
#pragma unroll
            for (int k = 1; k < 8; ++k) t[k] = tw_mul<-1>(t[k], twz[(za + 4 * j2) * k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[j2 + 8 * k] = t[k];
        }
#pragma unroll
        for (int ka = 0; ka < 8; ++ka) {
            float2 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = v[k + 8 * ka];
            dft8<-1>(t);
#pragma unroll
            for (int k = 1; k < 8; ++k) t[k] = tw_mul<-1>(t[k], twz[za * k * 8]);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k + 8 * ka] = t[k];
        }
        MIL_STAMP(1);
        float2 *mine = lds + za * (8 * COLS) + j * COLS + col;         // + kk * kXchg
        float2 *slot = lds + skk * kXchg + sp * COLS + scol;            // + a * 8 * COLS + i * 2 * COLS   (row sp + 2 i)
        float2 *park = lds + kQuarter * kXchg + tid;                   // + i * THREADS, this thread's own slots
        // Parking plan (values by register index): v[32..54] wait in LDS from the start; when quarter 2 / 3 picks its inputs
        // up, the results of quarter 0 / 1 take the freed places; everything comes back after the last quarter.
#pragma unroll
        for (int i = 0; i < kPark; ++i) park[i * THREADS] = v[32 + i];
        constexpr float h8 = 0.70710678118654752440f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    v[32 + i] = park[i * THREADS];
                    park[i * THREADS] = v[i];  // results of quarter 0
                }
            }
            if (q == 3) {
#pragma unroll
                for (int i = 16; i < kPark; ++i) {
                    v[32 + i] = park[i * THREADS];
                    park[i * THREADS] = v[i];  // 7 results of quarter 1
                }
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) mine[kk * kXchg] = v[16 * q + kk];
            if (q == 0) MIL_STAMP(8);
            lds_barrier();
            if (q == 0) MIL_STAMP(9);
            float2 u[16];  // [a][i]: z-residue a, row sp + 2 i
#pragma unroll
            for (int e = 0; e < 16; ++e) u[e] = slot[(e >> 2) * (8 * COLS) + (e & 3) * (2 * COLS)];
            // y stage of length 8 over the rows, split over the lane pair (decimation in time): each lane transforms its four
            // rows, the odd lane applies w8^k, then  row k = even + odd,  row k + 4 = even - odd
#pragma unroll
            for (int za2 = 0; za2 < 4; ++za2) {
                float2 *w = u + 4 * za2;
                dft4<-1>(w[0], w[1], w[2], w[3]);
                if (sp) {
                    w[1] = add_i<-1>(w[1], w[1]) * make_float2(h8, h8);
                    w[2] = rot90<-1>(w[2]);
                    w[3] = add_i<+1>(w[3], w[3]) * make_float2(-h8, -h8);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float2 oth = swap_pair(w[k]);
                    w[k] = sp ? c_sub(oth, w[k]) : c_add(w[k], oth);
                }
            }
            // z stage of length 4 over the residues
#pragma unroll
            for (int k = 0; k < 4; ++k) dft4<-1>(u[k], u[4 + k], u[8 + k], u[12 + k]);
            if (q == 0) MIL_STAMP(10);
#pragma unroll
            for (int e = 0; e < 16; ++e) u[e] = c_mul(u[e], o[q & 1][e]);
            if (q == 0) MIL_STAMP(11);
            if (q + 2 < 4) load_otf(q + 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) dft4<+1>(u[k], u[4 + k], u[8 + k], u[12 + k]);
            // y stage back (decimation in frequency): s = x_r + x_(r+4) on the even lane, d = (x_r - x_(r+4)) w8^-r on the odd
            // one, then each lane transforms its four: even lane -> rows 0, 2, 4, 6, odd lane -> rows 1, 3, 5, 7
#pragma unroll
            for (int za2 = 0; za2 < 4; ++za2) {
                float2 *w = u + 4 * za2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float2 oth = swap_pair(w[k]);
                    w[k] = sp ? c_sub(oth, w[k]) : c_add(w[k], oth);
                }
                if (sp) {
                    w[1] = add_i<+1>(w[1], w[1]) * make_float2(h8, h8);
                    w[2] = rot90<+1>(w[2]);
                    w[3] = add_i<-1>(w[3], w[3]) * make_float2(-h8, -h8);
                }
                dft4<+1>(w[0], w[1], w[2], w[3]);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) slot[(e >> 2) * (8 * COLS) + (e & 3) * (2 * COLS)] = u[e];
            if (q == 0) MIL_STAMP(12);
            lds_barrier();
            if (q == 0) MIL_STAMP(13);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) v[16 * q + kk] = mine[kk * kXchg];
            if (q < 2) MIL_STAMP(2 + q);
        }
#pragma unroll
        for (int i = 0; i < kPark; ++i) v[i] = park[i * THREADS];
        // ---- z inverse, stages L = 32 and L = 256
#pragma unroll
        for (int ka = 0; ka < 8; ++ka) {
            float2 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = v[k + 8 * ka];
#pragma unroll
            for (int k = 1; k < 8; ++k) t[k] = tw_mul<+1>(t[k], twz[za * k * 8]);
            dft8<+1>(t);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k + 8 * ka] = t[k];
                }
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) {
            float2 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = v[j2 + 8 * k];
#pragma unroll
            for (int k = 1; k < 8; ++k) t[k] = tw_mul<+1>(t[k], twz[(za + 4 * j2) * k]);
            dft8<+1>(t);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[j2 + 8 * k] = t[k];
                }
        MIL_STAMP(4);
        if (g != 0 || col != 0) {  // the packed column belongs to the pair blocks below
#pragma unroll
            for (int m = 0; m < 64; ++m) buf_store2<Aux<STREAM>::st_spec>(rs, voff, soff0 + m * zstep, v[m]);
        }
        MIL_STAMP(5);
#ifdef MIL_FX_STAMPS
        if (a.stamps) {
            __builtin_amdgcn_s_waitcnt(0);
            MIL_STAMP(6);
        }
#endif
        return;
    }
    packed_column_block<kNY, kNZ, NH, COLS, THREADS>(a, lds, twz, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------
// yz_conv where the tile fits the LDS: T rows (T = NY / 64, the y tail) x COLS columns x NZ z <= 16384 complex = 128 KB
// + padding -- COLS = 16 (128-byte row segments) for the smallest tiles, 8 / 4 / 2 columns for the taller ones (the host's rule:
// fft2p_any_yz.hip) -- so the sequence
//   y tail(T) . z . OTF . z^-1 . y tail(T)^-1
// runs on tile[z][T rows x COLS columns] with the classic LDS stage code, T * COLS "columns" per z position.  The y tail rides
// in the load and in the store -- the T rows of a (z, column pair) in one thread's registers -- instead of making two more
// passes over the tile in LDS (576 x 128: 160 -> 132 us).  One trip of the spectrum through HBM (3 C'') where the classic
// kernels make three (y tail | z-conv | y tail^-1: 7 C'').
// ---------------------------------------------------------------------------
// LDS of one workgroup: tile[NZ][max(T COLS, 2 T) + 1] + NZ twiddles (complex)
constexpr int yz_lds_bytes(int t, int nz, int cols) { return (nz * ((t * cols > 2 * t ? t * cols : 2 * t) + 1) + nz) * 8; }
// workgroups of 512 threads a CU can hold by LDS, as waves per SIMD for __launch_bounds__: 2 per workgroup, at most 6 (80 registers)
template <int NY, int NZ, int COLS>
constexpr int yz_lds_occ() {
    constexpr int wgs = 163840 / yz_lds_bytes(NY / 64, NZ, COLS);
    return wgs >= 3 ? 6 : wgs == 2 ? 4 : 2;
}
template <int NY, int NZ, int NH, int COLS, bool XCDPAIR = false, bool PREF = false>
__global__ __launch_bounds__(kYzThreads, (yz_lds_occ<NY, NZ, COLS>())) void yz_lds_kernel(YzArgs a, const float2 *__restrict__ tw_z) {
    constexpr int T = NY / 64;
    static_assert(T * NZ * COLS <= 16384 && (COLS == 16 || COLS == 8 || COLS == 4 || COLS == 2), "yz_lds_kernel: tile does not fit the LDS");
    const int kPitch = yz_pitch<NH>(a), kGroups = yz_groups<NH>(a);
    constexpr int W = T * COLS, WP = W + 1;  // row r, column c of the block sit at "column" r * COLS + c
    constexpr int QR = COLS / 2;             // float4 pieces per row segment
    constexpr int PWP = 2 * T + 1;           // packed_column_block's tile
    __shared__ float2 tile[NZ * WP > NZ * PWP ? NZ * WP : NZ * PWP];
    __shared__ float2 tw[NZ];
    const int tid = threadIdx.x;
    for (int k = tid; k < NZ; k += kYzThreads) tw[k] = tw_z[k];
    if ((int)blockIdx.x < a.pair_blocks) {
        lds_barrier();
        packed_column_block<NY, NZ, NH, COLS>(a, tile, tw, (int)blockIdx.x);
        return;
    }
    const int GROUPS = kGroups * (kW / COLS);
    const int tile_id = blockIdx.x - a.pair_blocks;
    int g = tile_id % GROUPS, b = tile_id / GROUPS;
    if constexpr (XCDPAIR) {
        // Narrow tiles (8 / 4 columns): the 2 / 4 pieces of a 128-byte row segment go to workgroups t, t + 8 (, t + 16,
        // t + 24), i.e. to the SAME XCD (workgroups are dealt round-robin over the 8 XCDs) a few dispatches apart: the first
        // one brings the line into that XCD's L2, the others find it there -- HBM sees whole lines again.  (16-column group,
        // row block) pairs are numbered l = group + groups x block, eight of them per round of 8 SUB workgroups; 64 blocks
        // of T rows, so the count is a multiple of 8 whatever the number of groups.
        static_assert(!XCDPAIR || COLS < kW, "pairing is for narrow tiles");
        constexpr int SUB = kW / COLS;
        const int tt = tile_id % (8 * SUB), q = tile_id / (8 * SUB);
        const int l = 8 * q + (tt % 8);
        g = SUB * (l % kGroups) + tt / 8;
        b = l / kGroups;
    }
    float2 *base = a.S + (long long)(T * b) * kPitch + g * COLS;
    const long long zstride = (long long)NY * kPitch;
    // ---- load + y tail: a thread takes the T rows of a column PAIR (float4 accesses), or of one column where 2 T complex values +
    // their transform would not fit the registers of three workgroups per CU (T >= 10)
    constexpr bool kPairCols = T < 10;
    if constexpr (kPairCols) {
#pragma unroll 1
        for (int i = tid; i < NZ * QR; i += kYzThreads) {
            const int q = i % QR, z = i / QR;
            float4 v[T];
#pragma unroll
            for (int r = 0; r < T; ++r) v[r] = *reinterpret_cast<const float4 *>(base + z * zstride + (long long)r * kPitch + 2 * q);
            float2 c0[T], c1[T];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                c0[r] = make_float2(v[r].x, v[r].y);
                c1[r] = make_float2(v[r].z, v[r].w);
            }
            tail_forward<T>(c0, 1);
            tail_forward<T>(c1, 1);
            float2 *p = tile + z * WP + 2 * q;
#pragma unroll
            for (int r = 0; r < T; ++r) {
                p[r * COLS] = c0[r];
                p[r * COLS + 1] = c1[r];
            }
        }
    } else {
#pragma unroll 1
        for (int i = tid; i < NZ * COLS; i += kYzThreads) {
            const int c = i % COLS, z = i / COLS;
            float2 c0[T];
#pragma unroll
            for (int r = 0; r < T; ++r) c0[r] = base[z * zstride + (long long)r * kPitch + c];
            tail_forward<T>(c0, 1);
            float2 *p = tile + z * WP + c;
#pragma unroll
            for (int r = 0; r < T; ++r) p[r * COLS] = c0[r];
        }
    }
    // ---- OTF ([row][group of COLS columns][kz][COLS], ConvPlan::set_yz_otf_layout): PREF requests the workgroup's block now, so that it arrives
    // under the forward transform (PIECES / 512 float4 per thread held in registers until the product)
    constexpr int PIECES = T * NZ * QR;  // float4 pieces
    constexpr int OPER = PREF ? (PIECES + kYzThreads - 1) / kYzThreads : 1;
    float4 oreg[OPER];
    auto otf_piece = [&](int idx) {
        const int q = idx % QR, kz = (idx / QR) % NZ, row = idx / (QR * NZ);
        const float4 *src = reinterpret_cast<const float4 *>(a.otf + ((((long long)(T * b + row) * GROUPS + g) * NZ + kz) * COLS) + 2 * q);
        if (a.otf_stream) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src));
            return make_float4(v.x, v.y, v.z, v.w);
        }
        return *src;
    };
    if constexpr (PREF) {
#pragma unroll
        for (int j = 0; j < OPER; ++j) {
            const int idx = j * kYzThreads + tid;
            if (idx < PIECES) oreg[j] = otf_piece(idx);
        }
    }
    lds_barrier();
    fft_dif<NZ, NZ, -1, W, kYzThreads, WP>(tile, tw);
    if constexpr (PREF) {
#pragma unroll
        for (int j = 0; j < OPER; ++j) {
            const int idx = j * kYzThreads + tid;
            const int q = idx % QR, kz = (idx / QR) % NZ, row = idx / (QR * NZ);
            if (idx < PIECES) {
                float2 *p = tile + kz * WP + row * COLS + 2 * q;
                p[0] = c_mul(p[0], make_float2(oreg[j].x, oreg[j].y));
                p[1] = c_mul(p[1], make_float2(oreg[j].z, oreg[j].w));
            }
        }
    } else {
#pragma unroll 1
        for (int it0 = 0; it0 < PIECES; it0 += kYzThreads * kBatch) {
            float4 v[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int idx = it0 + j * kYzThreads + tid;
                // the stream hint only on large boxes (a.otf_stream): the OTFs of small ones survive in the Infinity Cache from one
                // iteration to the next
                if (idx < PIECES) v[j] = otf_piece(idx);
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int idx = it0 + j * kYzThreads + tid;
                const int q = idx % QR, kz = (idx / QR) % NZ, row = idx / (QR * NZ);
                if (idx < PIECES) {
                    float2 *p = tile + kz * WP + row * COLS + 2 * q;
                    p[0] = c_mul(p[0], make_float2(v[j].x, v[j].y));
                    p[1] = c_mul(p[1], make_float2(v[j].z, v[j].w));
                }
            }
        }
    }
    lds_barrier();
    fft_dit<NZ, NZ, +1, W, kYzThreads>(tile, tw);
    if constexpr (kPairCols) {
        // ---- y tail back and store; the packed column (group 0, column 0) belongs to the pair blocks
#pragma unroll 1
        for (int i = tid; i < NZ * QR; i += kYzThreads) {
            const int q = i % QR, z = i / QR;
            const float2 *p = tile + z * WP + 2 * q;
            float2 c0[T], c1[T];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                c0[r] = p[r * COLS];
                c1[r] = p[r * COLS + 1];
            }
            tail_inverse<T>(c0, 1);
            tail_inverse<T>(c1, 1);
            float2 *dst = base + z * zstride + 2 * q;
#pragma unroll
            for (int r = 0; r < T; ++r) {
                if (g == 0 && q == 0) dst[(long long)r * kPitch + 1] = c1[r];
                else *reinterpret_cast<float4 *>(dst + (long long)r * kPitch) = make_float4(c0[r].x, c0[r].y, c1[r].x, c1[r].y);
            }
        }
    } else {
#pragma unroll 1
        for (int i = tid; i < NZ * COLS; i += kYzThreads) {
            const int c = i % COLS, z = i / COLS;
            const float2 *p = tile + z * WP + c;
            float2 c0[T];
#pragma unroll
            for (int r = 0; r < T; ++r) c0[r] = p[r * COLS];
            tail_inverse<T>(c0, 1);
            if (g != 0 || c != 0) {
#pragma unroll
                for (int r = 0; r < T; ++r) base[z * zstride + (long long)r * kPitch + c] = c0[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// xy pass: the 64 rows y = n2 + T m (T = NY / 64) of one z plane in registers (thread = column); y head^-1, then per 16 (32) rows
// x^-1 . op . x through LDS (the classic x kernel's tile), then y head
// ---------------------------------------------------------------------------
struct XyArgs {
    float2 *S;
    float *E;        // estimate (in / out), update modes
    const float *M;  // measured image, ratio mode
    const float2 *tw_x;
    float eps;
    long long spec_bytes, real_bytes;  // extents of S and of the real volumes (buffer descriptors)
    unsigned long long *stamps;  // timing experiments: 8 wall-clock stamps (100 MHz) per workgroup, or null
    int t;                       // ny / 64, for the kernels built with NY = 0
};


// The x transform of length NH runs as NH = 16 x M: element k = a + 16 b (a < 16, b < M) of the half-length sequence sits at
// position a * M + pos_M(b), where pos_M is the in-place order of the length-M sub-transforms.  Every trip of a stage
// through the LDS costs the same whatever its radix, and the kernel is bound by the latency of its trips, so the whole
// length M = 6 .. 32 is ONE stage: radix M in registers (dft_comp: 12 = 4 x 3, 18 = 2 x 9, 24 = 4 x 6 ...; the primes 11, 13
// by definition), pos_M the identity.  Measured at 128 x 512 x nx against the classic schedules (12 = 4 . 3, 18 = 2 . 3 . 3,
// 32 = 8 . 4 in two or three trips): nx = 384 3.2 -> 4.1 TB/s, nx = 576 1.7 -> 2.5, nx = 1024 3.2 -> 3.4
// (profiles/r05/twopass_any_variants.txt).  MIL_XY_ONE_STAGE_MAX (build-time) restores the staged form above it.
#ifndef MIL_XY_ONE_STAGE_MAX
#define MIL_XY_ONE_STAGE_MAX 32
#endif
constexpr int xy_radix(int L) { return L <= MIL_XY_ONE_STAGE_MAX ? L : (L == 18 || L == 30) ? 6 : pick_radix(L); }
template <int M>
__device__ __forceinline__ int xy_pos_m(int b) {
    if constexpr (M == 1) return 0;
    else {
        constexpr int R = xy_radix(M);
        return (b % R) * (M / R) + xy_pos_m<M / R>(b / R);
    }
}

// the LDS stages of the sub-transforms of length L (L divides M = NH / 16) on the skewed tile of RB rows
template <int NH, int NT, int RB, int L, bool DIT, int SIGN>
__device__ __forceinline__ void xy_stage_rec(float2 *tile, const float2 *tw) {
    constexpr int M = NH / 16;
    if constexpr (L > 1) {
        constexpr int R = xy_radix(L);
        if constexpr (DIT) {
            if constexpr (L / R > 1) {
                xy_stage_rec<NH, NT, RB, L / R, true, SIGN>(tile, tw);
                lds_barrier();
            }
            fft_stage<NH, R, L, true, SIGN, RB, NT, RB + 1, M>(tile, tw);
        } else {
            fft_stage<NH, R, L, false, SIGN, RB, NT, RB + 1, M>(tile, tw);
            if constexpr (L / R > 1) {
                lds_barrier();
                xy_stage_rec<NH, NT, RB, L / R, false, SIGN>(tile, tw);
            }
        }
    }
}

// the LDS stage(s) of length M = NH / 16 of the x transform
template <int NH, int NT, int RB, bool DIT, int SIGN>
__device__ __forceinline__ void xy_stage_m(float2 *tile, const float2 *tw) {
    xy_stage_rec<NH, NT, RB, NH / 16, DIT, SIGN>(tile, tw);
}

// Occupancy.  The 64 rows x NH columns of a workgroup live in registers (128 per thread), which left room for two waves per
// SIMD; two waves spent 31 % of the time both waiting (loads, barriers, LDS round trips).  For NH <= 256 the kernel is built
// for THREE (<= 168 registers, three / six workgroups per CU):
//   * PARK = 8 of the 64 values wait in LDS (thread-private slots, 16 KB at NH = 256 -- all that three workgroups leave);
//   * the fused radix-16 stage runs as radix 2 (across lanes l and l + 32) x radix 8 (in the lane): 8 + 8 instead of
//     16 + 16 complex temporaries;
//   * addresses and twiddles of the trip loop are recomputed per trip (hoisted, they were what got spilled).
// Beyond NH = 256 (five to eight waves per workgroup, 40 - 74 KB of tile, radix-18 .. 32 LDS stages in registers) the kernel is
// built for two waves per SIMD without parking: one workgroup per CU.  (Three waves per SIMD -- two workgroups of five or six
// waves for NH = 288 .. 384 -- needed 76 - 100 bytes of scratch per lane and ran 10 - 20 % slower.)  NH that is not a whole
// number of waves (96, 160, 224 ...) runs with the last half wave idle in the column phases (threads = NH rounded up to 64)
// and 32 idle lane pairs in the fused stage.
#ifndef MIL_XY_OCC3_MAX
#define MIL_XY_OCC3_MAX 256
#endif
template <int NH>
struct XyCfg {
    static constexpr int threads = (NH + 63) / 64 * 64;
    static constexpr int occ = NH > MIL_XY_OCC3_MAX ? 2 : 3;
    static constexpr int park = NH > MIL_XY_OCC3_MAX ? 0 : 8;
    // rows per trip through the LDS.  Where a CU holds one workgroup anyway (NH > 256) two trips of 32 rows replace four of
    // 16 -- half the barrier-separated phases (five per trip) for twice the tile: 128 x 512 x nx ratio kernel 189 -> 182 us at
    // nx = 576, 242 -> 224 us at nx = 1024 (profiles/r05/twopass_any_variants.txt)
#ifndef MIL_XY_RB32_FROM
#define MIL_XY_RB32_FROM (MIL_XY_OCC3_MAX + 1)
#endif
    static constexpr int rb = NH >= MIL_XY_RB32_FROM ? 32 : 16;
};

// lanes l and l + 32 exchange (v_permlane32_swap):  a' = (a of the lower lanes | b of the LOWER lanes),
// b' = (a of the UPPER lanes | b of the upper lanes)
__device__ __forceinline__ void swap32(float2 &a, float2 &b) {
    const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = make_float2(__uint_as_float(x[0]), __uint_as_float(y[0]));
    b = make_float2(__uint_as_float(x[1]), __uint_as_float(y[1]));
}

// m / t: reciprocal, two Newton steps, v_div_fixup for the special cases -- the IEEE sequence without its range scaling
// (6 instead of 10 instructions; 32 divisions per thread and trip were 13 % of the ratio kernel's arithmetic).  Within one
// unit in the last place of the correctly rounded quotient for normal |t| below 2^126; outside that range (zero and
// denormal denominators, where the reciprocal overflows and the Newton step would turn inf into NaN, or huge ones, whose
// reciprocal flushes to zero) the lanes concerned take the IEEE division -- the quotient of the reference's unguarded
// div3Dkernel, +-inf at t = 0 included.  Two compares per division; the branch is never taken on real data (T is N times
// a positive estimate).
__device__ __forceinline__ float xy_div(float m, float t) {
    float r = __builtin_amdgcn_rcpf(t);
    r = __builtin_fmaf(__builtin_fmaf(-t, r, 1.0f), r, r);
    float q = m * r;
    q = __builtin_fmaf(__builtin_fmaf(-t, q, m), r, q);
    q = __builtin_amdgcn_div_fixupf(q, t, m);
    constexpr int kNormal = 0x008 | 0x100;  // v_cmp_class: -normal | +normal
    if (__builtin_expect(!(__builtin_amdgcn_classf(t, kNormal) && __builtin_fabsf(t) < 0x1p126f), 0)) q = m / t;
    return q;
}

// NY = 0: the y extent comes from the arguments (a.t = ny / 64): one instantiation per NH serves every ny.
template <int MODE, int NY, int NH, bool STREAM>
__global__ __launch_bounds__(XyCfg<NH>::threads, XyCfg<NH>::occ) void xy_kernel(XyArgs a) {
    const int T = NY ? NY / 64 : a.t;  // length of the y tail = distance between the rows of this workgroup
    const int ny = 64 * T;
    constexpr int M = NH / 16;  // length of the LDS stage of the x transform
    constexpr int kPitch = Geo<NH>::pitch;
    constexpr int RB = XyCfg<NH>::rb, WP = RB + 1, NX = 2 * NH;  // RB rows per trip through LDS: tile[position][row]
    constexpr int PASSES = RB / 8;                           // passes of the fused stage (8 rows each)
    constexpr int PARK = XyCfg<NH>::park, NV = 64 - PARK;
    constexpr int NT = XyCfg<NH>::threads;  // NH rounded up to whole waves
    constexpr bool RAGGED = NT != NH;       // the last 32 lanes hold no column, the last 32 lane pairs no butterfly
    // position p of the half-length sequence, row r: tile[p * WP + p / M + r] -- the skew keeps the fold / split, which
    // step through the positions M at a time, from hammering two banks (54 % of the LDS cycles were conflicts without it)
    // RAGGED: 16 more places take the fold's writes of the lanes without a column (no branches in the trip loop)
    __shared__ float2 tile[NH * WP + 16 + (RAGGED ? RB : 0)];
    __shared__ float2 tw[NH];  // w_NH^k = w_NX^(2k): the packed complex transform
    __shared__ float2 parked[PARK > 0 ? PARK * NT : 1];
    const int tid = threadIdx.x;
    const bool live = !RAGGED || tid < NH;  // this thread holds a spectrum column
    float2 *park = parked + tid;  // + i * NT: this thread's own slots
    const int n2 = blockIdx.x % T, z = blockIdx.x / T;
    const float2 *twy = c_tw_y + tw_y_offset(T);
    if (live) tw[tid] = a.tw_x[2 * tid];
    // This thread's spectrum column is X[kc] (even columns) or X[NH - kc] (odd), kc = tid / 2; columns 0 / 1 are the packed
    // (X[0], X[NH]) and X[NH/2].  The element e = a + 16 b of the half-length sequence it folds to / splits from sits at
    // position M * slot(a) + pos_M(b): block slot(a) = 2 (a mod 8) + a / 8 of the 16 blocks of length M (the order the
    // radix-2 x radix-8 fused stage wants its sub-transforms in).
    const int kc = tid >> 1;
    const bool odd = tid & 1;
    const int ez = tid == 1 ? NH / 2 : (odd ? NH - kc : kc);
    const int pz = M * (2 * (ez & 7) + ((ez >> 3) & 1)) + xy_pos_m<M>(ez >> 4);
    const int pzs = pz * WP + pz / M;                 // its place in the skewed tile
    const int pzw = live ? pzs : NH * WP + 16;        // where the fold writes it
    constexpr unsigned kDropped = 0x80000000u;        // a lane offset beyond every buffer: loads return 0, stores are dropped
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.S, a.spec_bytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(MODE == kXRatio ? a.M : a.E, a.real_bytes);
    const unsigned voff = live ? (unsigned)tid * 8u : kDropped;
    const unsigned soff0 = (unsigned)((z * ny + n2) * kPitch) * 8u;  // + m * rstep
    const unsigned rstep = (unsigned)T * kPitch * 8u;
    // Fused stage: lanes l and l + 32 (h = 0, 1) share the butterflies i = sbi of row srow (+ 8 in the second pass).  Lane h
    // works on the positions 2 M c + M h + sbi (c < 8: block c, half h), which the radix-8 stage turns into the real-space
    // elements n = si + 2 M r (x = 2 n, 2 n + 1): a wave's M / E access is 256 contiguous bytes of each of its rows.
    // Tile row r is volume row y = n2 + T r + 16 T grp.
    // Lane pair P = 32 wave + lane mod 32 takes butterfly P mod M of row P / M: 8 M = NH / 2 pairs.
    const int sh = (tid >> 5) & 1;
    const int pair_id = (tid & 31) + 32 * (tid >> 6);
    const bool pair_live = !RAGGED || pair_id < NH / 2;
    const int pair_eff = pair_live ? pair_id : 0;  // idle pairs shadow pair 0: its tile values, M / E = 0, nothing written
    const int sbi = pair_eff % M, srow = pair_eff / M;
    const int si = sbi + M * sh;
    float2 *sp = tile + si * WP + sh + srow;  // + c * SCS (block c) + 8 (second pass)
    constexpr int SCS = 2 * M * WP + 2;
    const unsigned svoff = pair_live ? (unsigned)(srow * T * NX + 2 * si) * 4u : kDropped;
    const unsigned rsoff0 = (unsigned)((z * ny + n2) * NX) * 4u;  // + grp * 16 T rows, + pass * 8 T rows, + 4 M r floats
    const unsigned rgrp = (unsigned)RB * T * NX * 4u, spass = 8u * T * NX * 4u;
    constexpr unsigned srj = 4u * M * 4u;
    float2 v[64];
    float2 pre[16];  // M / E of the pass at hand and of the next one
    MIL_STAMP(0);
    // Hermitian fold / split.  With A = own + oth, B = own - oth (oth = the pair lane's value) both lanes of a pair evaluate
    // the SAME expressions -- no per-lane selects (they were a fifth of the kernel's arithmetic):
    //   fold   Z' = (A.x - c1 A.y + c2 B.x,      B.y + c1 B.x + c2 A.y)        c1 = +/- w.x (even / odd lane), c2 = w.y
    //   split  X  = (A.x / 2 + h1 A.y + h2 B.x,  B.y / 2 - h1 B.x + h2 A.y)    h1 = c1 / 2, h2 = c2 / 2,  w = w_NX^kc
    const float2 wf = a.tw_x[kc];
    const float fc1 = odd ? -wf.x : wf.x, fc2 = wf.y, fh1 = 0.5f * fc1, fh2 = 0.5f * fc2;
#pragma unroll
    for (int m = 0; m < 64; ++m) v[m] = buf_load2<Aux<STREAM>::ld_spec>(rs, voff, soff0 + m * rstep);
    // ---- y inverse head: DIT stages L = NY / 8 (over j2, m = j2 + 8 j1; element i = n2 of its sub-transform) and L = NY
    // (over j1; element i = n2 + T j2)
#pragma unroll
    for (int j1 = 0; j1 < 8; ++j1) {
        float2 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[k + 8 * j1];
#pragma unroll
        for (int k = 1; k < 8; ++k) t[k] = tw_mul<+1>(t[k], twy[n2 * k * 8]);
        dft8<+1>(t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k + 8 * j1] = t[k];
    }
#pragma unroll
    for (int j2 = 0; j2 < 8; ++j2) {
        float2 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[j2 + 8 * k];
#pragma unroll
        for (int k = 1; k < 8; ++k) t[k] = tw_mul<+1>(t[k], twy[(n2 + T * j2) * k]);
        dft8<+1>(t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[j2 + 8 * k] = t[k];
    }
    MIL_STAMP(1);
    // The rows form a queue: the trip works on its first 16, the results join at the end; its last PARK places are in LDS.
#pragma unroll
    for (int i = 0; i < PARK; ++i) park[i * NT] = v[NV + i];
    lds_barrier();  // tw
    // ---- x^-1 . op . x, RB rows per trip (rows m = r + RB grp).  A real loop: the rows in flight always sit in
    // v[0 .. RB - 1] and the queue moves up by one group per trip.  (Unrolled, the compiler parks half of the tile in scratch.)
#pragma unroll 1
    for (int grp = 0; grp < 64 / RB; ++grp) {
        const unsigned rsoff = rsoff0 + grp * rgrp;
        // Hermitian fold, straight from the registers: the lane pair (2k, 2k + 1) holds (X[k], X[NH-k]); each lane forms
        // its own element of the packed half-length sequence Z' and drops it at that element's position.
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float2 own = v[r], oth = swap_pair(own);
            const float2 A = c_add(own, oth), B = c_sub(own, oth);
            tile[pzw + r] = make_float2(A.x - fc1 * A.y + fc2 * B.x, B.y + fc1 * B.x + fc2 * A.y);
        }
        if (tid < 2) {  // lanes 0 / 1 hold the packed (X[0], X[NH]) and X[NH/2]: their elements are written again
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float2 own = v[r];
                tile[pzs + r] = tid == 0 ? make_float2(own.x + own.y, own.x - own.y)  // Z'[0]
                                         : make_float2(2.f * own.x, -2.f * own.y);     // Z'[NH/2] = 2 conj(X[NH/2])
            }
        }
        // M / E of the first pass of the fused stage: in flight during the LDS stage (v[0..15] are free now)
#pragma unroll
        for (int r = 0; r < 8; ++r) pre[r] = buf_load2<Aux<STREAM>::ld_real>(rr, svoff, rsoff + r * srj);
        if (grp == 0) MIL_STAMP(8);
        lds_barrier();
        if (grp == 0) MIL_STAMP(9);
        xy_stage_m<NH, NT, RB, true, +1>(tile, tw);  // DIT stage(s) of length M
        lds_barrier();
        if (grp == 0) MIL_STAMP(10);
        // ---- last inverse stages, real-space operation, first forward stages: radix 2 (sub-transforms of length 2 M, split
        // over the lanes l and l + 32) and radix 8 (length NH) on 8 elements per lane, two passes of 8 rows
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            float2 *p = sp + 8 * ps;
            float2 t[8], mm[8];
            // twiddle addresses and the radix-2 twiddle are recomputed / re-read per pass: hoisted out of the trip loop they
            // cost ~20 registers, which the compiler then spills
            int si_l = si, wi_l = 8 * sbi;
            asm volatile("" : "+v"(si_l), "+v"(wi_l));
            const float2 wl = tw[wi_l];  // w_2M^sbi
#pragma unroll
            for (int r = 0; r < 8; ++r) mm[r] = pre[8 * (ps & 1) + r];
            if (ps + 1 < PASSES) {  // the next pass's M / E
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    pre[8 * ((ps + 1) & 1) + r] = buf_load2<Aux<STREAM>::ld_real>(rr, svoff, rsoff + (ps + 1) * spass + r * srj);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) t[c] = p[c * SCS];
#pragma unroll
            for (int c = 0; c < 4; ++c) swap32(t[c], t[c + 4]);  // lower lane: both halves of blocks 0..3, upper: of 4..7
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float2 u = t[c], w = tw_mul<+1>(t[c + 4], wl);
                t[c] = c_add(u, w);
                t[c + 4] = c_sub(u, w);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) swap32(t[c], t[c + 4]);  // lane h: half h of blocks 0..7
#pragma unroll
            for (int k = 1; k < 8; ++k) t[k] = tw_mul<+1>(t[k], tw[si_l * k]);
            dft8<+1>(t);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float2 m = mm[j];
                if (MODE == kXRatio) {  // div3Dkernel: T = A / T (unguarded)
                    t[j] = make_float2(xy_div(m.x, t[j].x), xy_div(m.y, t[j].y));
                } else {  // multi3Dkernel + maxvalue3Dgpukernel: E = max(E * T, eps)
                    float2 e = make_float2(m.x * t[j].x, m.y * t[j].y);
                    e.x = (e.x > a.eps) ? e.x : a.eps;
                    e.y = (e.y > a.eps) ? e.y : a.eps;
                    buf_store2<Aux<STREAM>::st_real>(rr, svoff, rsoff + ps * spass + j * srj, e);
                    t[j] = e;
                }
            }
            if constexpr (MODE != kXUpdateLast) {
                dft8<-1>(t);
                asm volatile("" : "+v"(si_l));
#pragma unroll
                for (int k = 1; k < 8; ++k) t[k] = tw_mul<-1>(t[k], tw[si_l * k]);
#pragma unroll
                for (int c = 0; c < 4; ++c) swap32(t[c], t[c + 4]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float2 u = t[c], w = t[c + 4];
                    t[c] = c_add(u, w);
                    t[c + 4] = tw_mul<-1>(c_sub(u, w), wl);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) swap32(t[c], t[c + 4]);
                if (pair_live) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) p[c * SCS] = t[c];
                }
            }
        }
        if (grp == 0) MIL_STAMP(11);
        if constexpr (MODE == kXUpdateLast) {
#pragma unroll
            for (int m = 0; m < NV - RB; ++m) v[m] = v[m + RB];
#pragma unroll
            for (int i = 0; i < PARK; ++i) v[NV - RB + i] = park[i * NT];
            lds_barrier();  // the next fold overwrites the tile
            continue;
        }
        lds_barrier();
        xy_stage_m<NH, NT, RB, false, -1>(tile, tw);  // DIF stage(s) of length M
        lds_barrier();
        if (grp == 0) MIL_STAMP(12);
        // The queue moves up by one trip; then the Hermitian split back into this thread's column: each lane picks up its
        // element Z[k] / Z[NH-k], the pair swaps.
#pragma unroll
        for (int m = 0; m < NV - RB; ++m) v[m] = v[m + RB];
#pragma unroll
        for (int i = 0; i < PARK; ++i) v[NV - RB + i] = park[i * NT];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float2 own = tile[pzs + r], oth = swap_pair(own);
            const float2 A = c_add(own, oth), B = c_sub(own, oth);
            float2 o = make_float2(0.5f * A.x + fh1 * A.y + fh2 * B.x, 0.5f * B.y - fh1 * B.x + fh2 * A.y);
            if (tid < 2) o = tid == 0 ? make_float2(own.x + own.y, own.x - own.y)  // (X[0], X[NH]), both real
                                      : make_float2(own.x, -own.y);                // X[NH/2] = conj(Z[NH/2])
            if (r < RB - PARK) v[64 - RB + r] = o;
            else park[(r - (RB - PARK)) * NT] = o;
        }
        if (grp == 0) MIL_STAMP(13);
        lds_barrier();  // the next fold overwrites the tile
        if (grp == 0) MIL_STAMP(2);
    }
    MIL_STAMP(3);
    if constexpr (MODE == kXUpdateLast) return;
#pragma unroll
    for (int i = 0; i < PARK; ++i) v[NV + i] = park[i * NT];
    // ---- y forward head: DIF stages L = NY (over j1) and L = NY / 8 (over j2)
#pragma unroll
    for (int j2 = 0; j2 < 8; ++j2) {
        float2 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[j2 + 8 * k];
        dft8<-1>(t);
#pragma unroll
        for (int k = 1; k < 8; ++k) t[k] = tw_mul<-1>(t[k], twy[(n2 + T * j2) * k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[j2 + 8 * k] = t[k];
    }
#pragma unroll
    for (int j1 = 0; j1 < 8; ++j1) {
        float2 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[k + 8 * j1];
        dft8<-1>(t);
#pragma unroll
        for (int k = 1; k < 8; ++k) t[k] = tw_mul<-1>(t[k], twy[n2 * k * 8]);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k + 8 * j1] = t[k];
    }
    MIL_STAMP(4);
#pragma unroll
    for (int m = 0; m < 64; ++m) buf_store2<Aux<STREAM>::st_spec>(rs, voff, soff0 + m * rstep, v[m]);
    MIL_STAMP(5);
#ifdef MIL_FX_STAMPS
    if (a.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        MIL_STAMP(6);
    }
#endif
}

