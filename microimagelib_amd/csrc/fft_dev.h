// Device-side building blocks of the LDS-tiled transforms (csrc/fft.hip, csrc/fft2p.hip): complex helpers, the
// radix-2/4/8 and odd-prime butterflies, and one in-place mixed-radix stage over an LDS tile.
// Included inside namespace mil::fx { namespace { ... } } by the .hip files.
#pragma once

// Workgroup barrier for LDS traffic only.  __syncthreads() also drains the wave's outstanding GLOBAL loads
// (s_waitcnt vmcnt(0)), which turns every prefetch issued before a barrier into an exposed HBM round trip; the tiles
// here are exchanged through LDS alone, so waiting for the LDS / scalar counter is enough.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kW = 16;    // tile width in complex columns (128 B)
constexpr int kWP = 17;   // LDS row stride in float2
constexpr int kNT = 256;  // threads per workgroup
constexpr int kBatch = 8; // float4 loads in flight per lane while staging


// Written on the 2-vectors themselves: v_pk_add/mul/fma_f32 without relying on the SLP vectoriser (csrc/fft2p.hip is built
// without it), 35 % fewer vector instructions in a radix-8 butterfly than component by component.
__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return a + b; }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return a - b; }
__device__ __forceinline__ float2 c_mul(float2 a, float2 b) {
    return make_float2(a.x, a.x) * b + make_float2(a.y, a.y) * make_float2(-b.y, b.x);
}
__device__ __forceinline__ float2 c_mulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x, a.x) * make_float2(b.x, -b.y) + make_float2(a.y, a.y) * make_float2(b.y, b.x);
}
// tables hold w = exp(-2*pi*i*k/n); SIGN = -1 uses w, SIGN = +1 uses conj(w)
template <int SIGN>
__device__ __forceinline__ float2 tw_mul(float2 a, float2 w) {
    return SIGN < 0 ? c_mul(a, w) : c_mulc(a, w);
}
// multiply by SIGN * i
template <int SIGN>
__device__ __forceinline__ float2 rot90(float2 a) {
    return SIGN < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}
// a + SIGN * i * b and (SIGN * i * a) * s as ONE packed instruction each: b's halves swapped by op_sel, the sign in a constant
// vector ((-1, 1) or (1, -1) from an SGPR pair, a multiplication by one: exact, same bits as a + rot90(b)).  Written as
// c_add(a, rot90(b)) the compiler first BUILDS the rotated vector (v_xor_b32 + v_mov_b32: it folds a swap and a sign into
// the operands of v_pk_mul / v_pk_fma, not into v_pk_add's): 10 of the 36 vector instructions of a radix-8 butterfly,
// 620 of the 3 300 of the xy kernel (round 6).
template <int SIGN>
__device__ __forceinline__ float2 add_i(float2 a, float2 b) {
    return make_float2(b.y, b.x) * (SIGN > 0 ? make_float2(-1.f, 1.f) : make_float2(1.f, -1.f)) + a;
}
template <int SIGN>
__device__ __forceinline__ float2 rot_mul(float2 a, float s) {
    return make_float2(a.y, a.x) * (SIGN > 0 ? make_float2(-s, s) : make_float2(s, -s));
}

template <int SIGN>
__device__ __forceinline__ void dft2(float2 *v) {
    float2 a = v[0], b = v[1];
    v[0] = c_add(a, b);
    v[1] = c_sub(a, b);
}

template <int SIGN>
__device__ __forceinline__ void dft4(float2 &v0, float2 &v1, float2 &v2, float2 &v3) {
    float2 a0 = c_add(v0, v2), a1 = c_sub(v0, v2);
    float2 a2 = c_add(v1, v3), a3 = c_sub(v1, v3);
    v0 = c_add(a0, a2);
    v2 = c_sub(a0, a2);
    v1 = add_i<SIGN>(a1, a3);
    v3 = add_i<-SIGN>(a1, a3);
}

template <int SIGN>
__device__ __forceinline__ void dft8(float2 *v) {
    constexpr float h = 0.70710678118654752440f;
    dft4<SIGN>(v[0], v[2], v[4], v[6]);  // even part, outputs E[0..3] in v0,v2,v4,v6
    dft4<SIGN>(v[1], v[3], v[5], v[7]);  // odd part,  outputs O[0..3] in v1,v3,v5,v7
    float2 o1 = v[3], o2 = v[5], o3 = v[7];
    // w8^1, w8^2, w8^3 with w8 = exp(SIGN*i*pi/4):  (1 + SIGN i) / sqrt2,  SIGN i,  (-1 + SIGN i) / sqrt2
    o1 = add_i<SIGN>(o1, o1);   // times h below
    o3 = add_i<-SIGN>(o3, o3);  // (SIGN i o3 - o3) = -this
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
    v[0] = c_add(e0, o0);
    v[4] = c_sub(e0, o0);
    v[1] = o1 * make_float2(h, h) + e1;  // each product has ONE use: contracted into the sum (v_pk_fma_f32)
    v[5] = o1 * make_float2(-h, -h) + e1;
    v[2] = add_i<SIGN>(e2, o2);
    v[6] = add_i<-SIGN>(e2, o2);
    v[3] = o3 * make_float2(-h, -h) + e3;
    v[7] = o3 * make_float2(h, h) + e3;
}

// a * w16^M, w16 = exp(SIGN * 2 pi i / 16), M known at compile time
template <int SIGN, int M>
__device__ __forceinline__ float2 mul_w16(float2 a) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    if constexpr (M == 0) return a;
    else if constexpr (M == 4) return rot90<SIGN>(a);
    else if constexpr (M == 2) return add_i<SIGN>(a, a) * make_float2(h, h);
    else if constexpr (M == 6) return add_i<-SIGN>(a, a) * make_float2(-h, -h);
    else {  // a * (cos t + SIGN i sin t) = a cos t + rot90(a) sin t
        static_assert(M == 1 || M == 3 || M == 9, "mul_w16: exponent not used by dft16");
        constexpr float c = (M == 1) ? c1 : (M == 3) ? s1 : -c1;
        constexpr float s = (M == 1) ? s1 : (M == 3) ? c1 : -s1;
        return a * make_float2(c, c) + rot_mul<SIGN>(a, s);
    }
}

// 16-point transform, natural order in and out: n = 4 n1 + n2, k = k1 + 4 k2 (4 x 4 with the w16 twiddles in between)
template <int SIGN>
__device__ __forceinline__ void dft16(float2 *v) {
    dft4<SIGN>(v[0], v[4], v[8], v[12]);  // over n1 for n2 = 0: y[n2][k1] lands in v[n2 + 4 k1]
    dft4<SIGN>(v[1], v[5], v[9], v[13]);
    dft4<SIGN>(v[2], v[6], v[10], v[14]);
    dft4<SIGN>(v[3], v[7], v[11], v[15]);
    v[5] = mul_w16<SIGN, 1>(v[5]);
    v[9] = mul_w16<SIGN, 2>(v[9]);
    v[13] = mul_w16<SIGN, 3>(v[13]);
    v[6] = mul_w16<SIGN, 2>(v[6]);
    v[10] = mul_w16<SIGN, 4>(v[10]);
    v[14] = mul_w16<SIGN, 6>(v[14]);
    v[7] = mul_w16<SIGN, 3>(v[7]);
    v[11] = mul_w16<SIGN, 6>(v[11]);
    v[15] = mul_w16<SIGN, 9>(v[15]);
    dft4<SIGN>(v[0], v[1], v[2], v[3]);  // over n2 for k1 = 0: X[k1 + 4 k2] lands in v[k2 + 4 k1]
    dft4<SIGN>(v[4], v[5], v[6], v[7]);
    dft4<SIGN>(v[8], v[9], v[10], v[11]);
    dft4<SIGN>(v[12], v[13], v[14], v[15]);
    float2 t;  // transpose to natural order
#define MIL_SWAP16(a, b) t = v[a]; v[a] = v[b]; v[b] = t;
    MIL_SWAP16(1, 4) MIL_SWAP16(2, 8) MIL_SWAP16(3, 12) MIL_SWAP16(6, 9) MIL_SWAP16(7, 13) MIL_SWAP16(11, 14)
#undef MIL_SWAP16
}

// cos / sin(2*pi*k/R) for the odd radices
template <int R>
struct OddTable;
template <>
struct OddTable<3> {
    static constexpr float c[3] = {1.0f, -0.49999999999999978f, -0.50000000000000044f};
    static constexpr float s[3] = {0.0f, 0.86602540378443871f, -0.86602540378443837f};
};
template <>
struct OddTable<5> {
    static constexpr float c[5] = {1.0f, 0.30901699437494745f, -0.80901699437494734f, -0.80901699437494756f, 0.30901699437494723f};
    static constexpr float s[5] = {0.0f, 0.95105651629515353f, 0.58778525229247325f, -0.58778525229247303f, -0.95105651629515364f};
};
template <>
struct OddTable<7> {
    static constexpr float c[7] = {1.0f, 0.62348980185873359f, -0.22252093395631434f, -0.90096886790241903f, -0.90096886790241915f, -0.22252093395631459f, 0.62348980185873337f};
    static constexpr float s[7] = {0.0f, 0.7818314824680298f, 0.97492791218182362f, 0.43388373911755823f, -0.43388373911755801f, -0.97492791218182362f, -0.78183148246802991f};
};
template <>
struct OddTable<11> {
    static constexpr float c[11] = {1.0f, 0.84125353283118121f, 0.41541501300188644f, -0.142314838273285f, -0.65486073394528499f, -0.95949297361449737f, -0.95949297361449748f, -0.65486073394528521f, -0.14231483827328523f, 0.41541501300188605f, 0.84125353283118121f};
    static constexpr float s[11] = {0.0f, 0.54064081745559756f, 0.90963199535451833f, 0.9898214418809328f, 0.75574957435425827f, 0.28173255684142967f, -0.28173255684142939f, -0.75574957435425816f, -0.98982144188093268f, -0.90963199535451855f, -0.54064081745559744f};
};
template <>
struct OddTable<13> {
    static constexpr float c[13] = {1.0f, 0.88545602565320991f, 0.56806474673115592f, 0.12053668025532301f, -0.35460488704253545f, -0.74851074817110119f, -0.97094181742605201f, -0.97094181742605212f, -0.7485107481711013f, -0.3546048870425359f, 0.1205366802553232f, 0.56806474673115481f, 0.88545602565321002f};
    static constexpr float s[13] = {0.0f, 0.46472317204376851f, 0.82298386589365635f, 0.99270887409805397f, 0.93501624268541483f, 0.66312265824079519f, 0.23931566428755768f, -0.23931566428755743f, -0.66312265824079497f, -0.93501624268541472f, -0.99270887409805397f, -0.82298386589365702f, -0.4647231720437684f};
};

// R-point DFT for odd prime R by definition, folded over conjugate pairs:
//   out[k], out[R-k] = (v0 + sum_j (v_j + v_(R-j)) cos(jk)) +/- SIGN*i * sum_j (v_j - v_(R-j)) sin(jk)
template <int R, int SIGN>
__device__ __forceinline__ void dft_odd(float2 *v) {
    constexpr int H = (R - 1) / 2;
    float2 a[H], b[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
        a[j] = c_add(v[j + 1], v[R - 1 - j]);
        b[j] = c_sub(v[j + 1], v[R - 1 - j]);
    }
    const float2 v0 = v[0];
    float2 sum = v0;
#pragma unroll
    for (int j = 0; j < H; ++j) sum = c_add(sum, a[j]);
    v[0] = sum;
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        float2 p = v0, q = make_float2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int e = ((j + 1) * k) % R;
            p.x += a[j].x * OddTable<R>::c[e];
            p.y += a[j].y * OddTable<R>::c[e];
            q.x += b[j].x * OddTable<R>::s[e];
            q.y += b[j].y * OddTable<R>::s[e];
        }
        v[k] = add_i<SIGN>(p, q);  // p +/- SIGN * i * q
        v[R - k] = add_i<-SIGN>(p, q);
    }
}


// cos / sin(2*pi*k/R) for the composite radices (6 .. 15: two factors; 18 .. 32: the second factor composite again)
template <int R>
struct CompTable;
template <>
struct CompTable<6> {
    static constexpr float c[6] = {1.0f, 0.5000000000000001f, -0.4999999999999998f, -1.0f, -0.5000000000000004f, 0.5000000000000001f};
    static constexpr float s[6] = {0.0f, 0.8660254037844386f, 0.8660254037844387f, 0.0f, -0.8660254037844384f, -0.8660254037844386f};
};
template <>
struct CompTable<9> {
    static constexpr float c[9] = {1.0f, 0.766044443118978f, 0.17364817766693041f, -0.4999999999999998f, -0.9396926207859083f, -0.9396926207859084f, -0.5000000000000004f, 0.17364817766692997f, 0.7660444431189778f};
    static constexpr float s[9] = {0.0f, 0.6427876096865393f, 0.984807753012208f, 0.8660254037844387f, 0.3420201433256689f, -0.34202014332566866f, -0.8660254037844384f, -0.9848077530122081f, -0.6427876096865396f};
};
template <>
struct CompTable<10> {
    static constexpr float c[10] = {1.0f, 0.8090169943749475f, 0.30901699437494745f, -0.30901699437494734f, -0.8090169943749473f, -1.0f, -0.8090169943749476f, -0.30901699437494756f, 0.30901699437494723f, 0.8090169943749473f};
    static constexpr float s[10] = {0.0f, 0.5877852522924731f, 0.9510565162951535f, 0.9510565162951536f, 0.5877852522924732f, 0.0f, -0.587785252292473f, -0.9510565162951535f, -0.9510565162951536f, -0.5877852522924734f};
};
template <>
struct CompTable<12> {
    static constexpr float c[12] = {1.0f, 0.8660254037844387f, 0.5000000000000001f, 0.0f, -0.4999999999999998f, -0.8660254037844387f, -1.0f, -0.8660254037844388f, -0.5000000000000004f, 0.0f, 0.5000000000000001f, 0.8660254037844384f};
    static constexpr float s[12] = {0.0f, 0.49999999999999994f, 0.8660254037844386f, 1.0f, 0.8660254037844387f, 0.49999999999999994f, 0.0f, -0.4999999999999997f, -0.8660254037844384f, -1.0f, -0.8660254037844386f, -0.5000000000000004f};
};
template <>
struct CompTable<14> {
    static constexpr float c[14] = {1.0f, 0.9009688679024191f, 0.6234898018587336f, 0.22252093395631445f, -0.22252093395631434f, -0.6234898018587335f, -0.900968867902419f, -1.0f, -0.9009688679024191f, -0.6234898018587337f, -0.2225209339563146f, 0.22252093395631334f, 0.6234898018587334f, 0.9009688679024194f};
    static constexpr float s[14] = {0.0f, 0.4338837391175581f, 0.7818314824680298f, 0.9749279121818236f, 0.9749279121818236f, 0.7818314824680299f, 0.43388373911755823f, 0.0f, -0.433883739117558f, -0.7818314824680297f, -0.9749279121818236f, -0.9749279121818238f, -0.7818314824680299f, -0.4338837391175575f};
};
template <>
struct CompTable<15> {
    static constexpr float c[15] = {1.0f, 0.9135454576426009f, 0.6691306063588582f, 0.30901699437494745f, -0.10452846326765333f, -0.4999999999999998f, -0.8090169943749473f, -0.9781476007338057f, -0.9781476007338057f, -0.8090169943749476f, -0.5000000000000004f, -0.10452846326765423f, 0.30901699437494723f, 0.6691306063588585f, 0.913545457642601f};
    static constexpr float s[15] = {0.0f, 0.40673664307580015f, 0.7431448254773941f, 0.9510565162951535f, 0.9945218953682734f, 0.8660254037844387f, 0.5877852522924732f, 0.20791169081775931f, -0.20791169081775907f, -0.587785252292473f, -0.8660254037844384f, -0.9945218953682733f, -0.9510565162951536f, -0.743144825477394f, -0.40673664307580015f};
};
template <>
struct CompTable<18> {
    static constexpr float c[18] = {1.0f, 0.9396926207859084f, 0.766044443118978f, 0.5000000000000001f, 0.17364817766693041f, -0.1736481776669303f, -0.4999999999999998f, -0.7660444431189779f, -0.9396926207859083f, -1.0f, -0.9396926207859084f, -0.7660444431189783f, -0.5000000000000004f, -0.17364817766693033f, 0.17364817766692997f, 0.49999999999999933f, 0.7660444431189778f, 0.9396926207859084f};
    static constexpr float s[18] = {0.0f, 0.3420201433256687f, 0.6427876096865393f, 0.8660254037844386f, 0.984807753012208f, 0.984807753012208f, 0.8660254037844387f, 0.6427876096865395f, 0.3420201433256689f, 0.0f, -0.34202014332566866f, -0.6427876096865389f, -0.8660254037844384f, -0.984807753012208f, -0.9848077530122081f, -0.866025403784439f, -0.6427876096865396f, -0.3420201433256686f};
};
template <>
struct CompTable<20> {
    static constexpr float c[20] = {1.0f, 0.9510565162951535f, 0.8090169943749475f, 0.5877852522924731f, 0.30901699437494745f, 0.0f, -0.30901699437494734f, -0.587785252292473f, -0.8090169943749473f, -0.9510565162951535f, -1.0f, -0.9510565162951538f, -0.8090169943749476f, -0.5877852522924732f, -0.30901699437494756f, 0.0f, 0.30901699437494723f, 0.5877852522924729f, 0.8090169943749473f, 0.9510565162951535f};
    static constexpr float s[20] = {0.0f, 0.3090169943749474f, 0.5877852522924731f, 0.8090169943749475f, 0.9510565162951535f, 1.0f, 0.9510565162951536f, 0.8090169943749475f, 0.5877852522924732f, 0.3090169943749475f, 0.0f, -0.3090169943749469f, -0.587785252292473f, -0.8090169943749473f, -0.9510565162951535f, -1.0f, -0.9510565162951536f, -0.8090169943749476f, -0.5877852522924734f, -0.3090169943749476f};
};
template <>
struct CompTable<22> {
    static constexpr float c[22] = {1.0f, 0.9594929736144974f, 0.8412535328311812f, 0.6548607339452851f, 0.41541501300188644f, 0.14231483827328512f, -0.142314838273285f, -0.4154150130018863f, -0.654860733945285f, -0.8412535328311811f, -0.9594929736144974f, -1.0f, -0.9594929736144975f, -0.8412535328311812f, -0.6548607339452852f, -0.41541501300188716f, -0.14231483827328523f, 0.14231483827328487f, 0.41541501300188605f, 0.6548607339452845f, 0.8412535328311812f, 0.9594929736144974f};
    static constexpr float s[22] = {0.0f, 0.28173255684142967f, 0.5406408174555976f, 0.7557495743542583f, 0.9096319953545183f, 0.9898214418809327f, 0.9898214418809328f, 0.9096319953545184f, 0.7557495743542583f, 0.5406408174555978f, 0.28173255684142967f, 0.0f, -0.2817325568414294f, -0.5406408174555976f, -0.7557495743542582f, -0.909631995354518f, -0.9898214418809327f, -0.9898214418809328f, -0.9096319953545186f, -0.7557495743542587f, -0.5406408174555974f, -0.2817325568414298f};
};
template <>
struct CompTable<24> {
    static constexpr float c[24] = {1.0f, 0.9659258262890683f, 0.8660254037844387f, 0.7071067811865476f, 0.5000000000000001f, 0.25881904510252074f, 0.0f, -0.25881904510252063f, -0.4999999999999998f, -0.7071067811865475f, -0.8660254037844387f, -0.9659258262890682f, -1.0f, -0.9659258262890683f, -0.8660254037844388f, -0.7071067811865479f, -0.5000000000000004f, -0.25881904510252063f, 0.0f, 0.2588190451025203f, 0.5000000000000001f, 0.7071067811865474f, 0.8660254037844384f, 0.9659258262890681f};
    static constexpr float s[24] = {0.0f, 0.25881904510252074f, 0.49999999999999994f, 0.7071067811865475f, 0.8660254037844386f, 0.9659258262890683f, 1.0f, 0.9659258262890683f, 0.8660254037844387f, 0.7071067811865476f, 0.49999999999999994f, 0.258819045102521f, 0.0f, -0.2588190451025208f, -0.4999999999999997f, -0.7071067811865471f, -0.8660254037844384f, -0.9659258262890683f, -1.0f, -0.9659258262890684f, -0.8660254037844386f, -0.7071067811865477f, -0.5000000000000004f, -0.25881904510252157f};
};
template <>
struct CompTable<26> {
    static constexpr float c[26] = {1.0f, 0.970941817426052f, 0.8854560256532099f, 0.7485107481711011f, 0.5680647467311559f, 0.35460488704253557f, 0.120536680255323f, -0.12053668025532288f, -0.35460488704253545f, -0.5680647467311557f, -0.7485107481711012f, -0.8854560256532096f, -0.970941817426052f, -1.0f, -0.9709418174260521f, -0.8854560256532101f, -0.7485107481711013f, -0.5680647467311559f, -0.3546048870425359f, -0.12053668025532357f, 0.1205366802553232f, 0.35460488704253557f, 0.5680647467311548f, 0.7485107481711007f, 0.88545602565321f, 0.970941817426052f};
    static constexpr float s[26] = {0.0f, 0.23931566428755774f, 0.4647231720437685f, 0.6631226582407952f, 0.8229838658936564f, 0.9350162426854148f, 0.992708874098054f, 0.992708874098054f, 0.9350162426854148f, 0.8229838658936565f, 0.6631226582407952f, 0.46472317204376906f, 0.23931566428755768f, 0.0f, -0.23931566428755743f, -0.46472317204376806f, -0.663122658240795f, -0.8229838658936564f, -0.9350162426854147f, -0.992708874098054f, -0.992708874098054f, -0.9350162426854148f, -0.822983865893657f, -0.6631226582407955f, -0.4647231720437684f, -0.2393156642875578f};
};
template <>
struct CompTable<28> {
    static constexpr float c[28] = {1.0f, 0.9749279121818236f, 0.9009688679024191f, 0.7818314824680298f, 0.6234898018587336f, 0.4338837391175582f, 0.22252093395631445f, 0.0f, -0.22252093395631434f, -0.43388373911755806f, -0.6234898018587335f, -0.7818314824680295f, -0.900968867902419f, -0.9749279121818237f, -1.0f, -0.9749279121818237f, -0.9009688679024191f, -0.7818314824680296f, -0.6234898018587337f, -0.4338837391175583f, -0.2225209339563146f, 0.0f, 0.22252093395631334f, 0.43388373911755795f, 0.6234898018587334f, 0.7818314824680297f, 0.9009688679024194f, 0.9749279121818235f};
    static constexpr float s[28] = {0.0f, 0.2225209339563144f, 0.4338837391175581f, 0.6234898018587335f, 0.7818314824680298f, 0.9009688679024191f, 0.9749279121818236f, 1.0f, 0.9749279121818236f, 0.9009688679024191f, 0.7818314824680299f, 0.6234898018587339f, 0.43388373911755823f, 0.2225209339563141f, 0.0f, -0.22252093395631384f, -0.433883739117558f, -0.6234898018587338f, -0.7818314824680297f, -0.900968867902419f, -0.9749279121818236f, -1.0f, -0.9749279121818238f, -0.9009688679024193f, -0.7818314824680299f, -0.6234898018587337f, -0.4338837391175575f, -0.22252093395631464f};
};
template <>
struct CompTable<30> {
    static constexpr float c[30] = {1.0f, 0.9781476007338057f, 0.9135454576426009f, 0.8090169943749475f, 0.6691306063588582f, 0.5000000000000001f, 0.30901699437494745f, 0.10452846326765346f, -0.10452846326765333f, -0.30901699437494734f, -0.4999999999999998f, -0.6691306063588579f, -0.8090169943749473f, -0.913545457642601f, -0.9781476007338057f, -1.0f, -0.9781476007338057f, -0.9135454576426009f, -0.8090169943749476f, -0.6691306063588585f, -0.5000000000000004f, -0.30901699437494756f, -0.10452846326765423f, 0.10452846326765299f, 0.30901699437494723f, 0.5000000000000001f, 0.6691306063588585f, 0.8090169943749473f, 0.913545457642601f, 0.9781476007338057f};
    static constexpr float s[30] = {0.0f, 0.20791169081775931f, 0.40673664307580015f, 0.5877852522924731f, 0.7431448254773941f, 0.8660254037844386f, 0.9510565162951535f, 0.9945218953682733f, 0.9945218953682734f, 0.9510565162951536f, 0.8660254037844387f, 0.7431448254773945f, 0.5877852522924732f, 0.40673664307580004f, 0.20791169081775931f, 0.0f, -0.20791169081775907f, -0.4067366430758002f, -0.587785252292473f, -0.743144825477394f, -0.8660254037844384f, -0.9510565162951535f, -0.9945218953682733f, -0.9945218953682734f, -0.9510565162951536f, -0.8660254037844386f, -0.743144825477394f, -0.5877852522924734f, -0.40673664307580015f, -0.20791169081775898f};
};
template <>
struct CompTable<32> {
    static constexpr float c[32] = {1.0f, 0.9807852804032304f, 0.9238795325112867f, 0.8314696123025452f, 0.7071067811865476f, 0.5555702330196023f, 0.38268343236508984f, 0.19509032201612833f, 0.0f, -0.1950903220161282f, -0.3826834323650897f, -0.555570233019602f, -0.7071067811865475f, -0.8314696123025453f, -0.9238795325112867f, -0.9807852804032304f, -1.0f, -0.9807852804032304f, -0.9238795325112868f, -0.8314696123025455f, -0.7071067811865477f, -0.5555702330196022f, -0.38268343236509034f, -0.19509032201612866f, 0.0f, 0.1950903220161283f, 0.38268343236509f, 0.5555702330196018f, 0.7071067811865474f, 0.8314696123025452f, 0.9238795325112865f, 0.9807852804032303f};
    static constexpr float s[32] = {0.0f, 0.19509032201612825f, 0.3826834323650898f, 0.5555702330196022f, 0.7071067811865475f, 0.8314696123025452f, 0.9238795325112867f, 0.9807852804032304f, 1.0f, 0.9807852804032304f, 0.9238795325112867f, 0.8314696123025455f, 0.7071067811865476f, 0.5555702330196022f, 0.3826834323650899f, 0.1950903220161286f, 0.0f, -0.19509032201612836f, -0.38268343236508967f, -0.555570233019602f, -0.7071067811865475f, -0.8314696123025452f, -0.9238795325112865f, -0.9807852804032303f, -1.0f, -0.9807852804032304f, -0.9238795325112866f, -0.8314696123025455f, -0.7071067811865477f, -0.5555702330196022f, -0.3826834323650904f, -0.19509032201612872f};
};

// a * exp(SIGN * 2 pi i m / R), m known at compile time
template <int R, int SIGN, int M>
__device__ __forceinline__ float2 mul_wr(float2 a) {
    constexpr float c = CompTable<R>::c[M % R], s = CompTable<R>::s[M % R];
    if constexpr (c == 1.0f) return a;
    else if constexpr (c == -1.0f) return make_float2(-a.x, -a.y);
    else if constexpr (c == 0.0f && s == 1.0f) return rot90<SIGN>(a);
    else if constexpr (c == 0.0f && s == -1.0f) return rot90<-SIGN>(a);
    else return a * make_float2(c, c) + rot_mul<SIGN>(a, s);
}

template <int R, int SIGN>
__device__ __forceinline__ void dft_small(float2 *v);

// R = R1 x R2 points (6 = 2 x 3, 9 = 3 x 3, 10 = 2 x 5, 12 = 4 x 3, 14 = 2 x 7, 15 = 3 x 5; 18 = 2 x 9, 20 = 4 x 5, 22 = 2 x 11,
// 24 = 4 x 6, 26 = 2 x 13, 28 = 4 x 7, 30 = 2 x 15, 32 = 4 x 8), natural order in and out:
// n = R2 n1 + n2, k = k1 + R1 k2; R2 transforms of length R1, the twiddles w_R^(n2 k1), R1 transforms of length R2.
template <int R, int R1, int SIGN, int N2, int K1>
__device__ __forceinline__ void comp_twiddle(float2 *y) {
    if constexpr (N2 < R / R1) {
        if constexpr (K1 < R1) {
            y[N2 * R1 + K1] = mul_wr<R, SIGN, N2 * K1>(y[N2 * R1 + K1]);
            comp_twiddle<R, R1, SIGN, N2, K1 + 1>(y);
        } else {
            comp_twiddle<R, R1, SIGN, N2 + 1, 1>(y);
        }
    }
}
template <int R, int SIGN>
__device__ __forceinline__ void dft_comp(float2 *v) {
    constexpr int R1 = (R % 4 == 0) ? 4 : (R % 2 == 0) ? 2 : 3, R2 = R / R1;
    float2 y[R];  // y[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
        float2 t[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[R2 * n1 + n2];
        dft_small<R1, SIGN>(t);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) y[n2 * R1 + k1] = t[k1];
    }
    comp_twiddle<R, R1, SIGN, 1, 1>(y);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        float2 t[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = y[n2 * R1 + k1];
        dft_small<R2, SIGN>(t);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
    }
}
constexpr bool is_comp_radix(int R) {
    return R == 6 || R == 9 || R == 10 || R == 12 || R == 14 || R == 15 || (R >= 18 && R <= 32 && R % 2 == 0);
}

template <int R, int SIGN>
__device__ __forceinline__ void dft_small(float2 *v) {
    if constexpr (is_comp_radix(R)) dft_comp<R, SIGN>(v);
    else if constexpr (R == 16) dft16<SIGN>(v);
    else if constexpr (R == 8) dft8<SIGN>(v);
    else if constexpr (R == 4) dft4<SIGN>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 2) dft2<SIGN>(v);
    else dft_odd<R, SIGN>(v);
}

// One radix-R pass over sub-transforms of length L of an N x 16 tile.
//   DIF: y_k = w_L^(i*k) * DFT_R(x)[k]      DIT: y = DFT_R( w_L^(i*k) * x_k )
// Lane mapping: column fastest (16 lanes), then butterfly index.
// SKEW > 0: position p lives at tile[p * WP + p / SKEW + column] (csrc/fft2p.hip: keeps accesses that step through the
// positions SKEW at a time off a single pair of banks); only for stages whose sub-transforms do not straddle a skew period
// (L <= SKEW, SKEW a multiple of L).
template <int N, int R, int L, bool DIT, int SIGN, int W = kW, int NT = kNT, int WP = W + 1, int SKEW = 0>
__device__ __forceinline__ void fft_stage(float2 *__restrict__ tile, const float2 *__restrict__ tw) {
    static_assert(SKEW == 0 || (L <= SKEW && SKEW % L == 0), "fft_stage: a sub-transform must stay inside one skew period");
    constexpr int M = L / R;            // butterflies per sub-transform
    constexpr int NB = (N / R) * W;     // butterflies in the tile
    constexpr int TWS = N / L;          // table step: w_L = w_N^(N/L)
#pragma unroll
    for (int it = 0; it < (NB + NT - 1) / NT; ++it) {
        const int idx = it * NT + (int)threadIdx.x;
        if ((NB % NT) != 0 && idx >= NB) break;
        const int c = idx % W;
        const int b = idx / W;
        const int blk = b / M, i = b % M;
        float2 *p = tile + (blk * L + i) * WP + c + (SKEW ? (blk * L) / (SKEW ? SKEW : 1) : 0);
        float2 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = p[j * M * WP];
        if (DIT && M > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = tw_mul<SIGN>(v[k], tw[i * k * TWS]);
        }
        dft_small<R, SIGN>(v);
        if (!DIT && M > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = tw_mul<SIGN>(v[k], tw[i * k * TWS]);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) p[j * M * WP] = v[j];
    }
}

// Radix schedule of a length-L sub-transform: 8s first, then 4 / 2, then the odd primes of the
// 64*k sizes snapTransformSize produces (192 = 8*8*3, 320 = 8*8*5, ... 960 = 8*8*3*5).
constexpr int pick_radix(int L) {
    return (L % 8 == 0) ? 8 : (L % 4 == 0) ? 4 : (L % 2 == 0) ? 2 : (L % 3 == 0) ? 3 : (L % 5 == 0) ? 5
         : (L % 7 == 0) ? 7 : (L % 11 == 0) ? 11 : (L % 13 == 0) ? 13 : L;
}
template <int L>
struct Radix {
    static constexpr int value = pick_radix(L);
};

// Barrier contract: the tile must be complete (barrier done) on entry; on return every
// stage has been followed by a barrier.  LSTOP > 1 leaves the sub-transforms of length LSTOP undone (DIF) /
// expects them done already (DIT): the two-pass path splits a transform between two kernels there.
template <int N, int L, int SIGN, int W = kW, int NT = kNT, int WP = W + 1, int LSTOP = 1>
__device__ __forceinline__ void fft_dif(float2 *tile, const float2 *tw) {
    if constexpr (L > LSTOP) {
        constexpr int R = Radix<L>::value;
        fft_stage<N, R, L, false, SIGN, W, NT, WP>(tile, tw);
        lds_barrier();
        fft_dif<N, L / R, SIGN, W, NT, WP, LSTOP>(tile, tw);
    }
}
template <int N, int L, int SIGN, int W = kW, int NT = kNT, int LSTOP = 1>
__device__ __forceinline__ void fft_dit(float2 *tile, const float2 *tw) {
    if constexpr (L > LSTOP) {
        constexpr int R = Radix<L>::value;
        fft_dit<N, L / R, SIGN, W, NT, LSTOP>(tile, tw);
        fft_stage<N, R, L, true, SIGN, W, NT>(tile, tw);
        lds_barrier();
    }
}

// ---- N x 16 column tiles: staging between HBM (128-byte rows) and LDS tile[N][17]
template <int N, int NT>
__device__ __forceinline__ void tile_load(float2 *__restrict__ tile, const float2 *__restrict__ base, long long stride) {
    constexpr int TOTAL = N * 8;  // float4 pieces
#pragma unroll 1
    for (int it0 = 0; it0 < TOTAL; it0 += NT * kBatch) {
        float4 v[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int idx = it0 + j * NT + (int)threadIdx.x;
            if (idx < TOTAL) v[j] = *reinterpret_cast<const float4 *>(base + (long long)(idx >> 3) * stride + 2 * (idx & 7));
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int idx = it0 + j * NT + (int)threadIdx.x;
            if (idx < TOTAL) {
                float2 *q = tile + (idx >> 3) * kWP + 2 * (idx & 7);
                q[0] = make_float2(v[j].x, v[j].y);
                q[1] = make_float2(v[j].z, v[j].w);
            }
        }
    }
}

template <int N, int NT>
__device__ __forceinline__ void tile_store(const float2 *__restrict__ tile, float2 *__restrict__ base, long long stride) {
    constexpr int TOTAL = N * 8;
#pragma unroll 4
    for (int idx = (int)threadIdx.x; idx < TOTAL; idx += NT) {
        const float2 *q = tile + (idx >> 3) * kWP + 2 * (idx & 7);
        const float2 a = q[0], b = q[1];
        *reinterpret_cast<float4 *>(base + (long long)(idx >> 3) * stride + 2 * (idx & 7)) = make_float4(a.x, a.y, b.x, b.y);
    }
}

// tile[p][c] *= otf[p][c], otf tile contiguous [N][16]
template <int N, int NT>
__device__ __forceinline__ void tile_mul(float2 *__restrict__ tile, const float2 *__restrict__ otf) {
    constexpr int TOTAL = N * 8;
#pragma unroll 1
    for (int it0 = 0; it0 < TOTAL; it0 += NT * kBatch) {
        float4 v[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int idx = it0 + j * NT + (int)threadIdx.x;
            if (idx < TOTAL) v[j] = *reinterpret_cast<const float4 *>(otf + 2 * idx);
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int idx = it0 + j * NT + (int)threadIdx.x;
            if (idx < TOTAL) {
                float2 *q = tile + (idx >> 3) * kWP + 2 * (idx & 7);
                q[0] = c_mul(q[0], make_float2(v[j].x, v[j].y));
                q[1] = c_mul(q[1], make_float2(v[j].z, v[j].w));
            }
        }
    }
}

