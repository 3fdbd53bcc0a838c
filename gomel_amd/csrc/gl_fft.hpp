// gl_fft.hpp -- register/LDS real-FFT core for the Griffin-Lim iteration on gfx950 (CDNA4, wave64).
//
// Replaces, per frame, go-dsp's fft.FFTReal and fft.IFFT as the reference calls them (mel/mel.go:95,116) and the
// per-bin Abs/Phase/Rect + conjugate fill between them (mel/mel.go:98-108).
//
// Design (not a translation of the reference's complex128 radix-2 loops):
//  * An N-point real frame is packed as C = N/2 complex points z[n] = x[2n] + i x[2n+1]; one "stream" of
//    T = C/16 lanes owns a frame, 16 complex points per lane, so x / window / overlap-add live in registers
//    at pair index n = t + T*j (lane t, register j).
//  * Three in-register passes R1 x R2 x R3 (16x16x8, 16x8x8, 8x8x8, 8x4x8, 4x4x8 for N = 4096, 2048, 1024, 512, 256) with two LDS
//    exchanges (layouts A and B below, padded so the b64 reads of every pass are bank-conflict free).
//  * The last pass is always radix-8 and each lane runs the two mirrored butterflies b and B3-b, so the bins
//    k and C-k that the real-FFT split needs meet in ONE lane's registers: the split, the magnitude clamp /
//    phase keep (Y = M X/|X|), and the merge for the inverse all happen in registers with no LDS traffic.
//  * The inverse runs the same three passes transposed (R3, R2, R1) with conjugated twiddles, through the same
//    LDS layouts, and lands back in the n = t + T*j register layout ready for the synthesis window and OLA.
//  * Twiddles come from small tables (TW1: W_C^{t1 q1}, TW2: W_{C1}^{t2 q2}, WN: W_N^k) held in LDS or L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace glfft {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));
template <class V> struct ScalarOf;
template <> struct ScalarOf<v2f> { typedef float type; };
template <> struct ScalarOf<v2d> { typedef double type; };

#define GL_DEV __device__ __forceinline__

// ---------------------------------------------------------------------------------------------- geometry
template <int N_> struct Geo;
template <int N_, int R1_, int R2_> struct GeoBase {
    static constexpr int N = N_;          // real frame length (Resolut)
    static constexpr int C = N_ / 2;      // complex FFT length
    static constexpr int P = 16;          // complex points per lane
    static constexpr int T = C / P;       // lanes per frame stream
    static constexpr int R1 = R1_, R2 = R2_, R3 = 8;
    static_assert(R1 * R2 * R3 == C, "radix split");
    static constexpr int C1 = C / R1;     // length of the sub-FFTs after pass 1
    static constexpr int B3 = C / R3;     // number of pass-3 butterflies; Z index k = b + B3*r
    static constexpr int NB1 = P / R1;    // pass-1 butterflies per lane
    static constexpr int NB2 = P / R2;    // pass-2 butterflies per lane
    static constexpr int SA = C1 + 32 / R1;  // row stride of exchange layout A (b64 slots), forward: rows written whole, read by (q1, t2)
    static constexpr int SAI = C1 + 16 / R1; // the same layout in the inverse, where the (q1, t2) access is the WRITE: ds_write_b64 is
                                             // served in groups of 16 lanes over 32 banks (reads: 32 lanes over 64), so the 16 lanes of a
                                             // group -- R1 values of q1 x 16/R1 of t2 -- must land on 16 different slots mod 16
    static constexpr int SB = B3 + R1;       // row stride of exchange layout B
    static constexpr int SLOTS_A = R1 * SA, SLOTS_B = R3 * SB;
    static constexpr int LDS_SLOTS = SLOTS_A > SLOTS_B ? SLOTS_A : SLOTS_B;  // v2f slots per stream
    static constexpr int TW1_N = R1 * C1;  // == C entries, index q1*C1 + t1
    static constexpr int TW2_N = R2 * R3;  // index q2*8 + t2
};
template <> struct Geo<4096> : GeoBase<4096, 16, 16> {};
template <> struct Geo<2048> : GeoBase<2048, 16, 8> {};
template <> struct Geo<1024> : GeoBase<1024, 8, 8> {};
template <> struct Geo<512> : GeoBase<512, 8, 4> {};
template <> struct Geo<256> : GeoBase<256, 4, 4> {};

// Device-side view of the twiddle tables (global memory or LDS; the code is the same).
template <class V> struct Twiddles {
    const V* tw1;  // [R1][C1]  W_C^{t1 q1}
    const V* tw2;  // [R2][8]   W_{C1}^{t2 q2}
    const V* wn;   // [C]       W_N^k
    const V* w1r = nullptr;  // optional: THIS lane's pass-1 twiddles (index u + NB1*q) for q <= w1r_n, same idea as w2r; the rest from tw1
    int w1r_n = 0;
    const V* w2r = nullptr;  // optional: THIS lane's 16 pass-2 twiddles (index u*R2 + q) already in registers -- they depend on
                             // the lane only, so a kernel that walks many frames loads them once (k_gl_fused) instead of 30 LDS
                             // reads per frame (forward + inverse)
};

// ---------------------------------------------------------------------------------------------- complex helpers
// Written in explicit 2-lane vector form so that hipcc emits ONE packed op per line and folds the lane swaps /
// broadcasts / negations into VOP3P op_sel / neg modifiers (a complex multiply = v_pk_mul + v_pk_fma).  The scalar-form
// spelling gets SLP-packed into 3 packed ops + a v_mov with half the lanes thrown away.
#define GL_XX(v) __builtin_shufflevector(v, v, 0, 0)
#define GL_YY(v) __builtin_shufflevector(v, v, 1, 1)
#define GL_YX(v) __builtin_shufflevector(v, v, 1, 0)
template <class V> GL_DEV V vfma(V a, V b, V c) { return __builtin_elementwise_fma(a, b, c); }
template <class V> GL_DEV V rot_mi(V a) { V s = GL_YX(a); return V{s.x, -s.y}; }  // a * (-i) = ( a.y, -a.x)
template <class V> GL_DEV V rot_pi(V a) { V s = GL_YX(a); return V{-s.x, s.y}; }  // a * (+i) = (-a.y,  a.x)
template <class V> GL_DEV V cmul(V a, V b) { return vfma(rot_pi(a), GL_YY(b), a * GL_XX(b)); }   // a * b
template <class V> GL_DEV V cmulc(V a, V b) { return vfma(rot_mi(a), GL_YY(b), a * GL_XX(b)); }  // a * conj(b)
template <bool INV, class V> GL_DEV V mul_mi(V a) { return INV ? rot_pi(a) : rot_mi(a); }  // * (-i) fwd, (+i) inv
// a * (c - i s) forward, a * (c + i s) inverse
template <bool INV, class V> GL_DEV V twc(V a, double c_, double s_) {
    typedef typename ScalarOf<V>::type S;
    const V cc = V{(S)c_, (S)c_}, ss = V{(S)s_, (S)s_};
    return vfma(INV ? rot_pi(a) : rot_mi(a), ss, a * cc);
}

constexpr double kS2 = 0.70710678118654752440;  // sqrt(1/2)
constexpr double kC1 = 0.92387953251128675613;  // cos(pi/8)
constexpr double kS1 = 0.38268343236508977173;  // sin(pi/8)

template <bool INV, class V> GL_DEV void dft4_c(V& a0, V& a1, V& a2, V& a3) {
    V s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = mul_mi<INV>(a1 - a3);
    a0 = s02 + s13; a1 = d02 + d13; a2 = s02 - s13; a3 = d02 - d13;
}

// ---------------------------------------------------------------------------------------------- fp32 packed-math asm forms
// hipcc folds lane broadcasts into VOP3P op_sel but not lane SWAPS or single-lane negations: every multiply by +-i and
// every complex multiply costs an extra v_xor + v_mov.  These blocks spell the packed instructions with their modifiers.
// Hazard rule honoured by construction (CDNA3/4 ISA, manually inserted wait states): a VALU that uses OPSEL needs one
// wait state before a consumer of its result -- inside a block no instruction consumes the result of the one right
// before it (or an s_nop sits between), and hipcc pads the block's outputs itself after the statement.  Every block
// opens with its non-OPSEL reads or an s_nop, because hipcc does not check a block's inputs against the instruction
// it places in front of it.  -DGOMEL_NO_PKASM builds the plain C++ forms (bit-identical results; used as a cross-check).
#ifndef GOMEL_NO_PKASM
#define GL_PKASM 1
#else
#define GL_PKASM 0
#endif
#ifndef GL_LEAD_NOP
#define GL_LEAD_NOP "s_nop 0\n\t"   // every block opens with one wait state (measured again in round 2)
#endif

// (a0, a1) <- (a0 * w0, a1 * w1), or * conj(w) when CONJ.  w may be VGPR pairs or (uniform constants) SGPR pairs.
template <bool CONJ> GL_DEV void cmul2_asm(v2f& a0, v2f w0, v2f& a1, v2f w1) {
    v2f t0, t1;
    if (!CONJ)
        asm(GL_LEAD_NOP
            "v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %3, %5 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %2, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
            "v_pk_fma_f32 %3, %3, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
            : "=&v"(t0), "=&v"(t1), "+v"(a0), "+v"(a1) : "v"(w0), "v"(w1));
    else
        asm(GL_LEAD_NOP
            "v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %3, %5 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %2, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 %3, %3, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
            : "=&v"(t0), "=&v"(t1), "+v"(a0), "+v"(a1) : "v"(w0), "v"(w1));
}
template <bool CONJ> GL_DEV void cmul1_asm(v2f& a0, v2f w0) {
    v2f t0;
    if (!CONJ)
        asm(GL_LEAD_NOP
            "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
            "s_nop 0\n\t"
            "v_pk_fma_f32 %1, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
            : "=&v"(t0), "+v"(a0) : "v"(w0));
    else
        asm(GL_LEAD_NOP
            "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
            "s_nop 0\n\t"
            "v_pk_fma_f32 %1, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
            : "=&v"(t0), "+v"(a0) : "v"(w0));
}
// same with the multiplier a uniform constant held in an SGPR pair
template <bool CONJ> GL_DEV void cmul2_sconst_asm(v2f& a0, v2f k0, v2f& a1, v2f k1) {
    v2f t0, t1;
    if (!CONJ)
        asm(GL_LEAD_NOP
            "v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %3, %5 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %2, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
            "v_pk_fma_f32 %3, %3, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
            : "=&v"(t0), "=&v"(t1), "+v"(a0), "+v"(a1) : "s"(k0), "s"(k1));
    else
        asm(GL_LEAD_NOP
            "v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %3, %5 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %2, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 %3, %3, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
            : "=&v"(t0), "=&v"(t1), "+v"(a0), "+v"(a1) : "s"(k0), "s"(k1));
}
template <bool CONJ> GL_DEV void cmul1_sconst_asm(v2f& a0, v2f k0) {
    v2f t0;
    if (!CONJ)
        asm(GL_LEAD_NOP
            "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
            "s_nop 0\n\t"
            "v_pk_fma_f32 %1, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
            : "=&v"(t0), "+v"(a0) : "s"(k0));
    else
        asm(GL_LEAD_NOP
            "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
            "s_nop 0\n\t"
            "v_pk_fma_f32 %1, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
            : "=&v"(t0), "+v"(a0) : "s"(k0));
}
// radix-4 butterfly, 8 packed adds: the multiply of (a1 - a3) by -+i is folded into the last two adds' op_sel/neg.
template <bool INV> GL_DEV void dft4_asm(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
    v2f tA, tB;
    if (!INV)
        asm("v_pk_add_f32 %0, %2, %4\n\t"                                   // tA = a0 + a2
            "v_pk_add_f32 %2, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"         // a0 = a0 - a2
            "v_pk_add_f32 %1, %3, %5\n\t"                                   // tB = a1 + a3
            "v_pk_add_f32 %3, %3, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"         // a1 = a1 - a3
            "v_pk_add_f32 %4, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"         // a2 = tA - tB            (X2)
            "v_pk_add_f32 %0, %0, %1\n\t"                                   // tA = tA + tB            (X0)
            "v_pk_add_f32 %5, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"   // a3 = d02 + i d   (X3)
            "v_pk_add_f32 %3, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"       // a1 = d02 - i d   (X1)
            : "=&v"(tA), "=&v"(tB), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    else
        asm("v_pk_add_f32 %0, %2, %4\n\t"
            "v_pk_add_f32 %2, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %1, %3, %5\n\t"
            "v_pk_add_f32 %3, %3, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %4, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %0, %0, %1\n\t"
            "v_pk_add_f32 %5, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"   // a3 = d02 - i d   (X3, inverse)
            "v_pk_add_f32 %3, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"       // a1 = d02 + i d   (X1, inverse)
            : "=&v"(tA), "=&v"(tB), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    a0 = tA;
}

template <bool INV, class V> GL_DEV void dft4(V& a0, V& a1, V& a2, V& a3) {
    if constexpr (GL_PKASM && __is_same(V, v2f)) dft4_asm<INV>(a0, a1, a2, a3); else dft4_c<INV>(a0, a1, a2, a3);
}
// (a0, a1) *= W^{+-1}-style constant twiddles (c - i s) forward / (c + i s) inverse
template <bool INV, class V> GL_DEV void twc2(V& a0, double c0, double s0, V& a1, double c1, double s1) {
    if constexpr (GL_PKASM && __is_same(V, v2f)) {
        // a*(c - i s) = a*conj(c + i s): CONJ form forward, plain form inverse, constant (c, s) in an SGPR pair
        cmul2_sconst_asm<!INV>(a0, v2f{(float)c0, (float)s0}, a1, v2f{(float)c1, (float)s1});
    } else { a0 = twc<INV>(a0, c0, s0); a1 = twc<INV>(a1, c1, s1); }
}
template <bool INV, class V> GL_DEV void twc1(V& a0, double c0, double s0) {
    if constexpr (GL_PKASM && __is_same(V, v2f)) cmul1_sconst_asm<!INV>(a0, v2f{(float)c0, (float)s0});
    else a0 = twc<INV>(a0, c0, s0);
}
// v[i] *= w[i] (or conj) for i in [1, n), in interleaved pairs
template <bool CONJ, int STRIDE, int CNT, class V> GL_DEV void apply_tw(V* v, const V* w) {
    if constexpr (GL_PKASM && __is_same(V, v2f)) {
#pragma unroll
        for (int q = 1; q + 1 < CNT; q += 2) cmul2_asm<CONJ>(v[q * STRIDE], w[q * STRIDE], v[(q + 1) * STRIDE], w[(q + 1) * STRIDE]);
        if constexpr ((CNT - 1) % 2 == 1) cmul1_asm<CONJ>(v[(CNT - 1) * STRIDE], w[(CNT - 1) * STRIDE]);
    } else {
#pragma unroll
        for (int q = 1; q < CNT; q++) v[q * STRIDE] = CONJ ? cmulc(v[q * STRIDE], w[q * STRIDE]) : cmul(v[q * STRIDE], w[q * STRIDE]);
    }
}

// fp64 forms of the butterflies' constant twiddles whose real and imaginary parts are both +-sqrt(1/2) (round 5): a (c -+ i s)
// with |c| = |s| = k is k times a sum / difference of a's components, and the factor k rides on the multiply-adds of the radix-4 stage that
// follows -- 2 adds + 4 multiply-adds where "complex multiply, then add and subtract" takes 4 + 4 (the fp32 kernels keep their packed asm forms).
// (-DGL_D_FMATW=0 builds the plain "complex multiply, then radix-4" form for A/B: -48 fp64 instructions per frame and wave with it on)
#ifndef GL_D_FMATW
#define GL_D_FMATW 1
#endif
template <bool INV, class V> GL_DEV V q_pp(V a) { return INV ? V{a.x - a.y, a.x + a.y} : V{a.x + a.y, a.y - a.x}; }      // a (1 -+ i):  a twc(k, k) / k
template <bool INV, class V> GL_DEV V q_mp(V a) { return INV ? V{-a.x - a.y, a.x - a.y} : V{a.y - a.x, -a.x - a.y}; }    // a (-1 -+ i): a twc(-k, k) / k
// dft4 with a1 = k A, a3 = k B given as A, B
template <bool INV, class V> GL_DEV void dft4_k13(V& a0, V& a1, V& a2, V& a3, double k_) {
    typedef typename ScalarOf<V>::type S;
    const V kk = V{(S)k_, (S)k_}, s02 = a0 + a2, d02 = a0 - a2, s = a1 + a3, d = mul_mi<INV>(a1 - a3);
    a0 = vfma(s, kk, s02); a2 = vfma(s, -kk, s02); a1 = vfma(d, kk, d02); a3 = vfma(d, -kk, d02);
}
// dft4 with a2 = k Q given as Q (a1, a3 already twiddled)
template <bool INV, class V> GL_DEV void dft4_k2(V& a0, V& a1, V& a2, V& a3, double k_) {
    typedef typename ScalarOf<V>::type S;
    const V kk = V{(S)k_, (S)k_}, s02 = vfma(a2, kk, a0), d02 = vfma(a2, -kk, a0), s13 = a1 + a3, d13 = mul_mi<INV>(a1 - a3);
    a0 = s02 + s13; a1 = d02 + d13; a2 = s02 - s13; a3 = d02 - d13;
}

// In-place DFT of R registers at stride S in v[]; output q lands in v[q*S] (natural order).
template <bool INV, int S, class V> GL_DEV void dft8(V* v) {
    V u0[4], u1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { u0[j] = v[j * S] + v[(j + 4) * S]; u1[j] = v[j * S] - v[(j + 4) * S]; }
    dft4<INV>(u0[0], u0[1], u0[2], u0[3]);
    if constexpr (GL_D_FMATW && __is_same(V, v2d)) {
        u1[1] = q_pp<INV>(u1[1]); u1[3] = q_mp<INV>(u1[3]);
        u1[2] = mul_mi<INV>(u1[2]);
        dft4_k13<INV>(u1[0], u1[1], u1[2], u1[3], kS2);
    } else {
    twc2<INV>(u1[1], kS2, kS2, u1[3], -kS2, kS2);
    u1[2] = mul_mi<INV>(u1[2]);
    dft4<INV>(u1[0], u1[1], u1[2], u1[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { v[(2 * r) * S] = u0[r]; v[(2 * r + 1) * S] = u1[r]; }
}

template <bool INV, int S, class V> GL_DEV void dft16(V* v) {
    V u[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u[j][0] = v[j * S]; u[j][1] = v[(j + 4) * S]; u[j][2] = v[(j + 8) * S]; u[j][3] = v[(j + 12) * S];
        dft4<INV>(u[j][0], u[j][1], u[j][2], u[j][3]);
    }
    // u[j][p] *= W16^{j p}
    if constexpr (GL_D_FMATW && __is_same(V, v2d)) {
        u[1][1] = twc<INV>(u[1][1], kC1, kS1); u[1][3] = twc<INV>(u[1][3], kS1, kC1);
        u[3][1] = twc<INV>(u[3][1], kS1, kC1); u[3][3] = twc<INV>(u[3][3], -kC1, -kS1);
        u[2][1] = q_pp<INV>(u[2][1]); u[2][3] = q_mp<INV>(u[2][3]);           // columns 1, 3: the k-type twiddle sits on a2
        u[1][2] = q_pp<INV>(u[1][2]); u[3][2] = q_mp<INV>(u[3][2]);           // column 2: on a1 and a3
        u[2][2] = mul_mi<INV>(u[2][2]);
        dft4<INV>(u[0][0], u[1][0], u[2][0], u[3][0]);
        dft4_k2<INV>(u[0][1], u[1][1], u[2][1], u[3][1], kS2);
        dft4_k13<INV>(u[0][2], u[1][2], u[2][2], u[3][2], kS2);
        dft4_k2<INV>(u[0][3], u[1][3], u[2][3], u[3][3], kS2);
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int r = 0; r < 4; r++) v[(p + 4 * r) * S] = u[r][p];
    } else {
    twc2<INV>(u[1][1], kC1, kS1, u[1][2], kS2, kS2);
    twc2<INV>(u[1][3], kS1, kC1, u[2][1], kS2, kS2);
    u[2][2] = mul_mi<INV>(u[2][2]);
    twc2<INV>(u[2][3], -kS2, kS2, u[3][1], kS1, kC1);
    twc2<INV>(u[3][2], -kS2, kS2, u[3][3], -kC1, -kS1);
#pragma unroll
    for (int p = 0; p < 4; p++) {
        dft4<INV>(u[0][p], u[1][p], u[2][p], u[3][p]);
#pragma unroll
        for (int r = 0; r < 4; r++) v[(p + 4 * r) * S] = u[r][p];
    }
    }
}

template <bool INV, int R, int S, class V> GL_DEV void dftR(V* v) {
    if constexpr (R == 16) dft16<INV, S>(v);
    else if constexpr (R == 8) dft8<INV, S>(v);
    else { static_assert(R == 4, "radix"); dft4<INV>(v[0], v[S], v[2 * S], v[3 * S]); }
}

// ---------------------------------------------------------------------------------------------- diagnostic stamps
// Where a frame's cycles go: a diagnostic build (-DGOMEL_STAMPS, scripts/stamp_probe.py) passes a Stamps object whose mark(k)
// reads s_memtime (and drains the wave's LDS / scalar queue, so each phase is charged with its own LDS round trips) and adds the
// cycles since the previous mark to slot k.  The product build passes NoStamps: every mark is empty and no stamp executes.
struct NoStamps { GL_DEV void mark(int) {} };
struct Stamps {
    unsigned long long last = 0, acc[20] = {};
    GL_DEV void start() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    GL_DEV void mark(int k) {
        unsigned long long t_;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        acc[k] += t_ - last; last = t_;
    }
};

// ---------------------------------------------------------------------------------------------- exchange synchronisation
// The passes below synchronise a stream's lanes at two kinds of points: behind a WRITE burst (the partner's data must be there before
// anything is read: a full barrier, `sync()`), and behind a READ burst (the partner must have read before the area is written again).
// The second kind only has to hold by the NEXT WRITE, a whole butterfly later.  A plain callable (`s_barrier`, or nothing for a
// one-wave stream) takes both as `sync()`.  A split-phase object -- `arrive()` / `wait()`, e.g. per-wave event counters in LDS -- signals
// "my reads are issued" at once (`sync_reads_done`) and waits for the partner only in front of the next write burst
// (`sync_before_write`), by which time the partner has long arrived: that wait never stalls.
template <class S> struct IsSplitSync {
    template <class U> static auto test(U* u) -> decltype(u->arrive(), char());
    template <class U> static long test(...);
    static constexpr bool value = sizeof(test<S>(nullptr)) == sizeof(char);
};
template <class S> GL_DEV void sync_reads_done(S& s) { if constexpr (IsSplitSync<S>::value) s.arrive(); else s(); }
template <class S> GL_DEV void sync_before_write(S& s) { if constexpr (IsSplitSync<S>::value) s.wait(); }


// Write bursts interleaved with arithmetic (fp64 kernel with split-phase synchronisation only, one wave per SIMD): sixteen
// ds_write_b128 back to back fill the LDS FIFOs (SQ_LDS_DATA_FIFO_FULL + CMD_FIFO_FULL: 7 % of the wave cycles) while the vector unit
// idles; issued one by one between the twiddle multiplies / the last butterfly stage they drain as fast as they come.  The wait for the
// partner's reads moves in front of that arithmetic (where the compiler had hoisted it anyway).  GL_D_INTERLEAVE=0 builds the burst form.
#ifndef GL_D_INTERLEAVE
#define GL_D_INTERLEAVE 1
#endif
#define GL_SGB_VALU 0x002
#define GL_SGB_DSW 0x200
template <int NV, int NW> GL_DEV void sgb_valu_then_write() {   // NW times: NV vector instructions, one LDS write
#pragma unroll
    for (int i = 0; i < NW; i++) { __builtin_amdgcn_sched_group_barrier(GL_SGB_VALU, NV, 0); __builtin_amdgcn_sched_group_barrier(GL_SGB_DSW, 1, 0); }
}

// ---------------------------------------------------------------------------------------------- forward
// In: v[j] = z[t + T*j].  Out: v[0..7] = Z[bA + B3*r], v[8..15] = Z[bB + B3*r] with bA = t, bB = (t ? B3 - t : B3/2).
// `lds` is this stream's exchange area (G::LDS_SLOTS slots).  SYNC() is a barrier covering the stream's lanes.
// PAIRED (bit 0): the caller runs fft_inverse on the same exchange area right after the spectral step.  Pass 3 reads, and the inverse's
// first pass writes, the SAME slots from the SAME lane (r * SB + bA, r * SB + bB: every slot of layout B belongs to exactly one lane
// there), so the "reads done -> may write" synchronisation between the two is not needed: one barrier (or one event pair) less per frame.
// PAIRED bit 1: ... and the next thing to touch the area after that fft_inverse is this function again (the fused kernels' frame loop).
// Then layout A uses ONE row stride in both directions (SAI: the inverse's transposed write stays conflict-free, the forward's transposed
// read takes a two-way conflict on one lane pair per group), the inverse's last reads and the next forward's first writes are again
// same-lane-same-slot, and the synchronisation between THEM goes too.
// PAIRED bit 2 (round 6; geometries with one butterfly per lane and pass: N 4096): the pass-2 butterfly's results go back to the SLOTS ITS OPERANDS CAME
// FROM -- element (q1, t2, q2) at row q1, column t2 + 8 q2 of layout A -- instead of into layout B.  Every slot then has exactly one owner lane in
// every pass (the lane that reads it is the lane that overwrites it), so the whole frame needs no "the partner has read, I may write" synchronisation:
// what is left is one "the partner has written, I may read" per exchange (4 polls per frame instead of 6, no event published behind a read burst).
// Pass 3 finds the eight operands of butterfly b = q1 + R1 q2 in eight CONSECUTIVE slots of row q1 (conflict-free across a 16-lane group for the odd
// row stride SAI); layout B is not used.  Arithmetic and its order are unchanged: bit-identical results.
// (A/B knobs, profiles/r06/ab_inplace.txt: the odd stride SAI = 129 is the best of 129 / 130 / either with a column swizzle -- 1.2566 / 1.2718 / 1.2701 /
// 1.3127 ms; with ONE stride for the transposed reads AND writes one of the two conflicts two-way, and the writes' conflicts cost more.  The counters show it:
// SQ_LDS_BANK_CONFLICT 39 M per launch against 13 M for the two-layout form -- the in-place exchange wins by what it no longer waits for, not by its LDS pattern)
#ifndef GL_D_INPL_SA
#define GL_D_INPL_SA 0      // row stride of the in-place layout in slots (0: G::SAI)
#endif
#ifndef GL_D_INPL_XOR
#define GL_D_INPL_XOR 0     // 1: column ^ ((row >> 2) & 1)
#endif
// slot of (row q, column c) of the in-place layout
template <class G> GL_DEV int inpl_slot(int q, int c) {
    constexpr int S = GL_D_INPL_SA ? GL_D_INPL_SA : G::SAI;
    return q * S + (GL_D_INPL_XOR ? (c ^ ((q >> 2) & 1)) : c);
}
template <class G, int PAIRED = 0, class V, class SyncT, class St> GL_DEV void fft_forward(V (&v)[16], V* lds, int t, const Twiddles<V>& tw, SyncT sync, St& st) {
    constexpr bool INPL = (PAIRED & 4) && G::NB1 == 1 && G::NB2 == 1;
    static_assert(!(PAIRED & 4) || INPL, "the in-place exchange is built for one butterfly per lane and pass");
    constexpr int SA = ((PAIRED & 2) || INPL) ? G::SAI : G::SA;
    // ---- pass 1: NB1 radix-R1 butterflies; butterfly u holds registers j = u + NB1*m, t1 = t + T*u
    // The pass's twiddles are read into registers BEFORE any exchange store: table and exchange area live in the same
    // LDS array, so a twiddle read placed after a store could not be hoisted above it (one LDS round trip per element).
    V w1[16];
#pragma unroll
    for (int u = 0; u < G::NB1; u++)
#pragma unroll
        for (int q = 1; q < G::R1; q++) w1[u + G::NB1 * q] = (tw.w1r && q <= tw.w1r_n) ? tw.w1r[u + G::NB1 * q] : tw.tw1[q * G::C1 + t + G::T * u];
#pragma unroll
    for (int u = 0; u < G::NB1; u++) dftR<false, G::R1, G::NB1>(&v[u]);
    __builtin_amdgcn_sched_barrier(0);   // keep the twiddle reads above the butterfly: they land while it computes
    constexpr bool ILV = GL_D_INTERLEAVE && __is_same(V, v2d) && IsSplitSync<SyncT>::value;
    if constexpr (ILV) {
        st.mark(0);
        if constexpr (!(PAIRED & 2)) sync_before_write(sync);
#pragma unroll
        for (int u = 0; u < G::NB1; u++) {
            const int t1 = t + G::T * u;
#pragma unroll
            for (int q = 0; q < G::R1; q++) {
                if (q) v[u + G::NB1 * q] = cmul(v[u + G::NB1 * q], w1[u + G::NB1 * q]);
                if constexpr (INPL) lds[inpl_slot<G>(q, t1)] = v[u + G::NB1 * q]; else lds[q * SA + t1] = v[u + G::NB1 * q];
            }
        }
        sgb_valu_then_write<4, 16>();
    } else {
#pragma unroll
    for (int u = 0; u < G::NB1; u++) apply_tw<false, G::NB1, G::R1>(&v[u], &w1[u]);
    st.mark(0);
    if constexpr (!(PAIRED & 2)) sync_before_write(sync);
#pragma unroll
    for (int u = 0; u < G::NB1; u++) {
        const int t1 = t + G::T * u;
#pragma unroll
        for (int q = 0; q < G::R1; q++) { if constexpr (INPL) lds[inpl_slot<G>(q, t1)] = v[u + G::NB1 * q]; else lds[q * SA + t1] = v[u + G::NB1 * q]; }
    }
    }
    sync();
    st.mark(1);
    // ---- pass 2: NB2 radix-R2 butterflies beta = t + T*u2 -> (q1 = beta % R1, t2 = beta / R1)
#pragma unroll
    for (int u = 0; u < G::NB2; u++) {
        const int beta = t + G::T * u, q1 = beta % G::R1, t2 = beta / G::R1;
#pragma unroll
        for (int m = 0; m < G::R2; m++) { if constexpr (INPL) v[u * G::R2 + m] = lds[inpl_slot<G>(q1, t2 + 8 * m)]; else v[u * G::R2 + m] = lds[q1 * SA + t2 + 8 * m]; }
    }
    if constexpr (!INPL) sync_reads_done(sync);
    st.mark(2);
    V w2[16];
#pragma unroll
    for (int u = 0; u < G::NB2; u++) {
        const int t2 = (t + G::T * u) / G::R1;
#pragma unroll
        for (int q = 1; q < G::R2; q++) w2[u * G::R2 + q] = tw.w2r ? tw.w2r[u * G::R2 + q] : tw.tw2[q * 8 + t2];
    }
#pragma unroll
    for (int u = 0; u < G::NB2; u++) dftR<false, G::R2, 1>(&v[u * G::R2]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ILV) {
        st.mark(3);
        if constexpr (!INPL) sync_before_write(sync);
#pragma unroll
        for (int u = 0; u < G::NB2; u++) {
            const int beta = t + G::T * u, q1 = beta % G::R1, t2 = beta / G::R1;
#pragma unroll
            for (int q = 0; q < G::R2; q++) {
                if (q) v[u * G::R2 + q] = cmul(v[u * G::R2 + q], w2[u * G::R2 + q]);
                if constexpr (INPL) lds[inpl_slot<G>(q1, t2 + 8 * q)] = v[u * G::R2 + q];
                else lds[t2 * G::SB + q1 + G::R1 * q] = v[u * G::R2 + q];
            }
        }
        sgb_valu_then_write<4, 16>();
    } else {
#pragma unroll
    for (int u = 0; u < G::NB2; u++) apply_tw<false, 1, G::R2>(&v[u * G::R2], &w2[u * G::R2]);
    st.mark(3);
    if constexpr (!INPL) sync_before_write(sync);
#pragma unroll
    for (int u = 0; u < G::NB2; u++) {
        const int beta = t + G::T * u, q1 = beta % G::R1, t2 = beta / G::R1;
#pragma unroll
        for (int q = 0; q < G::R2; q++) {
            if constexpr (INPL) lds[inpl_slot<G>(q1, t2 + 8 * q)] = v[u * G::R2 + q];
            else lds[t2 * G::SB + q1 + G::R1 * q] = v[u * G::R2 + q];
        }
    }
    }
    sync();
    st.mark(4);
    // ---- pass 3: the mirrored radix-8 butterflies bA, bB
    const int bA = t, bB = t ? G::B3 - t : G::B3 / 2;
    if constexpr (INPL) {   // butterfly b = q1 + R1 q2: its operands t2 = 0 .. 7 sit side by side in row q1
#pragma unroll
        for (int r = 0; r < 8; r++) { v[r] = lds[inpl_slot<G>(bA % G::R1, 8 * (bA / G::R1) + r)]; v[8 + r] = lds[inpl_slot<G>(bB % G::R1, 8 * (bB / G::R1) + r)]; }
    } else {
#pragma unroll
    for (int r = 0; r < 8; r++) { v[r] = lds[r * G::SB + bA]; v[8 + r] = lds[r * G::SB + bB]; }
    }
    if constexpr (!(PAIRED & 1)) sync_reads_done(sync);
    st.mark(5);
    dft8<false, 1>(&v[0]);
    dft8<false, 1>(&v[8]);
    st.mark(6);
}

// ---------------------------------------------------------------------------------------------- inverse
// In: v[0..7] = Z'[bA + B3*r], v[8..15] = Z'[bB + B3*r].  Out: v[j] = sum_k Z'[k] e^{+2 pi i nk/C}, n = t + T*j
// (unnormalised; the 1/N of fft.IFFT is folded into the target magnitudes).
// `w1_out` (optional): receives this lane's pass-1 twiddles as they are read (index u + NB1*q, the layout of Twiddles::w1r), so that a
// kernel whose next step is a forward FFT can hand them to it instead of reading the table again.
template <class G, int PAIRED = 0, class V, class SyncT, class St> GL_DEV void fft_inverse(V (&v)[16], V* lds, int t, const Twiddles<V>& tw, SyncT sync, St& st, V* w1_out = nullptr) {
    const int bA = t, bB = t ? G::B3 - t : G::B3 / 2;
    constexpr bool INPL = (PAIRED & 4) && G::NB1 == 1 && G::NB2 == 1;   // (fft_forward)
    static_assert(!(PAIRED & 4) || INPL, "the in-place exchange is built for one butterfly per lane and pass");
    // pass 3's slots: layout B, or (INPL) eight consecutive slots of row q1 of layout A
    auto s3 = [&](int b, int r) { if constexpr (INPL) return inpl_slot<G>(b % G::R1, 8 * (b / G::R1) + r); else return r * G::SB + b; };
    constexpr bool ILV = GL_D_INTERLEAVE && __is_same(V, v2d) && IsSplitSync<SyncT>::value;
    if constexpr (ILV) {   // the first butterfly's eight outputs leave while the second one computes
        if constexpr (!(PAIRED & 1)) sync_before_write(sync);
        dft8<true, 1>(&v[0]);
#pragma unroll
        for (int r = 0; r < 8; r++) lds[s3(bA, r)] = v[r];
        dft8<true, 1>(&v[8]);
#pragma unroll
        for (int r = 0; r < 8; r++) lds[s3(bB, r)] = v[8 + r];
        __builtin_amdgcn_sched_group_barrier(GL_SGB_VALU, 56, 0);
        sgb_valu_then_write<7, 8>();
        sgb_valu_then_write<0, 8>();
    } else {
    dft8<true, 1>(&v[0]);
    dft8<true, 1>(&v[8]);
    st.mark(8);
    if constexpr (!(PAIRED & 1)) sync_before_write(sync);
#pragma unroll
    for (int r = 0; r < 8; r++) { lds[s3(bA, r)] = v[r]; lds[s3(bB, r)] = v[8 + r]; }
    }
    sync();
    st.mark(9);
#pragma unroll
    for (int u = 0; u < G::NB2; u++) {
        const int beta = t + G::T * u, q1 = beta % G::R1, t2 = beta / G::R1;
        V w2[16];
#pragma unroll
        for (int q = 0; q < G::R2; q++) {
            if constexpr (INPL) v[u * G::R2 + q] = lds[inpl_slot<G>(q1, t2 + 8 * q)];
            else v[u * G::R2 + q] = lds[t2 * G::SB + q1 + G::R1 * q];
            if (q) w2[q] = tw.w2r ? tw.w2r[u * G::R2 + q] : tw.tw2[q * 8 + t2];
        }
        apply_tw<true, 1, G::R2>(&v[u * G::R2], w2);
    }
    if constexpr (!INPL) sync_reads_done(sync);
    st.mark(10);
#pragma unroll
    for (int u = 0; u < G::NB2; u++) {
        const int beta = t + G::T * u, q1 = beta % G::R1, t2 = beta / G::R1;
        dftR<true, G::R2, 1>(&v[u * G::R2]);
        if constexpr (!INPL) { if (u == 0) sync_before_write(sync); }
#pragma unroll
        for (int m = 0; m < G::R2; m++) { if constexpr (INPL) lds[inpl_slot<G>(q1, t2 + 8 * m)] = v[u * G::R2 + m]; else lds[q1 * G::SAI + t2 + 8 * m] = v[u * G::R2 + m]; }
    }
    sync();
    st.mark(11);
#pragma unroll
    for (int u = 0; u < G::NB1; u++) {
        const int t1 = t + G::T * u;
        V w1[16];
#pragma unroll
        for (int q = 0; q < G::R1; q++) {
            if constexpr (INPL) v[u + G::NB1 * q] = lds[inpl_slot<G>(q, t1)]; else v[u + G::NB1 * q] = lds[q * G::SAI + t1];
            if (q) w1[G::NB1 * q] = (tw.w1r && q <= tw.w1r_n) ? tw.w1r[u + G::NB1 * q] : tw.tw1[q * G::C1 + t1];
            if (q && w1_out) w1_out[u + G::NB1 * q] = w1[G::NB1 * q];
        }
        apply_tw<true, G::NB1, G::R1>(&v[u], w1);
    }
    if constexpr (!(PAIRED & 2)) sync_reads_done(sync);
    st.mark(12);
#pragma unroll
    for (int u = 0; u < G::NB1; u++) dftR<true, G::R1, G::NB1>(&v[u]);
    st.mark(13);
}

// the product form: no stamps
template <class G, int PAIRED = 0, class V, class SyncT> GL_DEV void fft_forward(V (&v)[16], V* lds, int t, const Twiddles<V>& tw, SyncT sync) {
    NoStamps st; fft_forward<G, PAIRED>(v, lds, t, tw, sync, st);
}
template <class G, int PAIRED = 0, class V, class SyncT> GL_DEV void fft_inverse(V (&v)[16], V* lds, int t, const Twiddles<V>& tw, SyncT sync) {
    NoStamps st; fft_inverse<G, PAIRED>(v, lds, t, tw, sync, st);
}

// ---------------------------------------------------------------------------------------------- spectral step
// Lane t's eight bin pairs after fft_forward: slot s pairs bin kA = t + B3*s (v[s]) with C - kA (v[15 - s]).
// Lane 0 is special: its butterflies are self-mirrored -- VA[r] = Z[B3 r] pairs with VA[8-r], VB[r] = Z[B3/2 + B3 r] with VB[7-r].  It is
// brought into the same slot shape by the CHEAPEST permutation (round 4: eight register moves in one chain instead of thirteen plus
// temporaries -- the moves are issued for the whole wave, ~80 -> ~35 instructions per frame in the fp64 kernel):
//   slots 0..3 -> (VB[s], VB[7-s])   v[15-s] already holds VB[7-s];  v[s]    <- v[8+s]
//   slot  4    -> (VA4, VA4)         the self-paired bin C/2;        v[11]   <- v[4]
//   slots 5..7 -> (VA[s], VA[8-s])   v[s] already holds VA[s];       v[15-s] <- v[8-s]
// DC and Nyquist (both in VA[0] = v[0], saved by the caller) are handled apart.
template <class G> GL_DEV int slot_bin(int t, int s) {
    if (t) return t + G::B3 * s;
    return s < 4 ? G::B3 / 2 + G::B3 * s : G::B3 * s;
}

template <class G, class V> GL_DEV void lane0_permute(V (&v)[16], bool is0) {
    if (is0) {   // a chain, v[0] (DC / Nyquist) is the caller's by now: no temporaries
        v[0] = v[8]; v[8] = v[1]; v[1] = v[9]; v[9] = v[2]; v[2] = v[10]; v[10] = v[3]; v[3] = v[11]; v[11] = v[4];
    }
}
template <class G, class V> GL_DEV void lane0_unpermute(V (&v)[16], bool is0, V dc) {
    if (is0) {
        // one lane in 2 T needs this: keep it a real branch (waves without a lane 0 skip it, the others run it with one lane
        // enabled) instead of full-width v_cndmask per frame -- the empty asm statement stops the if-conversion
        asm volatile("; lane 0 only" ::: );
        v[11] = v[3]; v[3] = v[10]; v[10] = v[2]; v[2] = v[9]; v[9] = v[1]; v[1] = v[8]; v[8] = v[0]; v[0] = dc;   // (v[11] held the self-paired bin's second copy)
    }
}

// unit phasor of X with the reference's zero rule: cmplx.Phase(0) = 0 -> Rect(M, 0) = M (mel/mel.go:100-101)
GL_DEV v2f unit_phase(v2f x) {
    float p = x.x * x.x + x.y * x.y;
    float r = __builtin_amdgcn_rsqf(p);   // v_rsq_f32: 1 ulp
    return p > 0.f ? v2f{x.x * r, x.y * r} : v2f{1.f, 0.f};
}
// the same for inputs whose squared modulus under/overflows fp32 (rare path)
GL_DEV v2f unit_phase_safe(v2f x) {
    const float big = fmaxf(fabsf(x.x), fabsf(x.y));
    if (!(big > 0.f)) return v2f{1.f, 0.f};
    const float sc = 1.0f / big;            // bring the larger component to 1: no under/overflow in the square
    const v2f y = v2f{x.x * sc, x.y * sc};
    const float r = __builtin_amdgcn_rsqf(y.x * y.x + y.y * y.y);
    return v2f{y.x * r, y.y * r};
}

// Real-FFT split -> X[k], X[C-k]; Y = Ms * X/|X|; merge -> Z'[k], Z'[C-k].  ma/mb are the (1/N-scaled) target
// magnitudes of bins k and C-k, w = W_N^k.  a = Z[k], b = Z[C-k] on entry; Z'[k], Z'[C-k] on exit.
GL_DEV void clamp_pair_c(v2f& a, v2f& b, v2f w, float ma, float mb) {
    v2f bc = v2f{b.x, -b.y};
    v2f p = a + bc;                      // 2 E_k
    v2f d = a - bc;                      // 2 i O_k
    v2f q = cmul(w, rot_mi(d));          // 2 w O_k
    v2f xk = p + q;                      // 2 X[k]
    v2f xc = p - q;                      // 2 conj(X[C-k])
    v2f u = unit_phase(xk) * ma;         // Y[k]
    v2f vv = unit_phase(xc) * mb;        // conj(Y[C-k])  (conj of a unit phasor times a real)
    v2f e = u + vv;                      // 2 E'
    v2f o = cmulc(u - vv, w);            // 2 O' = conj(w) (U - V)
    v2f io = rot_pi(o);                  // i O'
    a = e + io;                          // E' + i O'
    v2f t = e - io;                      // E' - i O'
    b = v2f{t.x, -t.y};                  // conj(E' - i O')
}

// Two bin pairs at once in packed math with op_sel/neg modifiers (same arithmetic as clamp_pair_c, 28 VALU per pair
// instead of ~60).  The two pairs are interleaved so that no instruction consumes the result of the one before it.
GL_DEV void clamp_pairs2(v2f& a0, v2f& b0, v2f w0, float ma0, float mb0, v2f& a1, v2f& b1, v2f w1, float ma1, float mb1) {
#if GL_PKASM
    {   // a <- 2 X[k] = p + q, b <- 2 conj(X[C-k]) = p - q, with p = a + conj(b), q = w * (-i)(a - conj(b))
        v2f p0, q0, t0, p1, q1, t1;
        asm(GL_LEAD_NOP
            "v_pk_add_f32 %0, %6, %7 neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %3, %8, %9 neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %1, %6, %7 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
            "v_pk_add_f32 %4, %8, %9 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
            "v_pk_mul_f32 %2, %1, %10 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %5, %4, %11 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %1, %1, %10, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
            "v_pk_fma_f32 %4, %4, %11, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
            "v_pk_add_f32 %6, %0, %1\n\t"
            "v_pk_add_f32 %8, %3, %4\n\t"
            "v_pk_add_f32 %7, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %9, %3, %4 neg_lo:[0,1] neg_hi:[0,1]"
            : "=&v"(p0), "=&v"(q0), "=&v"(t0), "=&v"(p1), "=&v"(q1), "=&v"(t1), "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1)
            : "v"(w0), "v"(w1));
    }
    // phasors: |x|^2, v_rsq, scale by the target magnitude
    const float pa0 = a0.x * a0.x + a0.y * a0.y, pb0 = b0.x * b0.x + b0.y * b0.y;
    const float pa1 = a1.x * a1.x + a1.y * a1.y, pb1 = b1.x * b1.x + b1.y * b1.y;
    v2f u0 = a0 * (__builtin_amdgcn_rsqf(pa0) * ma0), v0 = b0 * (__builtin_amdgcn_rsqf(pb0) * mb0);
    v2f u1 = a1 * (__builtin_amdgcn_rsqf(pa1) * ma1), v1 = b1 * (__builtin_amdgcn_rsqf(pb1) * mb1);
    // rare, wave-uniform: a modulus that is exactly zero (reference: phase 0, Y = M) or outside rsq's safe range
    const float plo = fminf(fminf(pa0, pb0), fminf(pa1, pb1)), phi = fmaxf(fmaxf(pa0, pb0), fmaxf(pa1, pb1));
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(plo > 1e-30f) || !(phi < 1e30f)) != 0, 0)) {
        u0 = unit_phase_safe(a0) * ma0; v0 = unit_phase_safe(b0) * mb0;
        u1 = unit_phase_safe(a1) * ma1; v1 = unit_phase_safe(b1) * mb1;
    }
    v2f e0 = u0 + v0, f0 = u0 - v0, e1 = u1 + v1, f1 = u1 - v1;
    {   // o = conj(w) f;  a <- e + i o;  b <- conj(e - i o)
        v2f t0, t1;
        asm(GL_LEAD_NOP
            "v_pk_mul_f32 %0, %3, %6 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %5, %7 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %3, %6, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 %1, %5, %7, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
            "v_pk_add_f32 %3, %2, %0 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
            "v_pk_add_f32 %5, %4, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
            "v_pk_add_f32 %2, %2, %0 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
            "v_pk_add_f32 %4, %4, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]"
            : "=&v"(t0), "=&v"(t1), "+v"(e0), "+v"(f0), "+v"(e1), "+v"(f1)
            : "v"(w0), "v"(w1));
    }
    a0 = f0; b0 = e0; a1 = f1; b1 = e1;   // block 2 leaves Z'[k] in f, Z'[C-k] in e
#else
    clamp_pair_c(a0, b0, w0, ma0, mb0);
    clamp_pair_c(a1, b1, w1, ma1, mb1);
#endif
}

// Target magnitudes a lane needs for one frame.  A frame's row of N/2 magnitudes is stored in LANE-SLOT order
// (mag_slot_index below): entry e of lane t sits at float (e/4 * T + t) * 4 + e%4, e < 8: mag[k_e(t)], e >= 8:
// mag[C - k_{e-8}(t)] -- so a lane fetches everything with four coalesced 16-byte loads off one base pointer, issued
// early so the HBM latency hides behind the forward FFT.  Lane 0's duplicate entry (slot 4 pairs bin C/2 with itself)
// carries bin 0 (DC) instead.
struct MagRegs {
    float ma[8], mb[8];
    float m0, mn;  // bins 0 and N/2 (lane 0 only)
};
typedef float v4f __attribute__((ext_vector_type(4)));

// natural bin k in [0, C)  ->  float index inside the permuted row
template <class G> __host__ __device__ inline int mag_slot_index(int k) {
    const int r = k % G::B3, s = k / G::B3;
    int t, e;
    if (r == 0) { t = 0; e = (k == 0) ? 8 + 4 : (s == 4 ? 4 : (s > 4 ? s : 8 + (8 - s))); }   // lane 0: DC rides in slot 4's duplicate entry, VA[s] s > 4 -> ma[s], VA[s] s < 4 -> mb[8-s]
    else if (r == G::B3 / 2) { t = 0; e = (s < 4) ? s : 8 + (7 - s); }                          //         VB[s] s < 4 -> ma[s], s >= 4 -> mb[7-s]
    else if (r < G::B3 / 2) { t = r; e = s; }
    else { t = G::B3 - r; e = 8 + (7 - s); }
    return ((e / 4) * G::T + t) * 4 + (e % 4);
}

// The loads are issued EARLY (top of the frame) and must not be waited for until the spectral step, a whole forward FFT later:
// MagRaw holds the untouched load results; finish_mags() -- lane 0's fix-up included -- runs after fft_forward.  (Fixing lane 0 up
// right after the loads made the wave sit out the full HBM latency of the magnitude row at the top of every frame.)
// The Nyquist magnitude is fetched by a VECTOR load (lane t reads nyq[t & 3]; only lane 0's value is used; the array is padded
// by 4 floats): a uniform address would become a scalar load, whose out-of-order return turns every counted LDS wait behind it
// into lgkmcnt(0).
struct MagRaw { v4f c0, c1, c2, c3; float nq; };
template <class G> GL_DEV void load_mag_rows(MagRaw& r, int t, const float* __restrict__ mag_row) {
    const v4f* r4 = reinterpret_cast<const v4f*>(mag_row);
    r.c0 = r4[t]; r.c1 = r4[G::T + t]; r.c2 = r4[2 * G::T + t]; r.c3 = r4[3 * G::T + t];
}
template <class G> GL_DEV void load_mags_raw(MagRaw& r, int t, const float* __restrict__ mag_row, const float* __restrict__ nyq) {
    load_mag_rows<G>(r, t, mag_row);
    r.nq = nyq[t & 3];
}
GL_DEV void finish_mags(MagRegs& m, const MagRaw& r, bool is0) {
    const v4f c0 = r.c0, c1 = r.c1, c2 = r.c2, c3 = r.c3;
    m.ma[0] = c0.x; m.ma[1] = c0.y; m.ma[2] = c0.z; m.ma[3] = c0.w;
    m.ma[4] = c1.x; m.ma[5] = c1.y; m.ma[6] = c1.z; m.ma[7] = c1.w;
    m.mb[0] = c2.x; m.mb[1] = c2.y; m.mb[2] = c2.z; m.mb[3] = c2.w;
    m.mb[4] = c3.x; m.mb[5] = c3.y; m.mb[6] = c3.z; m.mb[7] = c3.w;
    m.m0 = is0 ? m.mb[4] : 0.f;
    m.mn = is0 ? r.nq : 0.f;
    m.mb[4] = is0 ? m.ma[4] : m.mb[4];
}
template <class G> GL_DEV void load_mags(MagRegs& m, int t, const float* __restrict__ mag_row, const float* __restrict__ nyq) {
    MagRaw r;
    load_mags_raw<G>(r, t, mag_row, nyq);
    finish_mags(m, r, t == 0);
}
// Wave-uniform 64-bit offset (SGPR pair): `kernel_arg_pointer + uniform offset + 32-bit lane index` lets the loads
// use "SGPR base + VGPR offset" addressing and keeps the global address space (a pointer rebuilt from integers would
// degrade to flat_load).  Only valid where every lane of the wave holds the same value.
GL_DEV long long uniform_i64(long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
// The eight W_N^{k_s} of a lane from ONE table value: k_s = t + B3*s and N/B3 = 16, so W_N^{k_s} = W_N^t * W_16^s (lane 0's
// permuted slots get their constants).  Costs a few multiplies per frame, saves 14 registers in the fused kernel.
template <class G, class V> GL_DEV void make_wn(V (&w)[8], int t, V wt) {
    typedef typename ScalarOf<V>::type S;
    constexpr double c16[8] = {1.0, kC1, kS2, kS1, 0.0, -kS1, -kS2, -kC1};
    constexpr double s16[8] = {0.0, kS1, kS2, kC1, 1.0, kC1, kS2, kS1};
    // W_32^{1,3,5,7} = cos/sin(pi m/16)
    constexpr double c32[4] = {0.98078528040323044913, 0.83146961230254523708, 0.55557023301960222474, 0.19509032201612826785};
    constexpr double s32[4] = {0.19509032201612826785, 0.55557023301960222474, 0.83146961230254523708, 0.98078528040323044913};
    const bool is0 = (t == 0);
#pragma unroll
    for (int s = 0; s < 8; s++) {
        V g = twc<false>(wt, c16[s], s16[s]);  // wt * W_16^s
        V z = s < 4 ? V{(S)c32[s], (S)-s32[s]} : (s == 4 ? V{(S)0, (S)-1} : V{(S)c16[s], (S)-s16[s]});   // W_N^{slot_bin(0, s)}
        w[s] = is0 ? z : g;
    }
}
template <class G> GL_DEV void load_wn(v2f (&w)[8], int t, const v2f* __restrict__ wn) {
#pragma unroll
    for (int s = 0; s < 8; s++) w[s] = wn[slot_bin<G>(t, s)];
}

// The whole spectral step on a lane's 16 registers (mel/mel.go:98-108 in the real-FFT form).
template <class G> GL_DEV void spectral_apply(v2f (&v)[16], int t, const MagRegs& m, const v2f (&w)[8]) {
    const bool is0 = (t == 0);
    v2f dc = v[0];
    lane0_permute<G>(v, is0);
#pragma unroll
    for (int s = 0; s < 8; s += 2) clamp_pairs2(v[s], v[15 - s], w[s], m.ma[s], m.mb[s], v[s + 1], v[14 - s], w[s + 1], m.ma[s + 1], m.mb[s + 1]);
    if (is0) {
        // DC and Nyquist: X[0] = Re+Im, X[N/2] = Re-Im (real); Y = +-M; Z'[0] = (Y0+Yn) + i (Y0-Yn)
        float x0 = dc.x + dc.y, xn = dc.x - dc.y;
        float y0 = x0 < 0.f ? -m.m0 : m.m0;
        float yn = xn < 0.f ? -m.mn : m.mn;
        dc = v2f{y0 + yn, y0 - yn};
    }
    lane0_unpermute<G>(v, is0, dc);
}

// ---------------------------------------------------------------------------------------------- fp64 spectral step ("precise" mode)
// Same arithmetic as the fp32 forms above, in double: used by the generic kernels when the handle's precision is 64 bits.
struct MagRegsD {
    double ma[8], mb[8];
    double m0, mn;
};
template <class G> GL_DEV void load_mags(MagRegsD& m, int t, const double* __restrict__ mag_row, const double* __restrict__ nyq) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
        m.ma[e] = mag_row[((e / 4) * G::T + t) * 4 + (e % 4)];
        m.mb[e] = mag_row[(((8 + e) / 4) * G::T + t) * 4 + (e % 4)];
    }
    m.m0 = 0.0; m.mn = 0.0;
    if (t == 0) { m.m0 = m.mb[4]; m.mb[4] = m.ma[4]; m.mn = nyq[0]; }
}
template <class G> GL_DEV void load_wn(v2d (&w)[8], int t, const v2d* __restrict__ wn) {
#pragma unroll
    for (int s = 0; s < 8; s++) w[s] = wn[slot_bin<G>(t, s)];
}
GL_DEV v2d unit_phase_d(v2d x) {   // cmplx.Phase(0) = 0 -> (1, 0); scaled so that the squared modulus cannot under/overflow
    const double big = fmax(fabs(x.x), fabs(x.y));
    if (!(big > 0.0)) return v2d{1.0, 0.0};
    const double sc = 1.0 / big;
    const v2d y = v2d{x.x * sc, x.y * sc};
    const double r = 1.0 / sqrt(y.x * y.x + y.y * y.y);
    return v2d{y.x * r, y.y * r};
}
// 1/sqrt(p) for p a positive normal number: v_rsq_f64 (16.5 cycles: four times an fp64 multiply-add, profiles/r05/mfma_f64_probe.txt) as the
// seed and ONE third-order step, y (1 + e/2 + 3 e^2 / 8) with e = 1 - p y^2 (error ~ e^3: within 2 ulp for any seed good to 2^-20) -- five
// multiply-adds where an IEEE 1.0 / sqrt(p) expands to ~26 instructions and the two coupled Goldschmidt steps of rounds 3-4 took seven;
// no intermediate leaves the normal range when p is normal, so the caller's range test is "p is a positive normal number" (v_cmp_class_f64)
GL_DEV double rsqrt_d(double p) {
    const double y = __builtin_amdgcn_rsq(p);
    const double e = __builtin_fma(-(p * y), y, 1.0);
    const double c = __builtin_fma(e, 0.375, 0.5);
    return __builtin_fma(y, c * e, y);
}
// GL_D_CLAMP_PATCH: the rare scaled form of the phasor as a patch BEHIND the common form (one branch per bin pair) instead of an if / else around both (two branches
// and a flag per pair in the compiled code: 34 scalar instructions and 8 branches more per frame).  Bit-identical; k_gl_fused_d<4096,5> -0.4 ... -0.7 % (n = 8 and 12,
// one box each), -0.35 % at N 2048 / hop 256 (n = 10), +1.5 % at N 1024 (n = 6), the persistent kernel +-0: taken from N 2048 on.  profiles/r06/ab_clamp_patch.txt
#ifndef GL_D_CLAMP_PATCH
#define GL_D_CLAMP_PATCH 1
#endif
#ifndef GL_D_CLAMP_PATCH_MINN
#define GL_D_CLAMP_PATCH_MINN 2048
#endif
template <bool PATCH> GL_DEV void clamp_pair_d(v2d& a, v2d& b, v2d w, double ma, double mb) {
    v2d bc = v2d{b.x, -b.y};
    v2d p = a + bc, d = a - bc;
    v2d q = cmul(w, rot_mi(d));
    v2d xk = p + q, xc = p - q;
    // Y = M X/|X| (mel/mel.go:98-102): unit phasor by reciprocal square root of the squared modulus; a modulus that is exactly zero
    // (the reference: phase 0, Y = M) or whose square is not a positive normal number takes the scaled form -- rare, wave-uniform branch
    const double pk = __builtin_fma(xk.x, xk.x, xk.y * xk.y), pc = __builtin_fma(xc.x, xc.x, xc.y * xc.y);
    v2d e, f;
    if constexpr (PATCH) {
    // the common form unconditionally, the scaled form as a PATCH behind it: compiles to ONE branch per pair, where the if / else spelling below
    // compiles to two branches and a flag
    {
        const double fk = rsqrt_d(pk) * ma, fc = rsqrt_d(pc) * mb;
        const v2d vv = xc * fc, kk = v2d{fk, fk};
        e = vfma(xk, kk, vv); f = vfma(xk, kk, -vv);
    }
#ifndef GOMEL_COUNT_ONLY
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!__builtin_amdgcn_class(pk, 0x100) || !__builtin_amdgcn_class(pc, 0x100)) != 0, 0)) {
        const v2d u = unit_phase_d(xk) * ma, vv = unit_phase_d(xc) * mb;
        e = u + vv; f = u - vv;
    }
#endif
    } else {
#ifdef GOMEL_COUNT_ONLY   // instruction-count audits: the rare branch compiled out (never shipped)
    if (false) {
#else
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!__builtin_amdgcn_class(pk, 0x100) || !__builtin_amdgcn_class(pc, 0x100)) != 0, 0)) {
        const v2d u = unit_phase_d(xk) * ma, vv = unit_phase_d(xc) * mb;
        e = u + vv; f = u - vv;
#endif
    } else {
        const double fk = rsqrt_d(pk) * ma, fc = rsqrt_d(pc) * mb;
        const v2d vv = xc * fc, kk = v2d{fk, fk};
        e = vfma(xk, kk, vv); f = vfma(xk, kk, -vv);       // Y[k] +- conj(Y[C-k]) with the first product riding on the multiply-adds
    }
    }
    v2d o = cmulc(f, w), io = rot_pi(o);
    a = e + io;
    v2d t = e - io;
    b = v2d{t.x, -t.y};
}
template <class G> GL_DEV void spectral_apply(v2d (&v)[16], int t, const MagRegsD& m, const v2d (&w)[8]) {
    const bool is0 = (t == 0);
    v2d dc = v[0];
    lane0_permute<G>(v, is0);
#pragma unroll
    for (int s = 0; s < 8; s++) clamp_pair_d<(GL_D_CLAMP_PATCH && G::N >= GL_D_CLAMP_PATCH_MINN)>(v[s], v[15 - s], w[s], m.ma[s], m.mb[s]);
    if (is0) {
        const double x0 = dc.x + dc.y, xn = dc.x - dc.y;
        const double y0 = x0 < 0.0 ? -m.m0 : m.m0, yn = xn < 0.0 ? -m.mn : m.mn;
        dc = v2d{y0 + yn, y0 - yn};
    }
    lane0_unpermute<G>(v, is0, dc);
}

}  // namespace glfft
