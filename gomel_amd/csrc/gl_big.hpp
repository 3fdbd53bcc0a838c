// gl_big.hpp -- Griffin-Lim / STFT for frame lengths above 8192 (up to 65536), where a frame no longer fits a CU's LDS.
//
// The reference takes ANY Resolut (go-dsp: radix-2 for powers of two, Bluestein otherwise; mel/mel.go:78,95,116).  gl_any.hpp keeps a
// whole frame in LDS, which ends at 8192 complex doubles.  Beyond that every DFT_N here is Bluestein's chirp-z through a power-of-two
// convolution of Mc >= 2N-1 points (32768 ... 131072) that lives in HBM, and each Mc-point FFT is ONE decimation-in-frequency step of
// radix R = Mc / 8192 (4, 8 or 16) in front of 8192-point transforms that run in LDS (gl_any.hpp's fft_dif):
//     X[r + R k'] = sum_{n' < 8192} [ (sum_{s < R} a[n' + 8192 s] W_R^{s r}) W_Mc^{n' r} ] W_8192^{n' k'},
// one 1024-thread workgroup per (frame, r): it reads the frame's Mc inputs, folds them, transforms, writes its 8192 bins (stride R).
// Inverse transforms are conj(FFT(conj(.))): a conjugation flag on the way in and out.  Per frame-iteration: four Mc-point FFTs (two per
// DFT_N), five pointwise passes.  fp64, the reference's LITERAL form (full complex DFT of the windowed frame, Rect(|spec|, Phase(S)) on
// the bins that keep a magnitude of their own, conjugate fill, full inverse, real part -- mel/mel.go:87-125), odd lengths included.
// A correctness path: ~40 Mc complex doubles of HBM traffic per frame-iteration; nobody tunes a 65536-point mel inversion.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gl_any.hpp"

namespace glbig {
using glany::cmuld;
using glany::conjd;
using glfft::v2d;

constexpr int kSub = 8192, kLogSub = 13;   // the LDS-resident transform
constexpr int kMaxN = 65536;               // 2N-1 <= 131072 = 16 x 8192

struct BigTables {
    const v2d* chirp;   // w[n] = e^{-i pi n^2 / N}, n < N
    const v2d* bhat;    // FFT_Mc of b[n] = conj(w[|n|]) (wrapped), NATURAL order, pre-divided by Mc (the convolution's inverse scale)
    int Mc, R;
};

// A[f][n] = x[xoff_f + n] * window[n] * chirp[n] for n < N, 0 up to Mc: the zero-padded, chirped input of a forward DFT_N (mel/mel.go:88-95)
__global__ __launch_bounds__(256) void k_big_prep(const double* __restrict__ x, const int64_t* __restrict__ frame_xoff, int64_t f0, const double* __restrict__ window,
                                                  v2d* __restrict__ A, int N, BigTables bt) {
    const int64_t f = blockIdx.y;
    const double* xs = x + frame_xoff[f0 + f];
    v2d* a = A + f * (int64_t)bt.Mc;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < bt.Mc; n += gridDim.x * blockDim.x) {
        v2d v = v2d{0.0, 0.0};
        if (n < N) { const double s = xs[n] * window[n]; const v2d c = bt.chirp[n]; v = v2d{s * c.x, s * c.y}; }
        a[n] = v;
    }
}

// One Mc-point FFT per frame, e^{-} (unnormalised): in -> out, natural order both.  conj_in / conj_out conjugate on the way in / out
// (inverse transforms); mul (nullable) multiplies bin k by mul[k] before the output conjugation (Bluestein's filter).
__global__ __launch_bounds__(glany::kAnyThreads) void k_big_fft(const v2d* __restrict__ in, v2d* __restrict__ out, BigTables bt, glany::AnyTables at8k,
                                                                const v2d* __restrict__ mul, int conj_in, int conj_out) {
    extern __shared__ __attribute__((aligned(16))) char big_smem[];
    v2d* buf = reinterpret_cast<v2d*>(big_smem);
    const int R = bt.R, r = blockIdx.x % R;
    const int64_t f = blockIdx.x / R;
    const v2d* a = in + f * (int64_t)bt.Mc;
    for (int n = threadIdx.x; n < kSub; n += glany::kAnyThreads) {
        v2d acc = v2d{0.0, 0.0};
        for (int s = 0; s < R; s++) {
            v2d v = a[n + kSub * s];
            if (conj_in) v = conjd(v);
            // W_R^{s r} = W_8192^{(s r mod R) 8192 / R}; the table holds the first half of the circle
            const int j = ((s * r) % R) * (kSub / R);
            v2d w = at8k.tw[j & (kSub / 2 - 1)];
            if (j >= kSub / 2) w = v2d{-w.x, -w.y};
            acc += cmuld(v, w);
        }
        double sn, cs;
        sincospi(-2.0 * (double)((int64_t)n * r) / (double)bt.Mc, &sn, &cs);   // W_Mc^{n r}
        buf[n] = cmuld(acc, v2d{cs, sn});
    }
    __syncthreads();
    glany::fft_dif<false>(buf, at8k);
    v2d* o = out + f * (int64_t)bt.Mc;
    for (int kp = threadIdx.x; kp < kSub; kp += glany::kAnyThreads) {
        const int k = r + R * kp;
        v2d v = buf[glany::bitrev((unsigned)kp, kLogSub)];
        if (mul) v = cmuld(v, mul[k]);
        if (conj_out) v = conjd(v);
        o[k] = v;
    }
}

// After the forward DFT_N's convolution C[f][k] (X[k] = chirp[k] C[k]): the clamp step mel/mel.go:98-108 and, fused with it, the input
// of the inverse DFT_N's Bluestein pass: B[n] = conj(S[n]) chirp[n] (n < N), 0 up to Mc, where S[k] = m_k X[k]/|X[k]| for the bins that
// keep a magnitude of their own (k <= own_top) and S[k] = conj(S[N-k]) above.  MODE 1 (ISTFT's write-back): spec[f][k] = S[k], k <= own_top.
template <int MODE>
__global__ __launch_bounds__(256) void k_big_clamp(const v2d* __restrict__ C, v2d* __restrict__ B, int64_t f0, const double* __restrict__ mag, const double* __restrict__ nyq,
                                                   const double* __restrict__ mid, double* __restrict__ spec, int N, BigTables bt) {
    const int64_t f = blockIdx.y, fg = f0 + f;
    const int H = N / 2;
    const bool odd = (N & 1) != 0;
    const int own_top = odd ? H + 1 : H;
    const v2d* c = C + f * (int64_t)bt.Mc;
    v2d* b = B + f * (int64_t)bt.Mc;
    const int lim = MODE == 1 ? own_top + 1 : bt.Mc;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < lim; n += gridDim.x * blockDim.x) {
        v2d v = v2d{0.0, 0.0};
        if (n < N) {
            const bool mirrored = n > own_top;
            const int k = mirrored ? N - n : n;
            const v2d X = cmuld(c[k], bt.chirp[k]);
            const double m = k < H ? mag[fg * H + k] : (k == own_top ? nyq[fg] : (mid ? mid[fg] : 0.0));
            const v2d u = glany::unit_phase_any(X);
            v2d S = v2d{u.x * m, u.y * m};
            if (mirrored) S = conjd(S);
            if (MODE == 1) { reinterpret_cast<v2d*>(spec)[fg * (own_top + 1) + n] = S; continue; }
            v = cmuld(conjd(S), bt.chirp[n]);
        }
        if (MODE == 0) b[n] = v;
    }
}

// y[n] = Re(IDFT_N(S))[n] * window[n] (mel/mel.go:116-120): D[n] = chirp[n] C2[n] = DFT_N(conj S)[n], IDFT = conj(D)/N, real part D.x/N
__global__ __launch_bounds__(256) void k_big_finish(const v2d* __restrict__ C2, int64_t f0, const double* __restrict__ window, double* __restrict__ yw, int N, BigTables bt) {
    const int64_t f = blockIdx.y;
    const v2d* c = C2 + f * (int64_t)bt.Mc;
    double* o = yw + (f0 + f) * (int64_t)N;
    const double sc = 1.0 / (double)N;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const v2d d = cmuld(c[n], bt.chirp[n]);
        o[n] = d.x * sc * window[n];
    }
}

// ToMel: |S[k]|, k = 0 ... N/2 (mel/mel.go:55-63); |chirp| = 1
__global__ __launch_bounds__(256) void k_big_absmag(const v2d* __restrict__ C, int64_t f0, double* __restrict__ smag, int N, BigTables bt) {
    const int64_t f = blockIdx.y;
    const int H = N / 2;
    const v2d* c = C + f * (int64_t)bt.Mc;
    double* o = smag + (f0 + f) * (int64_t)(H + 1);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= H; k += gridDim.x * blockDim.x) o[k] = hypot(c[k].x, c[k].y);
}

}  // namespace glbig
