// gl_any.hpp -- Griffin-Lim / STFT kernels for the frame lengths the register FFT cores do not cover.
//
// The reference takes ANY Resolut: go-dsp's fft.FFTReal / fft.IFFT (mel/mel.go:95,116) run radix-2 for powers of two and
// Bluestein's chirp-z otherwise.  The tuned kernels (gl_fft.hpp) exist for N = 256 ... 4096; everything else that is even and
//   * a power of two up to 8192, or
//   * any length up to 8192 (Bluestein through a power-of-two convolution of >= 2N-1 points: up to 8192 of them in one piece, 16384 --
//     for lengths above 4096 -- in two halves, dft_bluestein_split)
// (odd lengths included: Bluestein does not care, and the reference's index arithmetic for an odd N is followed bin by bin)
// runs here: one 1024-thread workgroup per frame, the frame as M complex doubles in LDS (up to 128 KiB of the CU's 160),
// plain in-place radix-2 passes -- decimation in frequency forward (natural in, bit-reversed out) and decimation in time inverse
// (bit-reversed in, natural out), so no permutation pass is ever needed.  The arithmetic follows the reference's LITERAL form:
// full-length complex DFT of the windowed frame, Rect(|spec|, Phase(S)) on bins 0 ... N/2, conjugate fill, full inverse, real
// part (mel/mel.go:87-125).  fp64 throughout; correctness path, not tuned (HBM traffic ~3 N doubles per frame-iteration).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gl_fft.hpp"

namespace glany {
using glfft::v2d;

struct AnyTables {
    const v2d* tw;     // W_M^j = e^{-2 pi i j / M}, j < M/2
    const v2d* chirp;  // Bluestein: w[n] = e^{-i pi n^2 / N}, n < N
    const v2d* bhat;   // Bluestein: FFT_M of b[n] = conj(w[|n|]) (wrapped), in the bit-reversed order fft_dif leaves
                       // (split form, bluestein == 2: FFT_2M of b, its even bins [M] then its odd bins [M], each in that order)
    const v2d* omega;  // split form only: W_2M^n = e^{-i pi n / M}, n < M
    int M, logM, bluestein;   // bluestein: 0 power of two, 1 Bluestein through an M-point convolution, 2 through a 2M-point one in two halves
};
constexpr int kAnyThreads = 1024;

__device__ __forceinline__ v2d cmuld(v2d a, v2d b) { return v2d{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ v2d conjd(v2d a) { return v2d{a.x, -a.y}; }
__device__ __forceinline__ unsigned bitrev(unsigned i, int bits) { return __brev(i) >> (32 - bits); }

// in-place radix-2, decimation in frequency: natural order in, bit-reversed order out.  INV: e^{+} twiddles (unnormalised).
template <bool INV> __device__ void fft_dif(v2d* buf, const AnyTables& at) {
    const int M = at.M, logM = at.logM;
    for (int s = 0; s < logM; s++) {
        const int half = M >> (s + 1);
        for (int idx = threadIdx.x; idx < M / 2; idx += kAnyThreads) {
            const int grp = idx >> (logM - 1 - s), pos = idx & (half - 1);
            const int i0 = (grp << (logM - s)) + pos, i1 = i0 + half;
            const v2d a = buf[i0], b = buf[i1];
            v2d w = at.tw[pos << s];
            if (INV) w = conjd(w);
            buf[i0] = a + b;
            buf[i1] = cmuld(a - b, w);
        }
        __syncthreads();
    }
}
// in-place radix-2, decimation in time: bit-reversed order in, natural order out.
template <bool INV> __device__ void fft_dit(v2d* buf, const AnyTables& at) {
    const int M = at.M, logM = at.logM;
    for (int s = 0; s < logM; s++) {
        const int half = 1 << s;
        for (int idx = threadIdx.x; idx < M / 2; idx += kAnyThreads) {
            const int grp = idx >> s, pos = idx & (half - 1);
            const int i0 = (grp << (s + 1)) + pos, i1 = i0 + half;
            v2d w = at.tw[pos << (logM - 1 - s)];
            if (INV) w = conjd(w);
            const v2d a = buf[i0], b = cmuld(buf[i1], w);
            buf[i0] = a + b;
            buf[i1] = a - b;
        }
        __syncthreads();
    }
}

// Forward DFT_N (e^{-}, unnormalised) of buf[0, N) for a length that is not a power of two: Bluestein.  Natural order in and out.
__device__ void dft_bluestein(v2d* buf, int N, const AnyTables& at) {
    const int M = at.M;
    for (int n = threadIdx.x; n < M; n += kAnyThreads) buf[n] = n < N ? cmuld(buf[n], at.chirp[n]) : v2d{0.0, 0.0};
    __syncthreads();
    fft_dif<false>(buf, at);
    for (int i = threadIdx.x; i < M; i += kAnyThreads) buf[i] = cmuld(buf[i], at.bhat[i]);
    __syncthreads();
    fft_dit<true>(buf, at);
    const double sc = 1.0 / (double)M;
    for (int k = threadIdx.x; k < N; k += kAnyThreads) { const v2d c = cmuld(buf[k], at.chirp[k]); buf[k] = v2d{c.x * sc, c.y * sc}; }
    __syncthreads();
}

// The same for 4096 < N <= 8192 (not a power of two): the convolution needs 2N-1 <= 16384 points, 256 KB of complex doubles -- more than
// a CU's LDS.  The zero-padded input fills only the lower half of those 2M points (N <= M = 8192), so one decimation-in-frequency step
// splits FFT_2M(a) into two M-point transforms of the SAME M inputs -- even bins: FFT_M(a), odd bins: FFT_M(a[n] W_2M^n) -- and the
// inverse, of which only the first N outputs are wanted, is z[m] = (E[m] + conj(W_2M^m) O[m]) / 2M with E, O the M-point inverse
// transforms of the even / odd bins.  Four M-point FFTs in the one 128 KB buffer; the input and the even half's result wait in
// registers (8 + 8 complex doubles per thread).
__device__ void dft_bluestein_split(v2d* buf, int N, const AnyTables& at) {
    const int M = at.M;
    constexpr int PER = 8192 / kAnyThreads;           // M == 8192 here
    v2d a[PER], e[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) { const int n = threadIdx.x + kAnyThreads * i; a[i] = n < N ? cmuld(buf[n], at.chirp[n]) : v2d{0.0, 0.0}; }
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
        for (int i = 0; i < PER; i++) { const int n = threadIdx.x + kAnyThreads * i; buf[n] = half ? cmuld(a[i], at.omega[n]) : a[i]; }
        __syncthreads();
        fft_dif<false>(buf, at);
        const v2d* bh = at.bhat + (size_t)half * M;
        for (int i = threadIdx.x; i < M; i += kAnyThreads) buf[i] = cmuld(buf[i], bh[i]);
        __syncthreads();
        fft_dit<true>(buf, at);
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < PER; i++) e[i] = buf[threadIdx.x + kAnyThreads * i];
            __syncthreads();
        }
    }
    const double sc = 1.0 / (2.0 * (double)M);
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int k = threadIdx.x + kAnyThreads * i;
        if (k < N) {
            const v2d z = e[i] + cmuld(buf[k], conjd(at.omega[k]));
            const v2d c = cmuld(z, at.chirp[k]);
            buf[k] = v2d{c.x * sc, c.y * sc};
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void dft_any(v2d* buf, int N, const AnyTables& at) {   // forward DFT_N of a length that is not a power of two
    if (at.bluestein == 2) dft_bluestein_split(buf, N, at); else dft_bluestein(buf, N, at);
}

__device__ __forceinline__ v2d unit_phase_any(v2d x) {   // cmplx.Phase(0) = 0 -> (1, 0) (mel/mel.go:100-101)
    const double big = fmax(fabs(x.x), fabs(x.y));
    if (!(big > 0.0)) return v2d{1.0, 0.0};
    const double sc = 1.0 / big, yr = x.x * sc, yi = x.y * sc, r = 1.0 / sqrt(yr * yr + yi * yi);
    return v2d{yr * r, yi * r};
}

// One Griffin-Lim iteration for one frame (MODE kAnyGl): x -> windowed frame -> DFT -> clamp -> IDFT -> windowed frame yw[frame][N];
// or the STFT magnitudes of ToMel (kAnyToMel): |S[k]|, k = 0 ... N/2 -> smag[frame][N/2+1]; or ISTFT's side effect on its
// spectrogram argument (kAnySpec, mel/mel.go:98-102): Rect(|spectrogram[i][k]|, Phase(S[k])) for the bins that keep a magnitude of
// their own, k = 0 ... N/2 (odd N: ... N/2+1) -> spec[frame][that many] complex (what the clamp step of an iteration on x leaves
// in those bins; the shim mirrors the rest).
constexpr int kAnyGl = 0, kAnyToMel = 1, kAnySpec = 2;
template <int MODE>
__global__ __launch_bounds__(kAnyThreads) void k_any_frames(const double* __restrict__ x, const int64_t* __restrict__ frame_xoff,
                                                            const double* __restrict__ mag, const double* __restrict__ nyq,
                                                            double* __restrict__ out, int64_t n_frames, int N,
                                                            const double* __restrict__ window, AnyTables at, const double* __restrict__ mid = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char any_smem[];
    v2d* buf = reinterpret_cast<v2d*>(any_smem);
    const int64_t frame = blockIdx.x;
    const int H = N / 2, logM = at.logM;
    const double* xs = x + frame_xoff[frame];
    for (int n = threadIdx.x; n < N; n += kAnyThreads) buf[n] = v2d{xs[n] * window[n], 0.0};   // mel/mel.go:88-94
    __syncthreads();
    if (at.bluestein) dft_any(buf, N, at); else fft_dif<false>(buf, at);                         // fft.FFTReal, mel/mel.go:95
    auto pos = [&](int k) { return at.bluestein ? k : (int)bitrev((unsigned)k, logM); };
    if (MODE == kAnyToMel) {
        double* o = out + frame * (H + 1);
        for (int k = threadIdx.x; k <= H; k += kAnyThreads) { const v2d s = buf[pos(k)]; o[k] = hypot(s.x, s.y); }
        return;
    }
    // mel/mel.go:98-102 on the bins whose magnitudes survive the conjugate fill of :105-108 (j = 1 ... N/2 - 1 -> N - j).
    // Even N: bins 0 ... N/2, bin N/2 carrying channel 1's last bin (undospectrum puts ch1[j] at N-1-j, mel/impl.go:402-403).
    // Odd N = 2H+1: bins 0 ... H-1 from channel 0, bin H is never written by undospectrum (magnitude 0), and bin H+1 = N-H --
    // out of the fill's reach -- keeps channel 1's last bin WITH ITS OWN PHASE; the spectrum is then not Hermitian and the
    // reference simply takes the real part of the inverse (mel/mel.go:120), as the last line of this kernel does.
    const bool odd = (N & 1) != 0;
    const int own_top = odd ? H + 1 : H;                 // highest bin that keeps a magnitude of its own
    for (int k = threadIdx.x; k <= own_top; k += kAnyThreads) {
        const int p = pos(k);
        // (odd N, bin H: nothing of a log-mel input lands there; the exported ISTFT's spectrogram has a magnitude for it -- `mid`)
        const double m = k < H ? mag[frame * H + k] : (k == own_top ? nyq[frame] : (mid ? mid[frame] : 0.0));
        const v2d u = unit_phase_any(buf[p]);
        if (MODE == kAnySpec) reinterpret_cast<v2d*>(out)[frame * (own_top + 1) + k] = v2d{u.x * m, u.y * m};
        else buf[p] = v2d{u.x * m, u.y * m};
    }
    if (MODE == kAnySpec) return;
    __syncthreads();
    for (int k = own_top + 1 + threadIdx.x; k < N; k += kAnyThreads) buf[pos(k)] = conjd(buf[pos(N - k)]);
    __syncthreads();
    if (at.bluestein) {   // inverse DFT = conj(forward DFT(conj(.)))
        for (int k = threadIdx.x; k < N; k += kAnyThreads) buf[k] = conjd(buf[k]);
        __syncthreads();
        dft_any(buf, N, at);
    } else {
        fft_dit<true>(buf, at);
    }
    const double sc = 1.0 / (double)N;                                                          // fft.IFFT's 1/N, mel/mel.go:116
    double* o = out + frame * (int64_t)N;
    for (int n = threadIdx.x; n < N; n += kAnyThreads) o[n] = buf[n].x * sc * window[n];         // real part either way; mel/mel.go:120
}

}  // namespace glany
