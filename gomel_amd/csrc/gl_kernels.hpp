// gl_kernels.hpp -- the HIP kernels of libgomel_hip.so (gfx950).  Included by gomel_hip.hip (host side + C ABI).
// Reference steps replaced (file:line relative to the reference repository):
//   k_mel_to_mag      spectral_denormalize mel/impl.go:421-427 + undomel :347-384 + undospectrum :386-408
//   k_init_x          random init mel/mel.go:80-83 (explicit x0 or documented splitmix64 generator)
//   k_gl_fused        one Griffin-Lim iteration (mel/mel.go:87-136) over runs of consecutive frames, hop = SH*N/16
//   k_gl_frames/k_ola the same iteration for any hop: per-frame kernel + overlap-add (mel/mel.go:112-125,135)
//   k_finalize, k_add_tails   float -> double output (mel/mel.go:138)
//   k_stft_mag, k_domel_log   ToMel mel/mel.go:46-74, domel mel/impl.go:310-345, spectral_normalize :410-419
//   k_phase_stft, k_phase_istft, k_phase_ola   phase.ToPhase / FromPhase, phase/phase.go:41-155
//   k_rfft, k_irfft   test hooks: what fft.FFTReal / fft.IFFT do per frame (mel/mel.go:95,116)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gl_types.hpp"
#include "gl_fft.hpp"
#include "gl_any.hpp"
#include "gl_big.hpp"

using namespace glfft;
static_assert(glbig::kMaxN == kBigMaxN, "gl_types.hpp and gl_big.hpp agree on the longest frame");

// Forward / inverse FFT as a pair (gl_fft.hpp, PAIRED) in every kernel whose loop touches the exchange area through those two calls alone: bit 0 = no synchronisation between forward pass 3's reads and
// the inverse's first writes (same lane, same slots); bit 1 = nor between the inverse's last reads and the next frame's first writes (one row
// stride for layout A in both directions).  Bit-identical results; k_gl_fused_d -0.7 ... -0.8 % with 3 on two boxes, k_gl_fused -0.4 ... -0.7 %
// with 1 and +0.4 % with 3 (profiles/r05/ab_fp64_round5_knobs.txt).
#ifndef GL_PAIRED
#define GL_PAIRED 1      // fp32 kernels
#endif
#ifndef GL_D_PAIRED
#define GL_D_PAIRED 3    // fp64 fused and persistent kernels (the generic and any-hop kernels take bit 0 only)
#endif
// Bit 2 (round 6): the in-place pass-2 exchange (gl_fft.hpp) wherever the geometry has one butterfly per lane and pass (N 4096) -- no "partner has
// read" synchronisation left in a frame.  GL_D_INPLACE=0 builds the two-layout form (A/B).
#ifndef GL_D_INPLACE
#define GL_D_INPLACE 1
#endif
#ifndef GL_INPLACE
#define GL_INPLACE 0     // the same for the fp32 fused / persistent kernels (experiment: 8-byte slots conflict two-way in pass 3)
#endif
template <class G> constexpr int f_paired() { return GL_PAIRED | ((GL_INPLACE && G::NB1 == 1 && G::NB2 == 1) ? 4 : 0); }
#ifndef GL_D_INPLACE_ANY
#define GL_D_INPLACE_ANY 0   // ... and for the fp64 generic / any-hop kernels (pass-3 joint paired only: bit 0 | bit 2)
#endif
template <class G> constexpr int g_paired_d() { return (GL_PAIRED & 1) | ((GL_D_INPLACE_ANY && G::NB1 == 1 && G::NB2 == 1) ? 4 : 0); }
template <class G> constexpr int d_paired() { return GL_D_PAIRED | ((GL_D_INPLACE && GL_D_PAIRED == 3 && G::NB1 == 1 && G::NB2 == 1) ? 4 : 0); }

// =============================================================================================== device tables
struct DevTables {
    glany::AnyTables any;  // generic-length path (gl_any.hpp); any.M == 0 when the register cores cover Resolut
    glbig::BigTables big;  // frame lengths above 8192 (gl_big.hpp): big.Mc != 0; `any` then describes the 8192-point LDS transform
    const float* window;   // [N]
    const v2f* tw1;        // fp32 twiddles (Griffin-Lim)
    const v2f* tw2;
    const v2f* wn;
    const v2d* tw1d;       // fp64 twiddles (ToMel)
    const v2d* tw2d;
    const v2d* wnd;
    const double* windowd; // [N]
    const BinRow* undo;    // [N/2]
    const BinRow* dom;     // [K]
    unsigned* sync_err;    // host-pinned word: a two-wave stream's wait for its partner wave gave up (StreamFlagSync); null: not reported
    unsigned sync_cap;     // polls such a wait may spend (kSyncCapDefault; GOMEL_SYNC_CAP at gomel_create: tests)
};
constexpr unsigned kSyncCapDefault = 1u << 24;   // ~100 cycles per poll: about a second -- a partner wave that far behind has stopped

// Every supported Resolut in one place: expands the statement with NN a compile-time constant.
#define GOMEL_DISPATCH_N(n, ...)                                   \
    switch (n) {                                                   \
        case 4096: { constexpr int NN = 4096; __VA_ARGS__; } break; \
        case 2048: { constexpr int NN = 2048; __VA_ARGS__; } break; \
        case 1024: { constexpr int NN = 1024; __VA_ARGS__; } break; \
        case 512:  { constexpr int NN = 512;  __VA_ARGS__; } break; \
        default:   { constexpr int NN = 256;  __VA_ARGS__; } break; \
    }
GL_DEV double apply_row(const BinRow r, const double* __restrict__ e) {
    if (r.mode == 0) return e[r.lo];
    if (r.mode == 1) return e[r.lo] * (1 - r.modlo) + e[r.hi] * r.modlo;
    double tot = 0;
    for (int k = r.lo; k < r.hi; k++) tot += e[k];
    return tot / r.div;
}

// =============================================================================================== K1: log-mel -> target magnitudes
// mag[f][slot(k)] = |(undomel(exp(mel))[k][0] - TuneAdd)/TuneMul| / N for k < N/2, nyq[f] the same from channel 1 at bin
// N/2-1 (the only survivor of channel 1).  fp64 arithmetic, fp32 (or fp64) result.
// A block walks kMagFramesPerBlock consecutive frames: each thread keeps the table rows of its N/512 bins in registers
// (one table read per block, not per frame); exp() of channel 0 of all the block's frames is taken first, in full rounds of
// 256 lanes, into LDS; each row is assembled in LDS in lane-slot order (gl_fft.hpp) and leaves as one coalesced stream.
// Channel 1 feeds bin N/2-1 only, so its exp() is taken for the one or two bands that row reads, not for all K (the fp64
// exp() was what bounded this kernel).  Reads K*16 B, writes (N/2+1)*4 B per frame.
constexpr int kMagFramesPerBlock = 8;        // fewer when NumMels is large: the exp() rows of a block must fit its LDS (launch_mel_to_mag)

template <class InT> GL_DEV double apply_row_ch1(const BinRow r, const InT* __restrict__ m) {   // apply_row on exp(channel 1), straight from the pairs
    if (r.mode == 0) return exp((double)m[2 * r.lo + 1]);
    if (r.mode == 1) return exp((double)m[2 * r.lo + 1]) * (1 - r.modlo) + exp((double)m[2 * r.hi + 1]) * r.modlo;
    double tot = 0;
    for (int k = r.lo; k < r.hi; k++) tot += exp((double)m[2 * k + 1]);
    return tot / r.div;
}

// InT: element type of the log-mel pairs as the caller handed them over (double: the Go API's [][2]float64; float: the fp32
// host-I/O entry points).  The arithmetic is fp64 either way.
// mag32 / nyq32 (nullable; precision schedules): the same rows once more, rounded to float -- the fp32 kernels' magnitudes leave in the
// same pass that makes the fp64 kernels' (what a separate narrowing pass would give, bit for bit, without re-reading 16 KB per frame).
template <int N, class MagT, class InT>
__global__ __launch_bounds__(256) void k_mel_to_mag(const InT* __restrict__ mel, const BinRow* __restrict__ undo,
                                                    MagT* __restrict__ mag, MagT* __restrict__ nyq, int K, int64_t n_frames,
                                                    double tune_add, double tune_mul, double inv_n, int FPB, float* __restrict__ mag32 = nullptr,
                                                    float* __restrict__ nyq32 = nullptr) {
    constexpr int H = N / 2, PER = H < 256 ? 1 : H / 256;   // N 256: half the block's lanes idle in the bin loops
    extern __shared__ double e_sh[];                       // [FPB][K] exp(channel 0), then the staged row [H] of MagT
    MagT* row = reinterpret_cast<MagT*>(e_sh + FPB * K);
    BinRow r[PER];
    int slot[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int bin = threadIdx.x + 256 * j < H ? threadIdx.x + 256 * j : H - 1;
        r[j] = undo[bin];
        slot[j] = mag_slot_index<Geo<N>>(bin);
    }
    const bool has_bin = (int)threadIdx.x < H;             // PER > 1 implies H >= 512: every lane has its bins
    const bool unit_mul = tune_mul == 1.0;                 // kernel-argument uniform
    const int64_t f0 = (int64_t)blockIdx.x * FPB;
    const int nf = (int)(n_frames - f0 < FPB ? n_frames - f0 : FPB);
    const InT* m0 = mel + f0 * (int64_t)K * 2;
    for (int k = threadIdx.x; k < nf * K; k += 256) e_sh[k] = exp((double)m0[2 * k]);   // frame-major: e_sh[ff*K + band]
    __syncthreads();
    for (int ff = 0; ff < nf; ff++) {
        const int64_t f = f0 + ff;
        const double* e = e_sh + ff * K;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (!has_bin) break;
            // (v - TuneAdd) / TuneMul, mel/impl.go:396-397; x / 1.0 is x exactly, so the NewMel default skips the fp64 divide
            double v0 = apply_row(r[j], e) - tune_add;
            if (!unit_mul) v0 /= tune_mul;
            row[slot[j]] = (MagT)(fabs(v0) * inv_n);
            if (threadIdx.x + 256 * j == H - 1) {
                double v1 = apply_row_ch1(r[j], m0 + (int64_t)ff * K * 2) - tune_add;
                if (!unit_mul) v1 /= tune_mul;
                nyq[f] = (MagT)(fabs(v1) * inv_n);
                if (nyq32) nyq32[f] = (float)(MagT)(fabs(v1) * inv_n);
            }
        }
        __syncthreads();
        MagT* o = mag + f * H;
#pragma unroll
        for (int j = 0; j < PER; j++) if (has_bin) o[threadIdx.x + 256 * j] = row[threadIdx.x + 256 * j];
        if (mag32) {
            float* o32 = mag32 + f * H;
#pragma unroll
            for (int j = 0; j < PER; j++) if (has_bin) o32[threadIdx.x + 256 * j] = (float)row[threadIdx.x + 256 * j];
        }
        __syncthreads();                                   // the next frame overwrites the staged row
    }
}

template <class MagT, class InT>
static hipError_t launch_mel_to_mag(int N, int64_t frames, hipStream_t s, const InT* mel, const BinRow* undo, MagT* mag, MagT* nyq,
                                    int K, double tune_add, double tune_mul, float* mag32 = nullptr, float* nyq32 = nullptr) {
    const double inv_n = 1.0 / (double)N;
    // frames per block: as many exp() rows as fit the default 64 KiB of dynamic LDS next to the staged output row, at most
    // kMagFramesPerBlock; a single row that does not fit (NumMels > ~7 k) raises the kernel's limit towards the 160 KiB of the CU.
    const size_t row_bytes = (size_t)(N / 2) * sizeof(MagT), budget = 64 * 1024;
    int fpb = kMagFramesPerBlock;
    while (fpb > 1 && (size_t)fpb * K * 8 + row_bytes > budget) fpb--;
    while (fpb > 1 && (frames + fpb - 1) / fpb < 1024) fpb--;   // a short batch (one utterance) spreads over the chip instead: 28 -> 7 us at 342 frames
    const size_t shmem = (size_t)fpb * K * 8 + row_bytes;
    const unsigned blocks = (unsigned)((frames + fpb - 1) / fpb);
    hipError_t e = hipSuccess;
    GOMEL_DISPATCH_N(N, {
        if (shmem > budget) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mel_to_mag<NN, MagT, InT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e == hipSuccess) hipLaunchKernelGGL((k_mel_to_mag<NN, MagT, InT>), dim3(blocks), dim3(256), shmem, s, mel, undo, mag, nyq, K, frames, tune_add, tune_mul, inv_n, fpb, mag32, nyq32);
    });
    return e;
}

// Linear magnitudes given directly (gomel_istft = the reference's exported ISTFT, mel/mel.go:76-139, which reads only
// |spectrogram[i][j]|, j <= N/2): in[f][N/2+1] doubles -> the same lane-slot rows / Nyquist array as k_mel_to_mag, 1/N folded in.
template <int N, class MagT>
__global__ __launch_bounds__(256) void k_mag_direct(const double* __restrict__ in, MagT* __restrict__ mag, MagT* __restrict__ nyq,
                                                    double inv_n) {
    constexpr int H = N / 2;
    const int64_t f = blockIdx.x;
    const double* m = in + f * (H + 1);
    for (int k = threadIdx.x; k < H; k += 256) mag[f * H + mag_slot_index<Geo<N>>(k)] = (MagT)(fabs(m[k]) * inv_n);
    if (threadIdx.x == 0) nyq[f] = (MagT)(fabs(m[H]) * inv_n);
}

template <class MagT>
static void launch_mag_direct(int N, int64_t frames, hipStream_t s, const double* in, MagT* mag, MagT* nyq) {
    const double inv_n = 1.0 / (double)N;
    GOMEL_DISPATCH_N(N, hipLaunchKernelGGL((k_mag_direct<NN, MagT>), dim3((unsigned)frames), dim3(256), 0, s, in, mag, nyq, inv_n));
}

// The same two kernels for the generic-length path: natural bin order, fp64, NOT divided by N (gl_any.hpp keeps fft.IFFT's 1/N explicit).
template <class InT> GL_DEV double apply_row_exp(const BinRow r, const InT* __restrict__ m, int c) {   // apply_row on exp(channel c)
    if (r.mode == 0) return exp((double)m[2 * r.lo + c]);
    if (r.mode == 1) return exp((double)m[2 * r.lo + c]) * (1 - r.modlo) + exp((double)m[2 * r.hi + c]) * r.modlo;
    double tot = 0;
    for (int k = r.lo; k < r.hi; k++) tot += exp((double)m[2 * k + c]);
    return tot / r.div;
}
template <class InT>
__global__ __launch_bounds__(256) void k_mel_to_mag_nat(const InT* __restrict__ mel, const BinRow* __restrict__ undo, double* __restrict__ mag,
                                                        double* __restrict__ nyq, int K, int H, double tune_add, double tune_mul) {
    const int64_t f = blockIdx.x;
    const InT* m = mel + f * (int64_t)K * 2;
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        const BinRow r = undo[k];
        mag[f * H + k] = fabs((apply_row_exp(r, m, 0) - tune_add) / tune_mul);                      // mel/impl.go:396-397, channel 0 -> bin k
        if (k == H - 1) nyq[f] = fabs((apply_row_exp(r, m, 1) - tune_add) / tune_mul);              // channel 1's last bin lands on bin N/2
    }
}
// (odd N = 2H+1: rows of H+2 magnitudes -- bins H and H+1 both lie outside the conjugate fill of mel/mel.go:105-108 and keep their
// own: bin H goes to `mid`, bin H+1 to `nyq`, the slot of "the highest bin with a magnitude of its own")
__global__ __launch_bounds__(256) void k_mag_direct_nat(const double* __restrict__ in, double* __restrict__ mag, double* __restrict__ nyq, double* __restrict__ mid,
                                                        int H, int odd) {
    const int64_t f = blockIdx.x;
    const double* m = in + f * (H + 1 + odd);
    for (int k = threadIdx.x; k < H; k += blockDim.x) mag[f * H + k] = fabs(m[k]);
    if (threadIdx.x == 0) {
        if (odd) { mid[f] = fabs(m[H]); nyq[f] = fabs(m[H + 1]); }
        else nyq[f] = fabs(m[H]);
    }
}

// =============================================================================================== K3: initial signal
GL_DEV double splitmix_u01(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// Four samples per lane: utterances start 16-byte aligned and are padded to multiples of four samples (batch_create), so the padding
// takes the (never read) extra values and every access is a full 16 / 32 bytes.
template <class OutT, class InT>
__global__ __launch_bounds__(256) void k_init_x(const UttDesc* __restrict__ utts, int n_utt, const InT* __restrict__ x0, OutT* __restrict__ x) {
    typedef OutT out4 __attribute__((ext_vector_type(4)));
    typedef InT in4 __attribute__((ext_vector_type(4)));
    for (int iu = blockIdx.y; iu < n_utt; iu += gridDim.y) {   // grid.y is capped at 65535: batches of more utterances stride
        const UttDesc u = utts[iu];
        const int64_t quads = (u.len + 3) >> 2;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
            const int64_t n = q << 2;
            out4 o;
            if (u.has_x0) {
                const in4 a = *reinterpret_cast<const in4*>(x0 + u.xoff + n);
                o = out4{(OutT)a.x, (OutT)a.y, (OutT)a.z, (OutT)a.w};
            } else {
                o = out4{(OutT)splitmix_u01(u.seed, (uint64_t)n), (OutT)splitmix_u01(u.seed, (uint64_t)n + 1), (OutT)splitmix_u01(u.seed, (uint64_t)n + 2),
                         (OutT)splitmix_u01(u.seed, (uint64_t)n + 3)};
            }
            *reinterpret_cast<out4*>(x + u.xoff + n) = o;
        }
    }
}

template <class OutT>
__global__ __launch_bounds__(256) void k_finalize(const float* __restrict__ x, OutT* __restrict__ out, int64_t n) {   // n: a multiple of 4
    typedef OutT out4 __attribute__((ext_vector_type(4)));
    const int64_t quads = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
        const v4f a = reinterpret_cast<const v4f*>(x)[q];
        reinterpret_cast<out4*>(out)[q] = out4{(OutT)a.x, (OutT)a.y, (OutT)a.z, (OutT)a.w};
    }
}

// fp64 -> fp32 hand-over of the signal between the two halves of a precision schedule (gomel_set_precision_schedule)
__global__ __launch_bounds__(256) void k_narrow(const double* __restrict__ x, float* __restrict__ out, int64_t n) {   // n: a multiple of 4
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int64_t quads = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
        const d4 a = reinterpret_cast<const d4*>(x)[q];
        reinterpret_cast<v4f*>(out)[q] = v4f{(float)a.x, (float)a.y, (float)a.z, (float)a.w};
    }
}

// =============================================================================================== generic path: frames kernel
// WG = 128 lanes = 128/T frame streams.  Each stream: gather frame * Hann -> rFFT -> clamp -> irFFT -> * Hann -> yw[frame][N].
template <int N>
__global__ __launch_bounds__(128) void k_gl_frames(const float* __restrict__ x, const int64_t* __restrict__ frame_xoff,
                                                   const float* __restrict__ mag, const float* __restrict__ nyq,
                                                   float* __restrict__ yw, int64_t n_frames, DevTables tb) {
    typedef Geo<N> G;
    constexpr int STREAMS = 128 / G::T;
    __shared__ v2f lds[STREAMS * G::LDS_SLOTS];
    const int stream = threadIdx.x / G::T, t = threadIdx.x % G::T;
    int64_t frame = (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < n_frames;
    if (!live) frame = n_frames - 1;  // keep the lanes in the barriers; results are dropped
    v2f* my = lds + stream * G::LDS_SLOTS;
    const Twiddles<v2f> tw{tb.tw1, tb.tw2, tb.wn};
    auto sync = [] {   // single-wave streams exchange through LDS in program order: no s_barrier needed (see k_gl_fused)
        if constexpr (Geo<N>::T <= 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };

    // frames of an odd hop start on odd samples: the gather promises 4-byte alignment only (the hardware takes dword-aligned
    // dwordx2 loads; where it does not, the compiler splits them)
    typedef v2f v2f_a4 __attribute__((aligned(4)));
    const v2f_a4* xs = reinterpret_cast<const v2f_a4*>(x + frame_xoff[frame]);
    const v2f* w2 = reinterpret_cast<const v2f*>(tb.window);
    v2f v[16], w[16], wn[8];
    // the frame's samples first (needed at once), its magnitudes behind them (needed a forward FFT later): vmcnt is in issue order
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = xs[t + G::T * j];
    MagRaw mraw;
    load_mags_raw<G>(mraw, t, mag + frame * G::C, nyq + frame);
    load_wn<G>(wn, t, tb.wn);
#pragma unroll
    for (int j = 0; j < 16; j++) { w[j] = w2[t + G::T * j]; v[j] *= w[j]; }
    fft_forward<G, (GL_PAIRED & 1)>(v, my, t, tw, sync);
    MagRegs mr;
    finish_mags(mr, mraw, t == 0);
    spectral_apply<G>(v, t, mr, wn);
    fft_inverse<G, (GL_PAIRED & 1)>(v, my, t, tw, sync);
    if (live) {
        v2f* ys = reinterpret_cast<v2f*>(yw + frame * N);
#pragma unroll
        for (int j = 0; j < 16; j++) ys[t + G::T * j] = v[j] * w[j];
    }
}


// fp64 ("precise") version of the generic frames kernel: same structure, double LDS / twiddles / window / magnitudes.
#ifndef GL_FRAMES_D_WAVES
#define GL_FRAMES_D_WAVES 2   // waves per SIMD asked of the compiler: 256 registers instead of the 264 it takes unasked -- four workgroups per CU instead of two
#endif
template <int N>
__global__ __launch_bounds__(128, GL_FRAMES_D_WAVES) void k_gl_frames_d(const double* __restrict__ x, const int64_t* __restrict__ frame_xoff,
                                                     const double* __restrict__ mag, const double* __restrict__ nyq,
                                                     double* __restrict__ yw, int64_t n_frames, DevTables tb) {
    typedef Geo<N> G;
    constexpr int STREAMS = 128 / G::T;
    __shared__ v2d lds[STREAMS * G::LDS_SLOTS];
    const int stream = threadIdx.x / G::T, t = threadIdx.x % G::T;
    int64_t frame = (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < n_frames;
    if (!live) frame = n_frames - 1;
    v2d* my = lds + stream * G::LDS_SLOTS;
    const Twiddles<v2d> tw{tb.tw1d, tb.tw2d, tb.wnd};
    auto sync = [] { __syncthreads(); };
    typedef v2d v2d_a8 __attribute__((aligned(8)));   // odd hops: 8-byte alignment only
    const v2d_a8* xs = reinterpret_cast<const v2d_a8*>(x + frame_xoff[frame]);
    const v2d* w2 = reinterpret_cast<const v2d*>(tb.windowd);
    v2d v[16], wn[8];
    MagRegsD mr;
    load_mags<G>(mr, t, mag + frame * G::C, nyq + frame);
    load_wn<G>(wn, t, tb.wnd);
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = xs[t + G::T * j] * w2[t + G::T * j];
    fft_forward<G, g_paired_d<G>()>(v, my, t, tw, sync);
    spectral_apply<G>(v, t, mr, wn);
    fft_inverse<G, g_paired_d<G>()>(v, my, t, tw, sync);
    if (live) {
        v2d* ys = reinterpret_cast<v2d*>(yw + frame * N);
#pragma unroll
        for (int j = 0; j < 16; j++) ys[t + G::T * j] = v[j] * w2[t + G::T * j];
    }
}

// the same sum written straight to the caller's precision: overlap-add + mel/mel.go:138 after the persistent kernel's last iteration
template <class OutT>
__global__ __launch_bounds__(256) void k_ola_out(const UttDesc* __restrict__ utts, int n_utt, const float* __restrict__ yw, OutT* __restrict__ out, int N, int hop) {
    for (int iu = blockIdx.y; iu < n_utt; iu += gridDim.y) {
        const UttDesc u = utts[iu];
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < u.len; n += (int64_t)gridDim.x * blockDim.x) {
            int64_t k0 = n >= N ? (n - N) / hop + 1 : 0;
            int64_t k1 = n / hop;
            if (k1 > u.frames - 1) k1 = u.frames - 1;
            float acc = 0;
            for (int64_t k = k0; k <= k1; k++) acc += yw[(u.fbase + k) * N + (n - k * hop)];
            out[u.xoff + n] = (OutT)acc;
        }
    }
}

// per-frame index tables of a batch from its utterance descriptors: where the frame starts in the signal buffers, whose it is
__global__ __launch_bounds__(256) void k_frame_index(const UttDesc* __restrict__ utts, int n_utt, int64_t n_frames, int hop, int64_t* __restrict__ frame_xoff,
                                                     int32_t* __restrict__ frame_utt) {
    const int64_t fr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (fr >= n_frames) return;
    int lo = 0, hi = n_utt - 1;   // the last utterance whose first frame is <= fr
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (utts[mid].fbase <= fr) lo = mid; else hi = mid - 1;
    }
    const UttDesc u = utts[lo];
    if (frame_xoff) frame_xoff[fr] = u.xoff + (fr - u.fbase) * hop;
    if (frame_utt) frame_utt[fr] = lo;
}

// x'[n] = sum over the frames covering n of yw[k][n - k*hop], in frame order (mel/mel.go:115-125).
template <class S>
__global__ __launch_bounds__(256) void k_ola(const UttDesc* __restrict__ utts, int n_utt, const S* __restrict__ yw, S* __restrict__ xn,
                                             int N, int hop) {
    for (int iu = blockIdx.y; iu < n_utt; iu += gridDim.y) {
        const UttDesc u = utts[iu];
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < u.len; n += (int64_t)gridDim.x * blockDim.x) {
            int64_t k0 = n >= N ? (n - N) / hop + 1 : 0;
            int64_t k1 = n / hop;
            if (k1 > u.frames - 1) k1 = u.frames - 1;
            S acc = 0;
            for (int64_t k = k0; k <= k1; k++) acc += yw[(u.fbase + k) * N + (n - k * hop)];
            xn[u.xoff + n] = acc;
        }
    }
}


// =============================================================================================== fused path
// One stream of T lanes walks a SEGMENT of G consecutive frames of one utterance.  With hop = SH*(N/16) the frame's
// sample pairs sit in registers by "row" (row j = pairs t + T*j, N/16 samples), and moving to the next frame is a shift
// by SH registers: SH new rows are loaded (hop samples), SH finished rows of the overlap-add are stored (hop samples).
// So per frame-iteration the kernel reads hop*4 + (N/2+1)*4 bytes and writes hop*4 -- the algorithmic minimum.
// Segment boundaries: a segment OWNS the samples [s0, s0 + G*hop) (the last one of an utterance: to the utterance's
// end).  What its frames add beyond that (N - hop samples, the next segment's head) goes to a per-segment TAIL buffer;
// whoever reads a head sample next iteration adds x + tail (two plain loads, deterministic, no atomics).
// Split-phase synchronisation of a two-wave stream through LDS (gl_fft.hpp, "exchange synchronisation"): every wave counts its own
// events -- write burst done, read burst issued -- in a word of its own; `arrive()` publishes the count (one ds_write_b32, queued behind
// the wave's data accesses: the LDS executes a wave's instructions in order), `wait()` polls the partner's word until it shows the same
// count.  Both waves run the same event sequence, so "the partner has reached my latest event" is the one condition ever needed.
// WIDE = false (streams within one wave): program order is the synchronisation, as everywhere else.
// BOUNDED (round 6): a wait polls at most `cap` times.  Both waves of a stream run the same event sequence and are resident together, so the
// partner is never more than a few hundred cycles behind -- unless it has stopped (a fault in its half of the frame): then this wave would
// spin until the queue is torn down under it.  Instead the first wait that runs out of polls raises the handle's `err` word (host-pinned;
// the host turns it into GOMEL_ERR_HIP at its next synchronisation with the device and the call's results are discarded), sets cap = 0 --
// every later wait then gives up after ONE poll -- and the wave runs through to the end of its segment: the grid always drains.
// The poll count is wave-uniform (SGPR arithmetic inside the poll loop only: nothing on the path of a wait that finds its partner there).
#ifndef GL_FLAG_RELAXED
#define GL_FLAG_RELAXED 0   // measured on k_gl_fused_d<4096,5>: 1.4541 (relaxed) vs 1.4483 ms (volatile), n = 6 each, profiles/r06/two_frames_in_flight/ab_flag.txt: no gain, the volatile form stays
#endif
template <bool WIDE> struct StreamFlagSync {
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    unsigned* mine; unsigned* other; unsigned& ev;
    unsigned& cap; unsigned* err;
    GL_DEV void arrive() {
        if constexpr (WIDE) {
            ev++;
            asm volatile("" ::: "memory");
#if GL_FLAG_RELAXED
            // every lane writes the same value: one ds_write_b32.  A RELAXED workgroup-scope atomic store instead of a volatile one: the
            // memory legalizer puts s_waitcnt lgkmcnt(0) behind every volatile LDS access, so each "my reads are issued" signal waited for
            // those very reads to land; the LDS executes a wave's instructions in order, which is all the ordering the protocol needs
            __hip_atomic_store(mine, ev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
            *(volatile lds_u32*)mine = ev;      // every lane writes the same value: one ds_write_b32
#endif
            asm volatile("" ::: "memory");
        } else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    }
    GL_DEV void wait() {
        if constexpr (WIDE) {
            asm volatile("" ::: "memory");
#ifdef GL_SYNC_UNBOUNDED   // (A/B builds only: what the cap costs)
            while (__builtin_amdgcn_readfirstlane(*(volatile lds_u32*)other) < ev) {}
#else
            // The FIRST poll is the unbounded loop's own code -- read, wait, compare, a branch that is not taken when the partner is there -- and
            // everything about the bound sits behind that branch, out of line.  With the give-up store below a GLOBAL store the bound costs
            // nothing against the unbounded loop (1.2650 against 1.2696 ms on k_gl_fused_d<4096,5>, profiles/r06/ab_stream.txt); while it was a
            // flat store the same code cost +0.4 ... 0.8 %, a two-exit loop in front +1.5 %, assembly spellings +1.4 ... 2.0 % (ab_cap.txt).
            if (__builtin_expect(__builtin_amdgcn_readfirstlane(*(volatile lds_u32*)other) < ev, 0)) {
                unsigned polls = 0;
                bool there;
                do { there = __builtin_amdgcn_readfirstlane(*(volatile lds_u32*)other) >= ev; } while (!there && ++polls <= cap);
                if (!there) {                                 // ran out of polls: the partner wave has stopped
                    cap = 0;
                    // (every lane stores the same word: no divergent branch.  A GLOBAL store, not a flat one through the generic pointer:
                    // a flat access counts on lgkmcnt as well, and one pending at the join behind this branch -- taken or not -- made the
                    // compiler turn every counted LDS wait that follows a poll into lgkmcnt(0): that, not the loop, was the bound's cost)
                    if (err) *(volatile __attribute__((address_space(1))) unsigned*)err = 1u;
                }
            }
#endif
            asm volatile("" ::: "memory");
        }
    }
    GL_DEV void operator()() { arrive(); wait(); }
};

// k_gl_fused moves the SH new / finished rows two per dwordx4 through a DPP quad exchange from this frame length on; below it every row is its
// own dwordx2 (the exchange's ~30 vector instructions per frame cost more than the saved memory instructions: -0.6 % at N 2048, -0.8 % at 1024,
// -0.1 % at 4096 where the packing stays; bit-identical; profiles/r05/ab_fp64_round5_knobs.txt)
#ifndef GL_ROWQUAD_MIN_N
#define GL_ROWQUAD_MIN_N 4096
#endif
#ifndef GL_WIN_B128
#define GL_WIN_B128 0
#endif


// The workgroup's tables (pass-1 / pass-2 twiddles, window) from global memory into LDS: 16 bytes per lane and instruction, EVERY load
// issued before the first LDS write, so the fill costs one memory round trip (the element-wise loops it replaces went out in batches
// of five 8-byte loads, each batch waited for).  A chunk-sized launch of a host-to-host call pays this once per 12.5 frames: -1 % on
// those launches at N 4096 (profiles/r03/ab_wide_fill.txt), nothing on whole-batch launches; the small tables of N <= 1024 keep the
// loops (the wide form measured +2.5 % there).
template <bool ALIGNED16, int N1, int N2, int N3>
GL_DEV void fill_tables_lds(v2f* d1, v2f* d2, v2f* d3, const v2f* __restrict__ s1, const v2f* __restrict__ s2, const v2f* __restrict__ s3) {
    if constexpr (ALIGNED16 && N1 % 2 == 0 && N2 % 2 == 0 && N3 % 2 == 0) {
        constexpr int Q1 = N1 / 2, Q2 = N2 / 2, Q3 = N3 / 2;
        constexpr int K1 = (Q1 + kFusedThreads - 1) / kFusedThreads, K2 = (Q2 + kFusedThreads - 1) / kFusedThreads, K3 = (Q3 + kFusedThreads - 1) / kFusedThreads;
        v4f a1[K1], a2[K2], a3[K3];
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < K1; k++) a1[k] = reinterpret_cast<const v4f*>(s1)[min(t + kFusedThreads * k, Q1 - 1)];
#pragma unroll
        for (int k = 0; k < K2; k++) a2[k] = reinterpret_cast<const v4f*>(s2)[min(t + kFusedThreads * k, Q2 - 1)];
#pragma unroll
        for (int k = 0; k < K3; k++) a3[k] = reinterpret_cast<const v4f*>(s3)[min(t + kFusedThreads * k, Q3 - 1)];
#pragma unroll
        for (int k = 0; k < K1; k++) if (t + kFusedThreads * k < Q1) reinterpret_cast<v4f*>(d1)[t + kFusedThreads * k] = a1[k];
#pragma unroll
        for (int k = 0; k < K2; k++) if (t + kFusedThreads * k < Q2) reinterpret_cast<v4f*>(d2)[t + kFusedThreads * k] = a2[k];
#pragma unroll
        for (int k = 0; k < K3; k++) if (t + kFusedThreads * k < Q3) reinterpret_cast<v4f*>(d3)[t + kFusedThreads * k] = a3[k];
    } else {
        for (int i = threadIdx.x; i < N1; i += kFusedThreads) d1[i] = s1[i];
        for (int i = threadIdx.x; i < N2; i += kFusedThreads) d2[i] = s2[i];
        for (int i = threadIdx.x; i < N3; i += kFusedThreads) d3[i] = s3[i];
    }
}
// Per-geometry register budget of k_gl_fused (254-256 VGPRs everywhere): how many pass-1 twiddles stay in registers, and whether
// the next frame's rows are requested before the forward FFT (N 4096: the inverse FFT alone does not cover their HBM latency) or
// after the spectral step (shorter live range: room for the twiddles).  Chosen so that no instantiation spills (hipcc -S audit).
constexpr int fused_w1_keep(int n, int sh) { return n >= 4096 ? (sh >= 8 ? 15 : sh >= 4 ? 11 : 0) : (n >= 2048 && sh <= 1) ? 0 : 15; }
constexpr bool fused_late_rows(int n) { return n <= 2048; }
constexpr bool fused_w2_regs(int n, int sh) { return !(n >= 4096 && sh <= 2); }   // hop = N/16, N/8 at N 4096: the 14-15 row accumulator leaves no room
// One window read per frame (the synthesis multiply's registers serve the next frame's analysis multiply, k_gl_fused) costs ~30
// registers around the loop's back edge: taken where no spill results (hipcc -S audit, tests/test_isa_hazards.py).
constexpr bool fused_one_window_read(int n, int sh) { return n >= 4096 || (n == 2048 && sh != 2) || sh == 4 || sh == 8; }

// Lane-pair form of two consecutive rows.  A row is T sample pairs, one per lane (8 bytes): as such it moves with dwordx2
// instructions, 512 B each, and what bounds the vector-memory path of a CU here is the NUMBER of wave-instructions it keeps in
// flight, not their bytes (stamps: ~450 cycles of issue stall per load/store with 20 of them per frame).  Two rows r, r+1 are
// therefore moved as ONE dwordx4 per lane: even lanes carry pairs (t, t+1) of row r, odd lanes pairs (t-1, t) of row r+1, and
// one DPP exchange between quad neighbours (lane ^ 1) turns that into "row r pair t, row r+1 pair t" and back.
GL_DEV float quad_neighbour(float x) {   // the value lane ^ 1 holds (DPP quad_perm:[1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
GL_DEV void rows_from_quad(v4f q, bool even, v2f& r0, v2f& r1) {
    const v2f nlo = v2f{quad_neighbour(q.x), quad_neighbour(q.y)}, nhi = v2f{quad_neighbour(q.z), quad_neighbour(q.w)};
    r0 = even ? v2f{q.x, q.y} : nhi;
    r1 = even ? nlo : v2f{q.z, q.w};
}
GL_DEV v4f quad_from_rows(v2f r0, v2f r1, bool even) {
    const v2f n0 = v2f{quad_neighbour(r0.x), quad_neighbour(r0.y)}, n1 = v2f{quad_neighbour(r1.x), quad_neighbour(r1.y)};
    return even ? v4f{r0.x, r0.y, n0.x, n0.y} : v4f{n1.x, n1.y, r1.x, r1.y};
}

// =============================================================================================== persistent small-batch path
// k_gl_persist: ALL iterations of mel/mel.go:85-137 in ONE launch, for batches whose frames fit the chip at once (a single
// utterance: 342 frames at the ToWavPng geometry).  One stream per frame as in k_gl_frames, but the stream keeps its frame for
// the whole call: window, magnitudes and twiddle operands are loaded once, and per iteration the only memory traffic is the
// frame's windowed output y (16-B write-through stores) and, next iteration, the gather of the overlap-add
// x[n] = sum_k yw[k][n - k hop] from the up to 2*ceil(N/hop)-1 neighbouring frames -- in ascending frame order, so the sums are
// the ones k_ola makes and the result is bit-identical to the k_gl_frames + k_ola pair.
//
// The exchange between frames is the critical path (a frame cannot start iteration i+1 before its neighbours finished i), so it
// carries no flag, no drain-then-signal and no fence: THE DATA IS THE SIGNAL (cdna_hip_programming.md Guideline 16, form R2,
// with a sentinel instead of a tag).  y lives in THREE buffers, iteration i writes buf[i % 3].  A slot a reader may look at holds
// either the sentinel (all-ones words, a NaN no arithmetic produces) or the fresh value, never an older y:
//   * stream P, iteration i, right after its gather: stores the sentinel over its own slot of buf[(i+1) % 3].  That slot held
//     y_P(i-2); its readers were the neighbours' gathers of iteration i-1, which are over, because P has just consumed what those
//     neighbours produced AFTER them (their y(i-1)).
//   * P then computes (two FFTs), executes s_waitcnt vmcnt(0) -- free by then, and it orders the sentinel before everything
//     that follows -- and stores y_P(i) into buf[i % 3] (which holds the sentinel P wrote one iteration earlier).
//   * a reader in iteration i+1 loads the quads it needs from buf[i % 3] with sc1 loads (agent-coherent: they do not hit a stale
//     L1 line) and re-loads those that still show the sentinel in ANY of their four words, so a torn 16-byte read cannot pass.
//     It cannot see y_P(i-3) there: it has consumed y_P(i-1) already, and P drained the sentinel that replaced y_P(i-3) before
//     storing y_P(i-1).
// The host fills buf[0] with the sentinel before every launch (what a previous launch left in buf[1] / buf[2] is covered by the
// same argument: P overwrites its slots there with the sentinel in iterations 0 and 1 and drains before storing y_P(0) / y_P(1),
// which every reader of those slots consumes first).  Every stream must be resident (the host checks the
// occupancy census before choosing this path); the re-load loop is bounded by a wall-clock limit, a timeout -- neighbours not
// resident, or data that really contains the sentinel pattern -- sets *err, after which every stream runs through without
// waiting, and the host recomputes the batch with the per-iteration kernels.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr int kPersistAutoNb = 7;            // ... and up to which the library chooses it by itself (hop >= N/8)
constexpr unsigned kPersistSpinTicks = 20u * 100000u;   // default spin limit: 20 ms of the 100 MHz wall clock
constexpr unsigned kPersistSentinel = 0xffffffffu;

GL_DEV bool persist_fresh(v4u q) { return q.x != kPersistSentinel && q.y != kPersistSentinel && q.z != kPersistSentinel && q.w != kPersistSentinel; }

#ifdef GOMEL_STAMPS   // diagnostic build only (scripts/stamp_probe.py, stamp_persist.py): per-phase cycle sums, all waves
__device__ unsigned long long g_stamp_acc[21];
#endif
// OCC = waves per SIMD the kernel is built for: 1 = 512 VGPRs (twiddles, window, spectral operands and four covering frames' quads
// in registers), 2 = twice the resident streams (256 VGPRs: tables re-read through the L1, which nothing invalidates here; two
// covering frames in flight).  DEPTH = covering frames in flight per row pair during the gather (each group of DEPTH is one round
// trip to memory, ~2.6 us): 4 at OCC 1, 2 at OCC 2.  (8, for hops below N/4, was measured SLOWER than two rounds of 4 -- 0.83 vs
// 0.76 ms at N 2048 / hop 256 -- because its 256 registers of quads leave none for the tables; not instantiated.)
template <int N, int OCC, int DEPTH>
__global__ __launch_bounds__(128, OCC) void k_gl_persist(const float* __restrict__ x0, const UttDesc* __restrict__ utts, const int32_t* __restrict__ frame_utt,
                                                      const float* __restrict__ mag, const float* __restrict__ nyq, float* yw,
                                                      unsigned int* err, int64_t n_frames, int iters, int hop, unsigned spin_ticks, int retry_nap, DevTables tb,
                                                      int64_t frame0, int64_t frame_end) {
    // frame0 ... frame_end: the frames THIS launch iterates (whole utterances; a batch of up to two chip-fulls runs as two launches, round 4);
    // offsets stay those of the whole batch
    typedef Geo<N> G;
    constexpr int T = G::T, STREAMS = 128 / T;
    // LEAN: the registers go to the gather (OCC 2: there are only 256; DEPTH 8: 64 quads in flight) -- twiddles, window, spectral
    // operands and the frame's own previous output are re-read (L1 / the exchange buffers) instead of kept
    constexpr bool LEAN = OCC != 1 || DEPTH == 8;
    __shared__ v2f lds[STREAMS * G::LDS_SLOTS];
    const int stream = threadIdx.x / T, t = threadIdx.x % T;
    int64_t frame = frame0 + (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < frame_end;
    if (!live) frame = frame_end - 1;  // keeps the lanes in the barriers; nothing of it is stored
    v2f* my = lds + stream * G::LDS_SLOTS;
    // every twiddle operand of this lane in registers for the whole call (the launch runs one wave per SIMD: 512 VGPRs)
    v2f w2r[16], w1r[16];
    if constexpr (!LEAN) {
#pragma unroll
        for (int uu = 0; uu < G::NB2; uu++)
#pragma unroll
            for (int q = 0; q < G::R2; q++) w2r[uu * G::R2 + q] = tb.tw2[q * 8 + (t + T * uu) / G::R1];
#pragma unroll
        for (int uu = 0; uu < G::NB1; uu++)
#pragma unroll
            for (int q = 0; q < G::R1; q++) w1r[uu + G::NB1 * q] = tb.tw1[q * G::C1 + t + T * uu];
    }
    const Twiddles<v2f> tw{tb.tw1, tb.tw2, nullptr, !LEAN ? w1r : nullptr, !LEAN ? 15 : 0, !LEAN ? w2r : nullptr};
    static_assert(DEPTH == 2 || DEPTH == 4 || DEPTH == 8, "gather depth");
    typedef unsigned long long mask_t;   // one bit per quad in flight: 8 row pairs x DEPTH
    auto sync = [] {
        if constexpr (Geo<N>::T <= 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    const UttDesc u = utts[frame_utt[frame]];
    const int f = (int)(frame - u.fbase), F = u.frames;
    const int nb = (N - 1) / hop;
    // this frame's constants: window pairs, spectral-step operands, magnitudes
    const v2f* w2 = reinterpret_cast<const v2f*>(tb.window);
    v2f w[16], wn[8];
    if constexpr (!LEAN) {
#pragma unroll
        for (int j = 0; j < 16; j++) w[j] = w2[t + T * j];
    }
    MagRaw mraw;
    MagRegs mr;
    if constexpr (!LEAN) {
        load_mags_raw<G>(mraw, t, mag + frame * G::C, nyq + frame);
        load_wn<G>(wn, t, tb.wn);
        finish_mags(mr, mraw, t == 0);
    }
    // gather geometry.  Rows 2jp, 2jp+1 move as one 16-byte quad per lane (quad_from_rows): lane t holds the four samples from
    // s = 2 (T (2jp + (t&1)) + (t & ~1)) of the frame.  With hop and N multiples of 4 all four have the same covering frames
    // k0 .. k0 + cnt - 1, which depend on the frame only: computed once.
    const bool even = !(t & 1);
    const int lane_s = 2 * (T * (t & 1) + (t & ~1));
    int geo[8];   // k0 | cnt << 20 | own << 25, own = where this frame itself sits among the covering frames (frames per utterance < 2^20: host check)
#pragma unroll
    for (int jp = 0; jp < 8; jp++) {
        const unsigned n = (unsigned)f * (unsigned)hop + (unsigned)(4 * T * jp + lane_s);   // < 2^31: the host admits short utterances only
        const unsigned a = n >= (unsigned)N ? (n - (unsigned)N) / (unsigned)hop + 1u : 0u;
        unsigned b = n / (unsigned)hop;
        if (b > (unsigned)(F - 1)) b = (unsigned)(F - 1);
        geo[jp] = (int)(a | (b - a + 1u) << 20 | ((unsigned)f - a) << 25);
    }
    // yw = the three buffers back to back, [3][n_frames][N] floats (< 2^32 bytes: host check); one descriptor, the buffer chosen by
    // the scalar offset operand
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(yw, 0, 0xffffffff, 0x00020000);
    const unsigned buf_bytes = (unsigned)(n_frames * N) * 4u;
    const unsigned my_off = (unsigned)(frame * N + lane_s) * 4u;   // byte offset of this lane's quad of row pair 0 (a buffer holds < 2^28 floats: host check)
    // byte offset of frame k0's copy of this lane's quad of row pair 0; + 4 T jp floats per row pair, + (N - hop) floats per covering frame
    const unsigned base_off = (unsigned)(u.fbase * N + (int64_t)f * hop + lane_s) * 4u;
    const unsigned kstep = (unsigned)(N - hop) * 4u;
    const v4u sentinel = v4u{kPersistSentinel, kPersistSentinel, kPersistSentinel, kPersistSentinel};
    bool dead = false;   // a wait timed out somewhere (here or, seen through *err, elsewhere): no more waiting, the host recomputes
#ifdef GOMEL_STAMPS
    Stamps st; st.start();
#else
    NoStamps st;
#endif
    v4f ownq[8];   // this frame's own windowed output of the previous iteration: its term of the overlap-add needs no load
    int cur = 0, prev = 2, next = 1;   // it % 3, (it - 1) % 3, (it + 1) % 3
    for (int it = 0; it < iters; it++) {
        v2f v[16];
        // OCC 2: the table addresses of an iteration (window, twiddles, magnitudes: some 40 64-bit pointers) are loop invariants the
        // compiler would hoist and then spill; an opaque copy of the lane index makes it rebuild them where they are used
        int tl = t;
        if constexpr (LEAN) asm volatile("" : "+v"(tl));
        if (it == 0) {
            typedef v2f v2f_a4 __attribute__((aligned(4)));
            const v2f_a4* xs = reinterpret_cast<const v2f_a4*>(x0 + u.xoff + (int64_t)f * hop);
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = xs[t + T * j];
        } else {
            st.mark(7);
            v4f acc[8];
#pragma unroll
            for (int jp = 0; jp < 8; jp++) acc[jp] = v4f{0.f, 0.f, 0.f, 0.f};
            for (int dk = 0; dk <= nb; dk += DEPTH) {   // cnt <= nb + 1; DEPTH frames' copies in flight per row pair (up to 32 x 16 B per lane)
                v4u a[8][DEPTH];
                mask_t todo = 0;   // bit DEPTH jp + q: the quad is a neighbour's and not in yet
#pragma unroll
                for (int jp = 0; jp < 8; jp++)
#pragma unroll
                    for (int q = 0; q < DEPTH; q++) {
                        a[jp][q] = v4u{0u, 0u, 0u, 0u};
                        if (dk + q < (geo[jp] >> 20 & 31) && (LEAN || dk + q != (geo[jp] >> 25 & 31))) todo |= (mask_t)1 << (DEPTH * jp + q);
                    }
                if (dead) todo = 0;
                unsigned long long t0 = 0;
                for (unsigned spins = 0;; spins++) {
#pragma unroll
                    for (int jp = 0; jp < 8; jp++)
#pragma unroll
                        for (int q = 0; q < DEPTH; q++)
                            if (todo & ((mask_t)1 << (DEPTH * jp + q)))
                                a[jp][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + (unsigned)(4 * T * jp) * 4u + (unsigned)((geo[jp] & 0xfffff) + dk + q) * kstep, (unsigned)prev * buf_bytes, 16);   // aux 16 = sc1
#pragma unroll
                    for (int jp = 0; jp < 8; jp++)
#pragma unroll
                        for (int q = 0; q < DEPTH; q++)
                            if ((todo & ((mask_t)1 << (DEPTH * jp + q))) && persist_fresh(a[jp][q])) todo &= ~((mask_t)1 << (DEPTH * jp + q));
#ifdef GOMEL_STAMPS
                    st.acc[2] += 1;   // load rounds
#endif
                    if (!__any(todo != 0)) break;
                    // some neighbour has not stored yet: look again, but not for ever
                    // (the re-load goes out at once: a look at *err costs a round trip of its own, so only every 16th time)
                    if (t0 == 0) t0 = wall_clock64();
                    const bool other = (spins & 15u) == 15u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                    if (other || wall_clock64() - t0 > spin_ticks) {
                        if (!other && (threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dead = true;
                        break;
                    }
                    for (int d = 0; d < retry_nap; d++) __builtin_amdgcn_s_sleep(1);   // re-loads at full rate slow the stores they are waiting for
                }
#pragma unroll
                for (int jp = 0; jp < 8; jp++)
#pragma unroll
                    for (int q = 0; q < DEPTH; q++) acc[jp] += (!LEAN && dk + q == (geo[jp] >> 25 & 31)) ? ownq[jp] : __builtin_bit_cast(v4f, a[jp][q]);   // absent frames add the zeros a[][] started as
            }
            st.mark(0);
#ifdef GOMEL_STAMPS_RELOAD   // diagnostic: the same loads once more, nobody writing: the bare round trip
            {
                v4u r[8][DEPTH];
#pragma unroll
                for (int jp = 0; jp < 8; jp++)
#pragma unroll
                    for (int q = 0; q < DEPTH; q++)
                        if (jp < GOMEL_STAMPS_RELOAD && (threadIdx.x < GOMEL_STAMPS_RELOAD_T) && q < (geo[jp] >> 20 & 31) && q != (geo[jp] >> 25 & 31)) r[jp][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + (unsigned)(4 * T * jp) * 4u + (unsigned)((geo[jp] & 0xfffff) + q) * kstep, (unsigned)prev * buf_bytes, 16);
                        else r[jp][q] = v4u{0u, 0u, 0u, 0u};
                unsigned x_ = 0;
#pragma unroll
                for (int jp = 0; jp < 8; jp++)
#pragma unroll
                    for (int q = 0; q < DEPTH; q++) x_ |= r[jp][q].x;
                asm volatile("" :: "v"(x_));
                st.mark(7);
            }
#endif
#pragma unroll
            for (int jp = 0; jp < 8; jp++) rows_from_quad(acc[jp], even, v[2 * jp], v[2 * jp + 1]);
        }
        // the slot iteration it + 1 will fill: nobody reads y(it - 2) any more (see above).  Early, so that the drain below is free.
        if (live) {
#pragma unroll
            for (int jp = 0; jp < 8; jp++) __builtin_amdgcn_raw_buffer_store_b128(sentinel, rs, my_off + (unsigned)(4 * T * jp) * 4u, (unsigned)next * buf_bytes, 16);
        }
        st.mark(1);
        if constexpr (LEAN) {
#pragma unroll
            for (int j = 0; j < 16; j++) w[j] = w2[tl + T * j];
        }
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] *= w[j];
        if constexpr (LEAN) {   // the spectral step's operands: issued here, needed a forward FFT later
            load_mags_raw<G>(mraw, tl, mag + frame * G::C, nyq + frame);
            load_wn<G>(wn, tl, tb.wn);
        }
        fft_forward<G, f_paired<G>()>(v, my, tl, tw, sync);
        st.mark(3);
        if constexpr (LEAN) finish_mags(mr, mraw, t == 0);
        spectral_apply<G>(v, t, mr, wn);
        st.mark(4);
        if constexpr (LEAN) asm volatile("" : "+v"(tl));
        fft_inverse<G, f_paired<G>()>(v, my, tl, tw, sync);
        st.mark(5);
        if constexpr (LEAN) {
            asm volatile("" : "+v"(tl));
#pragma unroll
            for (int j = 0; j < 16; j++) w[j] = w2[tl + T * j];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sentinel stores above are through: fresh data can never be overtaken by them
#pragma unroll
        for (int jp = 0; jp < 8; jp++) {
            const v4f q = quad_from_rows(v[2 * jp] * w[2 * jp], v[2 * jp + 1] * w[2 * jp + 1], even);
            if constexpr (!LEAN) ownq[jp] = q;
            if (live) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, q), rs, my_off + (unsigned)(4 * T * jp) * 4u, (unsigned)cur * buf_bytes, 16);   // aux 16 = sc1
        }
        st.mark(6);
        prev = cur; cur = next; next = next == 2 ? 0 : next + 1;
    }
#ifdef GOMEL_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int q = 0; q < 8; q++) atomicAdd(&g_stamp_acc[q], st.acc[q]);
        atomicAdd(&g_stamp_acc[20], (unsigned long long)iters);
    }
#endif
}

// The same persistent kernel in fp64 (64-bit mode, the mode that agrees with the reference's arithmetic on every input): one stream
// per frame, three exchange buffers of doubles, the same sentinel protocol.  A lane's unit of exchange is one pair of doubles
// (16 bytes: row j, pair t), so no quad packing; the sums run in ascending frame order, i.e. are k_ola<double>'s, and the result
// equals the fp64 generic pair's bit for bit.  Registers go to the FFT (fp64 temporaries) and to the gather (8 rows x 4 covering
// frames of pairs in flight, two passes over the rows): tables, magnitudes and the frame's own previous output are re-read.
// Needs hop even (a pair never straddles a hop) and at most kPersistMaxNb neighbours on each side.
typedef unsigned long long v2ul __attribute__((ext_vector_type(2)));
// A double is "not stored yet" when its HIGH word is all ones (a NaN no arithmetic produces: results carry 0x7ff8.../0xfff8...).  The low
// word cannot serve as well: 2^-32 of all finite doubles end in 32 one-bits, which over the 84 M values of one utterance would stall
// a call in fifty.  So this relies on an aligned 8-byte half of a 16-byte store never being torn (observed for 8-byte granules,
// MI355X_MICROARCH.md visibility section; a store's 16 bytes lie in one 128-byte line).
GL_DEV bool persist_fresh_d(v4u q) { return q.y != kPersistSentinel && q.w != kPersistSentinel; }

constexpr bool persist_d_tw1_in_lds(int n) { return n <= 2048; }
template <int N> static constexpr size_t persist_d_lds_bytes() {
    typedef Geo<N> G;
    return ((size_t)(128 / G::T) * G::LDS_SLOTS + G::C + G::TW2_N + (persist_d_tw1_in_lds(N) ? G::TW1_N : 0)) * 16;
}
template <int N>
__global__ __launch_bounds__(128, 1) void k_gl_persist_d(const double* __restrict__ x0, const UttDesc* __restrict__ utts, const int32_t* __restrict__ frame_utt,
                                                        const double* __restrict__ mag, const double* __restrict__ nyq, double* yw,
                                                        unsigned int* err, int64_t n_frames, int iters, int hop, unsigned spin_ticks, int retry_nap, DevTables tb,
                                                        int64_t frame0, int64_t frame_end) {
    // frame0 ... frame_end: the frames THIS launch iterates -- whole utterances (frames of an utterance exchange among themselves only); a batch
    // of up to twice the chip's capacity runs as two such launches one after the other (round 4).  Offsets stay those of the whole batch.
    typedef Geo<N> G;
    constexpr int T = G::T, STREAMS = 128 / T, DEPTH = 4;
    // LDS: the streams' exchange areas, then (round 4) the window and the pass-2 twiddles, which every iteration used to fetch from
    // global memory anew -- 62 of its ~117 table loads; the pass-1 twiddles join them where two workgroups per CU still fit (N <= 2048)
    extern __shared__ __attribute__((aligned(16))) char persist_d_smem[];
    v2d* lds = reinterpret_cast<v2d*>(persist_d_smem);
    v2d* win_l = lds + STREAMS * G::LDS_SLOTS;
    v2d* tw2_l = win_l + G::C;
    v2d* tw1_l = tw2_l + G::TW2_N;
    constexpr bool TW1_LDS = persist_d_tw1_in_lds(N);
    for (int i = threadIdx.x; i < G::C; i += 128) win_l[i] = reinterpret_cast<const v2d*>(tb.windowd)[i];
    for (int i = threadIdx.x; i < G::TW2_N; i += 128) tw2_l[i] = tb.tw2d[i];
    if constexpr (TW1_LDS) { for (int i = threadIdx.x; i < G::TW1_N; i += 128) tw1_l[i] = tb.tw1d[i]; }
    __syncthreads();
    const int stream = threadIdx.x / T, t = threadIdx.x % T;
    int64_t frame = frame0 + (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < frame_end;
    if (!live) frame = frame_end - 1;
    v2d* my = lds + stream * G::LDS_SLOTS;
    // N 4096 (pass-1 twiddles not in LDS): a lane's fifteen pass-1 twiddles are the same in every iteration -- this workgroup keeps its
    // frame -- so they are fetched once, not 2 x 15 times per iteration from L2 at the head of the forward and the tail of the inverse FFT
    v2d w1c[16];
    if constexpr (!TW1_LDS) {
#pragma unroll
        for (int uu = 0; uu < G::NB1; uu++)
#pragma unroll
            for (int q = 1; q < G::R1; q++) w1c[uu + G::NB1 * q] = tb.tw1d[q * G::C1 + (int)(threadIdx.x % T) + T * uu];
    }
    const Twiddles<v2d> tw{TW1_LDS ? (const v2d*)tw1_l : tb.tw1d, tw2_l, tb.wnd, TW1_LDS ? nullptr : w1c, TW1_LDS ? 0 : G::R1 - 1, nullptr};
    auto sync = [] { __syncthreads(); };
    const UttDesc u = utts[frame_utt[frame]];
    const int f = (int)(frame - u.fbase), F = u.frames;
    const int nb = (N - 1) / hop;
    const v2d* w2 = win_l;
    int geo[16];   // per row j (pair t + T j, samples 2 (t + T j), +1): k0 | cnt << 20 (covering frames k0 .. k0 + cnt - 1)
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const unsigned n = (unsigned)f * (unsigned)hop + 2u * (unsigned)(t + T * j);
        const unsigned a = n >= (unsigned)N ? (n - (unsigned)N) / (unsigned)hop + 1u : 0u;
        unsigned b = n / (unsigned)hop;
        if (b > (unsigned)(F - 1)) b = (unsigned)(F - 1);
        geo[j] = (int)(a | (b - a + 1u) << 20);
    }
    // yw = three buffers back to back, [3][n_frames][N] doubles (< 2^32 bytes: host check)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(yw, 0, 0xffffffff, 0x00020000);
    const unsigned buf_bytes = (unsigned)(n_frames * N) * 8u;
    const unsigned my_off = (unsigned)(frame * N + 2 * t) * 8u;                                  // + 2 T j doubles per row
    const unsigned base_off = (unsigned)(u.fbase * N + (int64_t)f * hop + 2 * t) * 8u;           // frame k0's copy of row 0's pair; + (N - hop) doubles per covering frame
    const unsigned kstep = (unsigned)(N - hop) * 8u;
    const v4u sentinel = v4u{kPersistSentinel, kPersistSentinel, kPersistSentinel, kPersistSentinel};
    bool dead = false;
    int cur = 0, prev = 2, next = 1;
    for (int it = 0; it < iters; it++) {
        v2d v[16];
        int tl = t;   // opaque copy of the lane index: keeps the compiler from hoisting (and spilling) this iteration's table addresses
        asm volatile("" : "+v"(tl));
        if (it == 0) {
            typedef v2d v2d_a8 __attribute__((aligned(8)));
            const v2d_a8* xs = reinterpret_cast<const v2d_a8*>(x0 + u.xoff + (int64_t)f * hop);
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = xs[t + T * j];
        } else {
#pragma unroll
            for (int half = 0; half < 2; half++) {   // rows 0..7, then 8..15: 32 pairs of 16 B in flight per lane
                v2d acc[8];
#pragma unroll
                for (int jj = 0; jj < 8; jj++) acc[jj] = v2d{0.0, 0.0};
                for (int dk = 0; dk <= nb; dk += DEPTH) {
                    v4u a[8][DEPTH];
                    unsigned todo = 0;
#pragma unroll
                    for (int jj = 0; jj < 8; jj++)
#pragma unroll
                        for (int q = 0; q < DEPTH; q++) {
                            a[jj][q] = v4u{0u, 0u, 0u, 0u};
                            if (dk + q < (geo[8 * half + jj] >> 20 & 31)) todo |= 1u << (DEPTH * jj + q);
                        }
                    if (dead) todo = 0;
                    unsigned long long t0 = 0;
                    for (unsigned spins = 0;; spins++) {
#pragma unroll
                        for (int jj = 0; jj < 8; jj++)
#pragma unroll
                            for (int q = 0; q < DEPTH; q++)
                                if (todo & (1u << (DEPTH * jj + q)))
                                    a[jj][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + (unsigned)(2 * T * (8 * half + jj)) * 8u + (unsigned)((geo[8 * half + jj] & 0xfffff) + dk + q) * kstep,
                                                                                     (unsigned)prev * buf_bytes, 16);   // aux 16 = sc1
#pragma unroll
                        for (int jj = 0; jj < 8; jj++)
#pragma unroll
                            for (int q = 0; q < DEPTH; q++)
                                if ((todo & (1u << (DEPTH * jj + q))) && persist_fresh_d(a[jj][q])) todo &= ~(1u << (DEPTH * jj + q));
                        if (!__any(todo != 0u)) break;
                        if (t0 == 0) t0 = wall_clock64();
                        const bool other = (spins & 15u) == 15u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                        if (other || wall_clock64() - t0 > spin_ticks) {
                            if (!other && (threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            dead = true;
                            break;
                        }
                        for (int d = 0; d < retry_nap; d++) __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int jj = 0; jj < 8; jj++)
#pragma unroll
                        for (int q = 0; q < DEPTH; q++) acc[jj] += __builtin_bit_cast(v2d, a[jj][q]);   // absent frames add the zeros a[][] started as
                }
#pragma unroll
                for (int jj = 0; jj < 8; jj++) v[8 * half + jj] = acc[jj];
            }
        }
        if (live) {   // the slot iteration it + 1 will fill (k_gl_persist's argument)
#pragma unroll
            for (int j = 0; j < 16; j++) __builtin_amdgcn_raw_buffer_store_b128(sentinel, rs, my_off + (unsigned)(2 * T * j) * 8u, (unsigned)next * buf_bytes, 16);
        }
        MagRegsD mr;
        v2d wn[8];
        load_mags<G>(mr, tl, mag + frame * G::C, nyq + frame);
        load_wn<G>(wn, tl, tb.wnd);
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] *= w2[tl + T * j];
        fft_forward<G, d_paired<G>()>(v, my, tl, tw, sync);
        spectral_apply<G>(v, t, mr, wn);
        asm volatile("" : "+v"(tl));
        fft_inverse<G, d_paired<G>()>(v, my, tl, tw, sync);
        asm volatile("" : "+v"(tl));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sentinel stores are through before any fresh pair goes out
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const v2d y = v[j] * w2[tl + T * j];
            if (live) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, y), rs, my_off + (unsigned)(2 * T * j) * 8u, (unsigned)cur * buf_bytes, 16);
        }
        prev = cur; cur = next; next = next == 2 ? 0 : next + 1;
    }
}

template <int N, int SH>
__global__ __launch_bounds__(kFusedThreads, 2) void k_gl_fused(const float* __restrict__ x_in, const float* __restrict__ tail_in,
                                                     float* __restrict__ x_out, float* __restrict__ tail_out,
                                                     const SegDesc* __restrict__ segs, int n_segs, int g_max,
                                                     const float* __restrict__ mag, const float* __restrict__ nyq, DevTables tb) {
    typedef Geo<N> G;
    constexpr int T = G::T, STREAMS = kFusedThreads / T, KEEP = 16 - SH, TAIL_PAIRS = KEEP * T;  // tail = (N - hop)/2 pairs
    // LDS: per-stream exchange areas, then the tables every stream of the WG shares (TW1, TW2, window as pairs)
    __shared__ __attribute__((aligned(16))) v2f lds[STREAMS * G::LDS_SLOTS + G::TW1_N + G::TW2_N + G::C];
    v2f* tw1_l = lds + STREAMS * G::LDS_SLOTS;
    v2f* tw2_l = tw1_l + G::TW1_N;
    v2f* win_l = tw2_l + G::TW2_N;
    // WPAIR (experiment GL_WIN_B128): the window table in LDS as ROW PAIRS -- entry jp * T + t = {w[2jp][t], w[2jp+1][t]}, 16 bytes -- so that the
    // one-read-per-frame form fetches its 16 window pairs with 8 ds_read_b128 instead of 16 ds_read_b64 (same bytes, half the LDS instructions)
    constexpr bool WPAIR = GL_WIN_B128 && fused_one_window_read(N, SH) && (STREAMS * G::LDS_SLOTS + G::TW1_N + G::TW2_N) % 2 == 0;
    if constexpr (WPAIR) {
        for (int i = threadIdx.x; i < G::TW1_N; i += kFusedThreads) tw1_l[i] = tb.tw1[i];
        for (int i = threadIdx.x; i < G::TW2_N; i += kFusedThreads) tw2_l[i] = tb.tw2[i];
        const v2f* wsrc = reinterpret_cast<const v2f*>(tb.window);
        for (int i = threadIdx.x; i < G::C / 2; i += kFusedThreads) {
            const int jp = i / T, tt = i % T;
            const v2f a = wsrc[T * (2 * jp) + tt], b = wsrc[T * (2 * jp + 1) + tt];
            reinterpret_cast<v4f*>(win_l)[i] = v4f{a.x, a.y, b.x, b.y};
        }
    } else
    fill_tables_lds<(STREAMS * G::LDS_SLOTS) % 2 == 0 && N >= 2048, G::TW1_N, G::TW2_N, G::C>(tw1_l, tw2_l, win_l, tb.tw1, tb.tw2, reinterpret_cast<const v2f*>(tb.window));
    __syncthreads();
    // this lane's pass-2 twiddles depend on the lane only: registers for the whole segment instead of 30 LDS reads per frame
    // (measured -3 % at N 4096, -2 % at N 1024: the kernel is power-limited and LDS table bytes are part of the bill; 254 VGPRs)
    v2f w2r[16];
#pragma unroll
    for (int u = 0; u < G::NB2; u++)
#pragma unroll
        for (int q = 0; q < G::R2; q++) w2r[u * G::R2 + q] = tb.tw2[q * 8 + ((int)(threadIdx.x % G::T) + G::T * u) / G::R1];
    // ... and as many of its pass-1 twiddles as the register file takes without spilling (W1_KEEP of the R1-1, by geometry:
    // the pass-1 peak already holds them as temporaries, so keeping them costs registers only outside that peak): -1.5 ... -2 %
    constexpr int W1_KEEP = fused_w1_keep(N, SH);
    v2f w1r[16];
#pragma unroll
    for (int u = 0; u < G::NB1; u++)
#pragma unroll
        for (int q = 0; q < G::R1; q++) w1r[u + G::NB1 * q] = q <= W1_KEEP ? tb.tw1[q * G::C1 + (int)(threadIdx.x % G::T) + G::T * u] : v2f{0.f, 0.f};
    const Twiddles<v2f> tw{tw1_l, tw2_l, nullptr, W1_KEEP ? w1r : nullptr, W1_KEEP, fused_w2_regs(N, SH) ? w2r : nullptr};
        // Exchange synchronisation.  A stream of T <= 64 lanes lives inside ONE wave, whose LDS operations execute in program
    // order: its exchanges need no s_barrier, only a compiler-level fence so that loads/stores are not reordered.  Wider
    // streams (N = 4096: two waves) use the workgroup barrier.
    const int stream = threadIdx.x / T, t = threadIdx.x % T;
    // (StreamFlagSync -- the fp64 kernel's per-stream synchronisation through LDS event counters -- was measured here too: bit-identical,
    // 0.587 -> 0.589 ms; with two waves per SIMD the other workgroup already fills what a barrier leaves idle.  s_barrier stays.)
    auto sync = [] {
        if constexpr (Geo<N>::T <= 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    constexpr bool COUPLED = STREAMS > 1;

    int seg = blockIdx.x * STREAMS + stream;
    const bool live = seg < n_segs;
    if (!live) seg = n_segs - 1;
    SegDesc d = segs[seg];
    v2f* my = lds + stream * G::LDS_SLOTS;
    if constexpr (T >= 64) {  // whole-wave streams: the descriptor is wave-uniform -> keep it (and all row logic) in SGPRs
        d.G = __builtin_amdgcn_readfirstlane(d.G);
        d.has_prev = __builtin_amdgcn_readfirstlane(d.has_prev);
        d.has_next = __builtin_amdgcn_readfirstlane(d.has_next);
        seg = __builtin_amdgcn_readfirstlane(seg);
    }

    // Bases = kernel-argument pointer + offset.  Streams of T >= 64 lanes are whole waves, so the segment (hence every
    // offset) is wave-uniform and is moved to SGPRs; lane offsets are then 32-bit.  (T = 32: two streams share a wave.)
    constexpr bool WU = (T >= 64);
    auto uo = [](long long v) { if constexpr (WU) return uniform_i64(v); else return v; };
    const int seg_prev = d.has_prev ? seg - 1 : seg;                          // clamped: only read with flag f_prev
    const v2f* xin_p = reinterpret_cast<const v2f*>(x_in) + uo(d.s0 / 2);     // row r, lane t -> xin_p[T*r + t]
    v2f* xout_p = reinterpret_cast<v2f*>(x_out) + uo(d.s0 / 2);
    const v2f* tprev = reinterpret_cast<const v2f*>(tail_in) + uo((int64_t)seg_prev * TAIL_PAIRS);
    const v2f* town = reinterpret_cast<const v2f*>(tail_in) + uo((int64_t)seg * TAIL_PAIRS);
    v2f* tout = reinterpret_cast<v2f*>(tail_out) + uo((int64_t)seg * TAIL_PAIRS);
    const float* mag_p = mag + uo(d.frame0 * G::C);
    const float* nyq_p = nyq + uo(d.frame0);
    const float f_prev = d.has_prev ? 1.f : 0.f, f_next = d.has_next ? 1.f : 0.f;
    const int rows_owned = d.G * SH;  // rows >= this belong to the next segment's head
    if constexpr (COUPLED) {  // streams of one WG share the barriers: loop to the longest segment among them
        __shared__ int g_sh[STREAMS];
        if (t == 0) g_sh[stream] = live ? d.G : 0;
        __syncthreads();
        int m = 0;
#pragma unroll
        for (int i = 0; i < STREAMS; i++) m = max(m, g_sh[i]);
        g_max = m;
    } else {
        g_max = d.G;
    }

#ifdef GOMEL_STAMPS
    Stamps st; st.start();
#else
    NoStamps st;
#endif
    v2f xin[16], acc[KEEP];
    const v2f* wl = win_l + t;  // window pair of row j: wl[T*j]
    // the eight W_N^k of this lane's bin pairs: frame-independent, computed once per segment from one table value and kept
    // (16 registers; rebuilding them every frame was 16 packed multiplies per frame)
    v2f wn[8];
    make_wn<G>(wn, t, tb.wn[t]);
    // Tail terms are loaded unconditionally (clamped row, times a 0/1 flag) so that all loads of a frame go out together
    // instead of one branch + vmcnt(0) each; tail buffers are zero-filled at allocation, unused tails stay zero.
#pragma unroll
    for (int j = 0; j < 16; j++) {
        v2f a = xin_p[T * j + t];
        if (j < KEEP) a += tprev[T * j + t] * f_prev;
        const int rn = j >= rows_owned ? j - rows_owned : 0;   // < KEEP because G >= ceil(KEEP/SH) whenever has_next
        a += town[T * rn + t] * (j >= rows_owned ? f_next : 0.f);
        xin[j] = a;
    }
#pragma unroll
    for (int j = 0; j < KEEP; j++) acc[j] = v2f{0.f, 0.f};

    // lane-pair addressing of two consecutive rows (see rows_from_quad): pair index of this lane's 16 bytes inside the row pair
    const bool even = (t & 1) == 0;
    const int lane_off = T * (t & 1) + (t & ~1);
    constexpr bool ROWQUAD = N >= GL_ROWQUAD_MIN_N;
    constexpr int NPAIR = ROWQUAD ? SH / 2 : 0;   // rows moved two per dwordx4 (otherwise every row as its own dwordx2, no DPP exchange)
    // Nyquist magnitudes of all the segment's frames: one load per wave (lane = frame, G <= kMaxSegFrames = 64), one v_readlane
    // per frame; narrower streams (several per wave) keep the per-frame vector load
    float nyq_all = 0.f;
    if constexpr (T >= 64) {
        const int l = threadIdx.x & 63;
        nyq_all = nyq_p[l < d.G ? l : 0];
        // consume the load HERE: a value still "in flight" at the loop header would make the compiler put a vmcnt wait in front
        // of the v_readlane of every frame, and vmcnt being in issue order that wait would drain the previous frame's stores
        asm volatile("v_mov_b32 %0, %0" : "+v"(nyq_all));
    }

    // The window is read from LDS ONCE per frame: the synthesis multiply at the end of frame g (y * w) and the analysis multiply of
    // frame g+1 (x * w) sit next to each other in time, so the 16 window registers loaded for the first serve the second as well,
    // and the windowed input v crosses the loop's back edge instead of being rebuilt at the top (16 of the ~156 LDS reads of a frame,
    // 16 KB of the ~158 KB a frame moves through the LDS: the kernel is power-limited and LDS bytes are part of the bill).
    // (ROT: where the register file takes it without spilling -- fused_one_window_read; the other instantiations read the window twice)
    constexpr bool ROT = fused_one_window_read(N, SH);
    const v4f* wl4 = reinterpret_cast<const v4f*>(win_l) + t;   // WPAIR: rows 2jp, 2jp+1 of this lane: wl4[T * jp]
    auto load_window = [&](v2f (&w_)[16]) {
        if constexpr (WPAIR) {
#pragma unroll
            for (int jp = 0; jp < 8; jp++) { const v4f q = wl4[T * jp]; w_[2 * jp] = v2f{q.x, q.y}; w_[2 * jp + 1] = v2f{q.z, q.w}; }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) w_[j] = wl[T * j];
        }
    };
    v2f v[16];
    if constexpr (ROT) {
        load_window(v);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] *= xin[j];
    }

    for (int g = 0; g < g_max; g++) {
        const bool active = live && g < d.G;
        const int gg = g < d.G ? g : d.G - 1;
        // this frame's magnitudes: issued now, not touched before the spectral step (a whole forward FFT of cover; fetching them a
        // frame ahead was measured: no change on any geometry, 16 more registers)
        MagRaw mraw;
        load_mag_rows<G>(mraw, t, mag_p + (int64_t)gg * G::C);
        if constexpr (T >= 64) mraw.nq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nyq_all), gg));
        else mraw.nq = nyq_p[gg + (t & 3)];
        st.mark(16);
        if constexpr (!ROT) {
            // all 16 window reads go out back to back (one LDS round trip, not sixteen), then the multiplies
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = wl[T * j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] *= xin[j];
        }
        st.mark(17);
        // The next frame's SH new rows (hop samples), two rows per dwordx4, and -- in the last frames of a segment with a
        // successor -- the successor's head from the tail buffer (rare: behind a wave-uniform branch).  The loaded values stay RAW
        // until the inverse FFT is done, so nothing waits for them where they are issued: before the forward FFT at N 4096 (the
        // inverse FFT alone does not cover their HBM latency there), after the spectral step otherwise (shorter live range: the
        // registers go to the pass-1 twiddles).
        const bool more = g + 1 < d.G;
        const int row0 = (g + 1) * SH + KEEP;                                // first of the SH new rows
        const bool any_next = more && d.has_next && row0 + SH - 1 >= rows_owned;
        v4f xq[NPAIR > 0 ? NPAIR : 1];
        v2f xs = v2f{0.f, 0.f};
        v2f xr[SH];          // (!ROWQUAD only)
        v2f tl[SH];          // (read only under any_next: no zero-fill of 3 SH registers in every frame)
        float fl[SH];
        auto issue_rows = [&] {
#pragma unroll
            for (int j = 0; j < KEEP; j++) xin[j] = xin[j + SH];             // slide the input window
#pragma unroll
            for (int pq = 0; pq < NPAIR; pq++)
                xq[pq] = *reinterpret_cast<const v4f*>(xin_p + T * (more ? row0 + 2 * pq : 0) + lane_off);   // clamped: unused when !more
            if constexpr (SH & 1) xs = xin_p[T * (more ? row0 + SH - 1 : 0) + t];
            if constexpr (!ROWQUAD) {
#pragma unroll
                for (int jj = 0; jj < SH - (SH & 1); jj++) xr[jj] = xin_p[T * (more ? row0 + jj : 0) + t];
            }
            if (any_next) {
#pragma unroll
                for (int jj = 0; jj < SH; jj++) {
                    const bool in_next = row0 + jj >= rows_owned;
                    tl[jj] = town[T * (in_next ? row0 + jj - rows_owned : 0) + t];
                    fl[jj] = in_next ? 1.f : 0.f;
                }
            }
        };
        if constexpr (!fused_late_rows(N)) issue_rows();
        st.mark(15);
        fft_forward<G, f_paired<G>()>(v, my, t, tw, sync, st);
        {
            MagRegs mr;
            finish_mags(mr, mraw, t == 0);
            spectral_apply<G>(v, t, mr, wn);
        }
        st.mark(7);
        if constexpr (fused_late_rows(N)) issue_rows();
        fft_inverse<G, f_paired<G>()>(v, my, t, tw, sync, st);
        // The next frame's rows: unpacked BEFORE this frame's stores are issued.  vmcnt counts loads and stores in issue order,
        // so a wait placed behind the (conditional) stores below would have to drain them too -- a full store round trip at the
        // end of every frame; here only loads issued a frame's worth of cycles ago are pending.
#pragma unroll
        for (int pq = 0; pq < NPAIR; pq++) rows_from_quad(xq[pq], even, xin[KEEP + 2 * pq], xin[KEEP + 2 * pq + 1]);
        if constexpr (SH & 1) xin[KEEP + SH - 1] = xs;
        if constexpr (!ROWQUAD) {
#pragma unroll
            for (int jj = 0; jj < SH - (SH & 1); jj++) xin[KEEP + jj] = xr[jj];
        }
        if (any_next) {
#pragma unroll
            for (int jj = 0; jj < SH; jj++) xin[KEEP + jj] += tl[jj] * fl[jj];
        }
#pragma unroll
        for (int jj = 0; jj < SH; jj++) asm volatile("" : "+v"(xin[KEEP + jj]));
        __builtin_amdgcn_sched_barrier(0);
        st.mark(18);
        v2f w[16];   // the window: all 16 reads go out back to back (one LDS round trip, not sixteen)
        if constexpr (ROT) {
            load_window(w);
            __builtin_amdgcn_sched_barrier(0);
        }
        auto W = [&](int j) { if constexpr (ROT) return w[j]; else return wl[T * j]; };
        if (active) {
            // overlap-add in registers: row j of this frame lands on accumulator row j (rows >= KEEP start a new sum);
            // the first SH rows are finished by this frame and leave (two rows per dwordx4), the rest shift down by SH.
            v2f* xo = xout_p + T * (g * SH);
            v2f yo[SH];
#pragma unroll
            for (int j = 0; j < SH; j++) {
                const v2f y = v[j] * W(j);
                yo[j] = (j < KEEP) ? acc[j] + y : y;
            }
#pragma unroll
            for (int pq = 0; pq < NPAIR; pq++) *reinterpret_cast<v4f*>(xo + T * (2 * pq) + lane_off) = quad_from_rows(yo[2 * pq], yo[2 * pq + 1], even);
            if constexpr (SH & 1) xo[T * (SH - 1) + t] = yo[SH - 1];
            if constexpr (!ROWQUAD) {
#pragma unroll
                for (int j = 0; j < SH - (SH & 1); j++) xo[T * j + t] = yo[j];
            }
            v2f nacc[KEEP];
#pragma unroll
            for (int j = 0; j < KEEP; j++) {
                const v2f y = v[j + SH] * W(j + SH);
                nacc[j] = (j + SH < KEEP) ? acc[j + SH] + y : y;
            }
#pragma unroll
            for (int j = 0; j < KEEP; j++) acc[j] = nacc[j];
            if (g == d.G - 1) {
                // what is left belongs past the owned range: next segment's head (tail buffer) or the utterance's end
                if (d.has_next) {
#pragma unroll
                    for (int j = 0; j < KEEP; j++) tout[T * j + t] = acc[j];
                } else {
#pragma unroll
                    for (int j = 0; j < KEEP; j++) xout_p[T * (rows_owned + j) + t] = acc[j];
                }
            }
        }
        // the next frame's windowed input, with the window registers of this frame's synthesis
        if constexpr (ROT) {
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = w[j] * xin[j];
        }
        st.mark(14);
    }
#ifdef GOMEL_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 20; k++) atomicAdd(&g_stamp_acc[k], st.acc[k]);
        atomicAdd(&g_stamp_acc[20], (unsigned long long)g_max);
    }
#endif
}

// =============================================================================================== fused path, any hop
// k_gl_fused covers hop = SH * N/16 because there moving to the next frame is a shift by whole register rows.  For every other
// (even) hop the same sliding-window scheme works with ONE more idea: rows are just the linear pair index i = t + T j cut into 16
// pieces, so a shift by hop/2 pairs is "write the 16 rows to the stream's LDS area, read them back hop/2 slots further".  Per
// frame that is two extra LDS passes -- the input window before the forward FFT (its registers then already hold the next
// frame's window, and the loads of its hop new samples fly during the FFTs, exactly as in k_gl_fused), the overlap-add
// accumulator after the inverse -- which a mock measured at +8 % of the fused kernel's time, against the 2.3x of the generic
// kernels (two launches, the frames' windowed outputs through HBM).  Same segments and ownership as k_gl_fused (a segment owns
// the samples [s0, s0 + G hop); what its frames add beyond goes to its tail buffer, N - hop floats), but the tails are FOLDED into
// x by a small kernel behind every launch (k_add_tails<float>, in place: +3 % of the iteration) instead of being read beside x by
// the next iteration: with a runtime hop the "does this pair reach into the successor's head" test would sit in front of every
// load of the last ceil((N-hop)/hop) frames of every segment -- a quarter of all frames at hop = N/7.  Runtime hop, one
// builds per N.  Needs: hop even, hop < N, and the planner's G >= ceil((N - hop) / hop) for segments with a successor.
// NR = how many of the 16 rows may receive new samples when the window moves on (rows 16-NR .. 15; the host picks the smallest
// of 4 / 8 / 16 that covers hop/2 pairs): those rows load from memory in straight-line code and select by lane; the others only
// read LDS.  Deciding that per row at run time put every load behind a branch, and the waits the compiler places at such joins
// (vmcnt(0) before each LDS read) serialised the frame: 0.87 ms per launch at N 4096 against 0.60 for k_gl_fused.
// ODD = the hop is odd: a frame then starts in the middle of a pair of its predecessor, so the shift runs on floats instead of
// pairs -- LDS reads of 8 bytes at 4-byte alignment (ds_read2_b32), two selects per pair, global accesses dword-aligned, and the
// one float that ends a frame's finished (or a segment's left-over) samples stored on its own.
template <int N, int NR, bool ODD>
__global__ __launch_bounds__(kFusedThreads, 2) void k_gl_fused_any(const float* __restrict__ x_in, float* __restrict__ x_out, float* __restrict__ tail_out,
                                                         const SegDesc* __restrict__ segs, int n_segs, int g_max,
                                                         const float* __restrict__ mag, const float* __restrict__ nyq, int hop, DevTables tb) {
    typedef Geo<N> G;
    constexpr int T = G::T, STREAMS = kFusedThreads / T, P = N / 2;
    static_assert(G::LDS_SLOTS > P, "the stream's exchange area must take one window of pairs (and the float behind it)");
    typedef v2f v2f_a4 __attribute__((aligned(4)));
    __shared__ __attribute__((aligned(16))) v2f lds[STREAMS * G::LDS_SLOTS + G::TW1_N + G::TW2_N + G::C];
    v2f* tw1_l = lds + STREAMS * G::LDS_SLOTS;
    v2f* tw2_l = tw1_l + G::TW1_N;
    v2f* win_l = tw2_l + G::TW2_N;
    fill_tables_lds<(STREAMS * G::LDS_SLOTS) % 2 == 0 && N >= 2048, G::TW1_N, G::TW2_N, G::C>(tw1_l, tw2_l, win_l, tb.tw1, tb.tw2, reinterpret_cast<const v2f*>(tb.window));
    __syncthreads();
    const int stream = threadIdx.x / T, t = threadIdx.x % T;
    // pass-2 twiddles in registers where the build has room for them (the 4-row builds at even hops: -3 % as in k_gl_fused); the
    // other builds need the registers for the rows they load, and read every twiddle from LDS
    constexpr bool W2R = NR == 4 && !ODD && N <= 2048;   // (N 4096: five spilled registers)
    v2f w2r[16];
    if constexpr (W2R) {
#pragma unroll
        for (int u = 0; u < G::NB2; u++)
#pragma unroll
            for (int q = 0; q < G::R2; q++) w2r[u * G::R2 + q] = tb.tw2[q * 8 + (t + G::T * u) / G::R1];
    }
    const Twiddles<v2f> tw{tw1_l, tw2_l, nullptr, nullptr, 0, W2R ? w2r : nullptr};
    auto sync = [] {
        if constexpr (Geo<N>::T <= 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    int seg = blockIdx.x * STREAMS + stream;
    const bool live = seg < n_segs;
    if (!live) seg = n_segs - 1;
    SegDesc d = segs[seg];
    v2f* my = lds + stream * G::LDS_SLOTS;
    if constexpr (T >= 64) {
        d.G = __builtin_amdgcn_readfirstlane(d.G);
        d.has_prev = __builtin_amdgcn_readfirstlane(d.has_prev);
        d.has_next = __builtin_amdgcn_readfirstlane(d.has_next);
        seg = __builtin_amdgcn_readfirstlane(seg);
    }
    constexpr bool WU = (T >= 64);
    auto uo = [](long long v) { if constexpr (WU) return uniform_i64(v); else return v; };
    const int h2 = hop / 2, keep2 = P - h2;                                   // pairs a frame moves on by / shares with its successor
    const v2f* xin_p = reinterpret_cast<const v2f*>(x_in) + uo(d.s0 / 2);     // pair i of frame g: xin_p[g h2 + i]
    v2f* xout_p = reinterpret_cast<v2f*>(x_out) + uo(d.s0 / 2);
    v2f* tout = reinterpret_cast<v2f*>(tail_out) + uo((int64_t)seg * keep2);
    // the same in floats (ODD): hop floats per frame, keepf = N - hop shared with the successor / left over at a segment's end
    const int keepf = N - hop;
    const float* xin_f = x_in + uo(d.s0);
    float* xout_f = x_out + uo(d.s0);
    float* tout_f = tail_out + uo((int64_t)seg * keepf);
    const float* myf = reinterpret_cast<const float*>(lds + stream * G::LDS_SLOTS);
    const float* mag_p = mag + uo(d.frame0 * G::C);
    const float* nyq_p = nyq + uo(d.frame0);
    const int owned2 = d.G * h2;   // pairs >= this (segment-relative) belong to the next segment
    if constexpr (STREAMS > 1) {
        __shared__ int g_sh[STREAMS];
        if (t == 0) g_sh[stream] = live ? d.G : 0;
        __syncthreads();
        int m = 0;
#pragma unroll
        for (int i = 0; i < STREAMS; i++) m = max(m, g_sh[i]);
        g_max = m;
    } else {
        g_max = d.G;
    }
    v2f xin[16], acc[16];
    const v2f* wl = win_l + t;
    v2f wn[8];
    make_wn<G>(wn, t, tb.wn[t]);
#pragma unroll
    for (int j = 0; j < 16; j++) {   // the first frame's window (x is complete: the tails of the previous iteration have been folded in)
        if constexpr (ODD) xin[j] = *reinterpret_cast<const v2f_a4*>(xin_f + 2 * (t + T * j));
        else xin[j] = xin_p[t + T * j];
        acc[j] = v2f{0.f, 0.f};
    }
    float nyq_all = 0.f;
    if constexpr (T >= 64) {
        const int l = threadIdx.x & 63;
        nyq_all = nyq_p[l < d.G ? l : 0];
        asm volatile("v_mov_b32 %0, %0" : "+v"(nyq_all));
    }
    // the first window must have ARRIVED before the loop: were its loads still in flight at the loop header, the compiler would
    // count them down in front of the first LDS writes of every frame (vmcnt is in issue order) -- and in steady state that
    // countdown waits for the previous frame's stores instead
#pragma unroll
    for (int j = 0; j < 16; j++) asm volatile("" : "+v"(xin[j]));
    for (int g = 0; g < g_max; g++) {
        const bool active = live && g < d.G;
        const int gg = g < d.G ? g : d.G - 1;
        MagRaw mraw;
        load_mag_rows<G>(mraw, t, mag_p + (int64_t)gg * G::C);
        if constexpr (T >= 64) mraw.nq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nyq_all), gg));
        else mraw.nq = nyq_p[gg + (t & 3)];
        v2f v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = wl[T * j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] *= xin[j];
        // slide the input window by hop/2 pairs: through LDS for the pairs that stay, from memory for the hop/2 new ones (their
        // loads are in flight during the FFTs; nothing reads xin again before the next frame)
        const bool more = g + 1 < d.G;
        const int base2 = (g + 1) * h2;                                   // segment-relative pair index of the next frame's pair 0
#pragma unroll
        for (int j = 0; j < 16; j++) my[t + T * j] = xin[j];
        sync();
        const unsigned nb2 = (unsigned)(more ? base2 : 0);               // (clamped address when there is no next frame)
        const unsigned nbf = (unsigned)(more ? (g + 1) * hop : 0);       // the same in floats
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int i = t + T * j;
            if constexpr (ODD) {
                const int f0 = 2 * i + hop;                               // the floats that move down to this slot: f0, f0 + 1, if inside the window
                const v2f a = *reinterpret_cast<const v2f_a4*>(myf + (f0 < N ? f0 : 0));
                if (j >= 16 - NR) {
                    const v2f gl = *reinterpret_cast<const v2f_a4*>(xin_f + nbf + 2u * (unsigned)i);
                    xin[j] = v2f{f0 < N ? a.x : gl.x, f0 + 1 < N ? a.y : gl.y};
                } else xin[j] = a;
            } else {
                const v2f a = my[i < keep2 ? i + h2 : 0];                  // the pair that moves down to this slot, if one does
                if (j >= 16 - NR) {                                        // (compile-time) rows that may take new samples: by lane
                    const v2f gl = xin_p[nb2 + (unsigned)i];
                    xin[j] = i < keep2 ? a : gl;
                } else xin[j] = a;
            }
        }
        sync();   // the exchange area goes to the FFT now
        fft_forward<G, (GL_PAIRED & 1)>(v, my, t, tw, sync);
        {
            MagRegs mr;
            finish_mags(mr, mraw, t == 0);
            spectral_apply<G>(v, t, mr, wn);
        }
        fft_inverse<G, (GL_PAIRED & 1)>(v, my, t, tw, sync);
        // The next frame's new samples have had two FFTs to arrive: take delivery HERE, before this frame's stores are issued.
        // vmcnt counts loads and stores in issue order and the stores below sit behind wave-uniform branches (their number is not
        // known at compile time), so a wait for these loads placed after them -- at the top of the next frame -- would be a
        // vmcnt(0): a full store round trip per frame (measured: the whole difference to k_gl_fused, 30 % at N 4096).
#pragma unroll
        for (int j = 0; j < 16; j++) asm volatile("" : "+v"(xin[j]));
        __builtin_amdgcn_sched_barrier(0);
        // overlap-add in registers; the first hop/2 pairs are finished by this frame and leave; the rest moves down by hop/2 pairs
#pragma unroll
        for (int j = 0; j < 16; j++) acc[j] += v[j] * wl[T * j];
        if (active) {
            v2f* xo = xout_p + g * h2;
            float* xof = xout_f + g * hop;
#pragma unroll
            for (int j = 0; j < NR; j++) {                                 // rows 0 .. NR-1 hold every finished sample (host: NR covers them)
                const int i = t + T * j;
                if constexpr (ODD) {
                    if (2 * i + 1 < hop) *reinterpret_cast<v2f_a4*>(xof + 2 * i) = acc[j];
                    else if (2 * i < hop) xof[2 * i] = acc[j].x;            // the odd hop's last sample
                } else {
                    if (i < h2) xo[i] = acc[j];
                }
            }
        }
        // (no barrier here: fft_inverse ends with one behind its last reads of the exchange area -- only butterflies follow it)
#pragma unroll
        for (int j = 0; j < 16; j++) my[t + T * j] = acc[j];
        sync();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int i = t + T * j;
            if constexpr (ODD) {
                const int f0 = 2 * i + hop;
                const v2f a = *reinterpret_cast<const v2f_a4*>(myf + (f0 < N ? f0 : 0));
                acc[j] = v2f{f0 < N ? a.x : 0.f, f0 + 1 < N ? a.y : 0.f};
            } else {
                const v2f a = my[i < keep2 ? i + h2 : 0];
                acc[j] = i < keep2 ? a : v2f{0.f, 0.f};
            }
        }
        sync();   // ... before the next frame's window goes through the same area
        if (active && g == d.G - 1) {
            // what is left (N - hop samples) lies past the owned range: the next segment's head (tail buffer) or the utterance's end
            v2f* dst = d.has_next ? tout : xout_p + owned2;
            float* dstf = d.has_next ? tout_f : xout_f + d.G * hop;
            // (the lane index is laundered so that these sixteen addresses are formed HERE, once per segment: hoisted out of the frame loop as
            // 64-bit values they did not fit the 256 registers of a two-waves-per-SIMD kernel, were spilled, and every store below reloaded its
            // address from scratch behind a vmcnt(0) -- 10 to 28 such reloads per segment in the N 2048 / 4096 builds)
            int tl = t;
            asm volatile("" : "+v"(tl));
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int i = tl + T * j;
                if constexpr (ODD) {
                    if (2 * i + 1 < keepf) *reinterpret_cast<v2f_a4*>(dstf + 2 * i) = acc[j];
                    else if (2 * i < keepf) dstf[2 * i] = acc[j].x;
                } else {
                    if (i < keep2) dst[i] = acc[j];
                }
            }
        }
    }
}

// =============================================================================================== fused path, fp64 ("64-bit mode")
// The same sliding-window kernel in double precision: what makes 64-bit mode (the only mode that agrees with the reference's fp64
// arithmetic on EVERY input) affordable.  Same segments, same ownership / tail rule, same lane-slot magnitude
// rows (double), same FFT core (its C++ forms: the packed-math asm blocks are fp32-only).  Differences forced by the type: one
// wave per SIMD (launch_bounds(256, 1): 374 registers, 118 of them AGPRs -- 16 complex doubles of frame data, 16 of input window, the
// accumulator, magnitudes, twiddles), all LDS dynamic (152 KB at N 4096: exchange areas + TW1 + TW2 + window + W_N^k in double), a lane's sample pair
// is already 16 bytes (plain dwordx4 loads / stores, no DPP row pairing).  Algorithmic bytes per frame-iteration: twice the fp32
// kernel's.  The fp64 vector unit runs at half the fp32 rate and packs nothing, so expect ~2.3x the fp32 time.
// Build knobs of k_gl_fused_d / k_gl_fused_any_d (round 4).  The kernel allocates ALL 512 registers of its one wave per SIMD (256 + 246
// AGPRs, to which the compiler sends global loads and parks what is live across the FFTs: ~260 v_accvgpr_read per frame, 12 % of its
// vector instructions).  GL_D_WN_LDS=1 (default): a lane's eight W_N^k of the spectral step live in LDS (8 x T pairs, read per frame)
// instead of 32 registers for the whole launch: 380 registers, ~100 fewer AGPR reads per frame, -2.2 % at N 4096, -1.8 % at 2048, -5.4 %
// at 1024.  GL_D_W1CARRY=0 (reads the pass-1 twiddles from LDS in both FFTs): +1.8 %, stays 1.  profiles/r04/ab_fp64_registers.txt.
#ifndef GL_D_W1CARRY
#define GL_D_W1CARRY 1
#endif
#ifndef GL_D_WN_LDS
#define GL_D_WN_LDS 1
#endif
#ifndef GL_D_WN_LDS_MINN
#define GL_D_WN_LDS_MINN 256
#endif
struct MagRawD { double a[16]; double nq; };
template <class G> GL_DEV void load_mag_rows_d(MagRawD& r, int t, const double* __restrict__ mag_row) {
#pragma unroll
    for (int e = 0; e < 16; e++) r.a[e] = mag_row[((e / 4) * G::T + t) * 4 + (e % 4)];
}
GL_DEV void finish_mags_d(MagRegsD& m, const MagRawD& r, bool is0) {
#pragma unroll
    for (int e = 0; e < 8; e++) { m.ma[e] = r.a[e]; m.mb[e] = r.a[8 + e]; }
    m.m0 = is0 ? m.mb[4] : 0.0;
    m.mn = is0 ? r.nq : 0.0;
    m.mb[4] = is0 ? m.ma[4] : m.mb[4];
}

template <int N, int SH>
__global__ __launch_bounds__(kFusedThreads, 1) void k_gl_fused_d(const double* __restrict__ x_in, const double* __restrict__ tail_in,
                                                                double* __restrict__ x_out, double* __restrict__ tail_out,
                                                                const SegDesc* __restrict__ segs, int n_segs, int g_max,
                                                                const double* __restrict__ mag, const double* __restrict__ nyq, DevTables tb) {
    typedef Geo<N> G;
    constexpr int T = G::T, STREAMS = kFusedThreads / T, KEEP = 16 - SH, TAIL_PAIRS = KEEP * T;
    extern __shared__ __attribute__((aligned(16))) char fused_d_smem[];
    v2d* lds = reinterpret_cast<v2d*>(fused_d_smem);      // [STREAMS * LDS_SLOTS | TW1 | TW2 | window pairs | STREAMS ints]
    v2d* tw1_l = lds + STREAMS * G::LDS_SLOTS;
    v2d* tw2_l = tw1_l + G::TW1_N;
    v2d* win_l = tw2_l + G::TW2_N;
    constexpr bool WN_LDS = GL_D_WN_LDS && N >= GL_D_WN_LDS_MINN;
    v2d* wn_l = win_l + G::C;                                  // [8][T] (WN_LDS only)
    int* g_sh = reinterpret_cast<int*>(wn_l + (WN_LDS ? 8 * T : 0));
    unsigned* cnt_sh = reinterpret_cast<unsigned*>(g_sh + STREAMS);   // [STREAMS][2]: one event counter per wave of a two-wave stream (N 4096)
    if constexpr (T > 64) { if (threadIdx.x < 2 * STREAMS) cnt_sh[threadIdx.x] = 0u; }
    if constexpr (WN_LDS) {
        if (threadIdx.x < T) {
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++) wn_l[s8 * T + threadIdx.x] = tb.wnd[slot_bin<G>(threadIdx.x, s8)];
        }
    }
    for (int i = threadIdx.x; i < G::TW1_N; i += kFusedThreads) tw1_l[i] = tb.tw1d[i];
    for (int i = threadIdx.x; i < G::TW2_N; i += kFusedThreads) tw2_l[i] = tb.tw2d[i];
    for (int i = threadIdx.x; i < G::C; i += kFusedThreads) win_l[i] = reinterpret_cast<const v2d*>(tb.windowd)[i];
    __syncthreads();
    const Twiddles<v2d> tw{tw1_l, tw2_l, nullptr};
    const int stream = threadIdx.x / T, t = threadIdx.x % T;
    // The two waves of a 4096-point stream synchronise through per-wave event counters in LDS (StreamFlagSync) instead of the workgroup's
    // s_barrier: the workgroup's two streams no longer run in lockstep -- with one wave per SIMD the four waves used to be released
    // together by every barrier and to send their 16 x ds_read_b128 bursts into the one LDS pipe at the same moment -- and the
    // read -> write points become split-phase (the wait sits in front of the next write burst, where it never stalls):
    // 1.534 -> 1.378 ms per launch at config 3 (profiles/r03/ab_fp64_flagsync.txt).  Streams inside one wave need neither.
    unsigned ev = 0, sync_cap = tb.sync_cap;
    StreamFlagSync<(Geo<N>::T > 64)> sync{cnt_sh + 2 * stream + ((threadIdx.x % T) >> 6), cnt_sh + 2 * stream + (((threadIdx.x % T) >> 6) ^ 1), ev, sync_cap, tb.sync_err};
    int seg = blockIdx.x * STREAMS + stream;
    const bool live = seg < n_segs;
    if (!live) seg = n_segs - 1;
    SegDesc d = segs[seg];
    v2d* my = lds + stream * G::LDS_SLOTS;
    if constexpr (T >= 64) {
        d.G = __builtin_amdgcn_readfirstlane(d.G);
        d.has_prev = __builtin_amdgcn_readfirstlane(d.has_prev);
        d.has_next = __builtin_amdgcn_readfirstlane(d.has_next);
        seg = __builtin_amdgcn_readfirstlane(seg);
    }
    constexpr bool WU = (T >= 64);
    auto uo = [](long long v) { if constexpr (WU) return uniform_i64(v); else return v; };
    const int seg_prev = d.has_prev ? seg - 1 : seg;
    const v2d* xin_p = reinterpret_cast<const v2d*>(x_in) + uo(d.s0 / 2);
    v2d* xout_p = reinterpret_cast<v2d*>(x_out) + uo(d.s0 / 2);
    const v2d* tprev = reinterpret_cast<const v2d*>(tail_in) + uo((int64_t)seg_prev * TAIL_PAIRS);
    const v2d* town = reinterpret_cast<const v2d*>(tail_in) + uo((int64_t)seg * TAIL_PAIRS);
    v2d* tout = reinterpret_cast<v2d*>(tail_out) + uo((int64_t)seg * TAIL_PAIRS);
    const double* mag_p = mag + uo(d.frame0 * G::C);
    const double* nyq_p = nyq + uo(d.frame0);
    const double f_prev = d.has_prev ? 1.0 : 0.0, f_next = d.has_next ? 1.0 : 0.0;
    const int rows_owned = d.G * SH;
    constexpr bool COUPLED = STREAMS > 1 && T <= 64;   // (two-wave streams have their own synchronisation: each runs its own length)
    if constexpr (COUPLED) {
        if (t == 0) g_sh[stream] = live ? d.G : 0;
        __syncthreads();
        int m = 0;
#pragma unroll
        for (int i = 0; i < STREAMS; i++) m = max(m, g_sh[i]);
        g_max = m;
    } else {
        g_max = d.G;
    }
    v2d xin[16], acc[KEEP], wn[8];
    const v2d* wl = win_l + t;
    if constexpr (!WN_LDS) load_wn<G>(wn, t, tb.wnd);   // (rebuilt per frame from one table value, as make_wn does for the fp32 kernel: +3 % vector instructions here)
#pragma unroll
    for (int j = 0; j < 16; j++) {
        v2d a = xin_p[T * j + t];
        if (j < KEEP) a += tprev[T * j + t] * f_prev;
        const int rn = j >= rows_owned ? j - rows_owned : 0;
        a += town[T * rn + t] * (j >= rows_owned ? f_next : 0.0);
        xin[j] = a;
    }
#pragma unroll
    for (int j = 0; j < KEEP; j++) acc[j] = v2d{0.0, 0.0};
    // One pass-1 twiddle read per frame at N 4096: the 15 values the inverse FFT reads at its end serve the next frame's forward FFT,
    // which starts a few hundred instructions later (15 of a frame's ~156 LDS reads; +1.6 % vector instructions for the registers
    // that cross the back edge; measured -1.2 % at N 4096, +1.7 % at N 1024 -- there the table is read twice, profiles/r03/ab_fp64_w1carry.txt)
    constexpr bool W1CARRY = GL_D_W1CARRY && N >= 4096;
    v2d w1c[16];
    if constexpr (W1CARRY) {
#pragma unroll
        for (int u = 0; u < G::NB1; u++)
#pragma unroll
            for (int q = 0; q < G::R1; q++) w1c[u + G::NB1 * q] = tw1_l[q * G::C1 + t + T * u];
    }

#ifdef GOMEL_STAMPS   // diagnostic build only (scripts/stamp_probe.py with GOMEL_FP64_ITERS=-1): the slots of k_gl_fused's table
    Stamps st; st.start();
#else
    NoStamps st;
#endif
    for (int g = 0; g < g_max; g++) {
        const bool active = live && g < d.G;
        const int gg = g < d.G ? g : d.G - 1;
        MagRawD mraw;                                                         // issued now, finished after the forward FFT
        load_mag_rows_d<G>(mraw, t, mag_p + (int64_t)gg * G::C);
        mraw.nq = nyq_p[gg];
        st.mark(16);
        // (k_gl_fused's single window read per frame was tried here too: 64 more live registers around the back edge, 3 % MORE
        // vector instructions through the AGPR traffic they cause -- the window is read twice)
        v2d v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = wl[T * j] * xin[j];
#pragma unroll
        for (int j = 0; j < KEEP; j++) xin[j] = xin[j + SH];
        const bool more = g + 1 < d.G;
        const int row0 = (g + 1) * SH + KEEP;
        const bool any_next = more && d.has_next && row0 + SH - 1 >= rows_owned;
        v2d xl[SH], tl[SH];
        double fl[SH];
#pragma unroll
        for (int jj = 0; jj < SH; jj++) xl[jj] = xin_p[T * (more ? row0 + jj : 0) + t];
        if (any_next) {
#pragma unroll
            for (int jj = 0; jj < SH; jj++) {
                const bool in_next = row0 + jj >= rows_owned;
                tl[jj] = town[T * (in_next ? row0 + jj - rows_owned : 0) + t];
                fl[jj] = in_next ? 1.0 : 0.0;
            }
        }
        st.mark(15);   // (window reads + multiply, slide, next rows' loads issued)
        if constexpr (W1CARRY) {
            const Twiddles<v2d> twf{tw1_l, tw2_l, nullptr, w1c, G::R1 - 1, nullptr};
            fft_forward<G, d_paired<G>()>(v, my, t, twf, sync, st);
        } else {
            fft_forward<G, d_paired<G>()>(v, my, t, tw, sync, st);
        }
        {
            MagRegsD mr;
            finish_mags_d(mr, mraw, t == 0);
            if constexpr (WN_LDS) {
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) wn[s8] = wn_l[s8 * T + t];
            }
            spectral_apply<G>(v, t, mr, wn);
        }
        st.mark(7);
        fft_inverse<G, d_paired<G>()>(v, my, t, tw, sync, st, W1CARRY ? w1c : nullptr);
#pragma unroll
        for (int jj = 0; jj < SH; jj++) xin[KEEP + jj] = xl[jj];
        if (any_next) {
#pragma unroll
            for (int jj = 0; jj < SH; jj++) xin[KEEP + jj] += tl[jj] * fl[jj];
        }
#pragma unroll
        for (int jj = 0; jj < SH; jj++) asm volatile("" : "+v"(xin[KEEP + jj]));
        __builtin_amdgcn_sched_barrier(0);
        st.mark(18);
        if (active) {
            v2d* xo = xout_p + T * (g * SH);
#pragma unroll
            for (int j = 0; j < SH; j++) {
                const v2d y = v[j] * wl[T * j];
                xo[T * j + t] = (j < KEEP) ? acc[j] + y : y;
            }
            v2d nacc[KEEP];
#pragma unroll
            for (int j = 0; j < KEEP; j++) {
                const v2d y = v[j + SH] * wl[T * (j + SH)];
                nacc[j] = (j + SH < KEEP) ? acc[j + SH] + y : y;
            }
#pragma unroll
            for (int j = 0; j < KEEP; j++) acc[j] = nacc[j];
            if (g == d.G - 1) {
                if (d.has_next) {
#pragma unroll
                    for (int j = 0; j < KEEP; j++) tout[T * j + t] = acc[j];
                } else {
#pragma unroll
                    for (int j = 0; j < KEEP; j++) xout_p[T * (rows_owned + j) + t] = acc[j];
                }
            }
        }
        st.mark(14);
    }
#ifdef GOMEL_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 20; k++) atomicAdd(&g_stamp_acc[k], st.acc[k]);
        atomicAdd(&g_stamp_acc[20], (unsigned long long)g_max);
    }
#endif
}
// The any-hop form of the fp64 fused kernel: k_gl_fused_any's scheme (window and overlap-add accumulator shifted through the stream's
// LDS area by hop samples per frame, tails folded into x behind every launch) on k_gl_fused_d's types, so that 64-bit mode keeps a
// fused kernel at hops that are no multiple of N/16 (441, 882, 275 ...) instead of falling to the generic fp64 pair.  ODD shifts by
// doubles (16-byte LDS reads at 8-byte alignment).  NR: rows that may take new samples, 8 (hops up to N/2) or 16.
template <int N, int NR, bool ODD>
__global__ __launch_bounds__(kFusedThreads, 1) void k_gl_fused_any_d(const double* __restrict__ x_in, double* __restrict__ x_out, double* __restrict__ tail_out,
                                                                    const SegDesc* __restrict__ segs, int n_segs, int g_max,
                                                                    const double* __restrict__ mag, const double* __restrict__ nyq, int hop, DevTables tb) {
    typedef Geo<N> G;
    constexpr int T = G::T, STREAMS = kFusedThreads / T, P = N / 2;
    static_assert(G::LDS_SLOTS > P, "the stream's exchange area must take one window of pairs (and the double behind it)");
    typedef v2d v2d_a8 __attribute__((aligned(8)));
    extern __shared__ __attribute__((aligned(16))) char fused_d_smem[];
    v2d* lds = reinterpret_cast<v2d*>(fused_d_smem);      // [STREAMS * LDS_SLOTS | TW1 | TW2 | window pairs | STREAMS ints]
    v2d* tw1_l = lds + STREAMS * G::LDS_SLOTS;
    v2d* tw2_l = tw1_l + G::TW1_N;
    v2d* win_l = tw2_l + G::TW2_N;
    constexpr bool WN_LDS = GL_D_WN_LDS && N >= GL_D_WN_LDS_MINN;
    v2d* wn_l = win_l + G::C;                                  // [8][T] (WN_LDS: a lane's eight W_N^k, read per frame -- k_gl_fused_d)
    int* g_sh = reinterpret_cast<int*>(wn_l + (WN_LDS ? 8 * T : 0));
    if constexpr (WN_LDS) {
        if (threadIdx.x < T) {
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++) wn_l[s8 * T + threadIdx.x] = tb.wnd[slot_bin<G>(threadIdx.x, s8)];
        }
    }
    for (int i = threadIdx.x; i < G::TW1_N; i += kFusedThreads) tw1_l[i] = tb.tw1d[i];
    for (int i = threadIdx.x; i < G::TW2_N; i += kFusedThreads) tw2_l[i] = tb.tw2d[i];
    for (int i = threadIdx.x; i < G::C; i += kFusedThreads) win_l[i] = reinterpret_cast<const v2d*>(tb.windowd)[i];
    __syncthreads();
    const Twiddles<v2d> tw{tw1_l, tw2_l, nullptr};
    const int stream = threadIdx.x / T, t = threadIdx.x % T;
    // (k_gl_fused_d's StreamFlagSync was measured here, plain and split-phase: 1.81 -> 2.41 ms per launch at N 4096 / hop 882 either way --
    // this kernel has six more exchange points per frame for its two shift passes, and spills.  s_barrier stays; the sync_* helpers below
    // are then plain barriers where the kernel had them before.)
    auto sync = [] {
        if constexpr (Geo<N>::T <= 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    int seg = blockIdx.x * STREAMS + stream;
    const bool live = seg < n_segs;
    if (!live) seg = n_segs - 1;
    SegDesc d = segs[seg];
    v2d* my = lds + stream * G::LDS_SLOTS;
    if constexpr (T >= 64) {
        d.G = __builtin_amdgcn_readfirstlane(d.G);
        d.has_next = __builtin_amdgcn_readfirstlane(d.has_next);
        seg = __builtin_amdgcn_readfirstlane(seg);
    }
    constexpr bool WU = (T >= 64);
    auto uo = [](long long v) { if constexpr (WU) return uniform_i64(v); else return v; };
    const int h2 = hop / 2, keep2 = P - h2;                                   // pairs a frame moves on by / shares with its successor
    const v2d* xin_p = reinterpret_cast<const v2d*>(x_in) + uo(d.s0 / 2);     // pair i of frame g: xin_p[g h2 + i]
    v2d* xout_p = reinterpret_cast<v2d*>(x_out) + uo(d.s0 / 2);
    v2d* tout = reinterpret_cast<v2d*>(tail_out) + uo((int64_t)seg * keep2);
    const int keepf = N - hop;                                                // the same in doubles (ODD)
    const double* xin_f = x_in + uo(d.s0);
    double* xout_f = x_out + uo(d.s0);
    double* tout_f = tail_out + uo((int64_t)seg * keepf);
    const double* myf = reinterpret_cast<const double*>(lds + stream * G::LDS_SLOTS);
    const double* mag_p = mag + uo(d.frame0 * G::C);
    const double* nyq_p = nyq + uo(d.frame0);
    const int owned2 = d.G * h2;
    if constexpr (STREAMS > 1) {
        if (t == 0) g_sh[stream] = live ? d.G : 0;
        __syncthreads();
        int m = 0;
#pragma unroll
        for (int i = 0; i < STREAMS; i++) m = max(m, g_sh[i]);
        g_max = m;
    } else {
        g_max = d.G;
    }
    v2d xin[16], acc[16], wn[8];
    const v2d* wl = win_l + t;
    if constexpr (!WN_LDS) load_wn<G>(wn, t, tb.wnd);
#pragma unroll
    for (int j = 0; j < 16; j++) {   // the first frame's window (x is complete: the tails of the previous iteration have been folded in)
        if constexpr (ODD) xin[j] = *reinterpret_cast<const v2d_a8*>(xin_f + 2 * (t + T * j));
        else xin[j] = xin_p[t + T * j];
        acc[j] = v2d{0.0, 0.0};
    }
#pragma unroll
    for (int j = 0; j < 16; j++) asm volatile("" : "+v"(xin[j]));   // arrived before the loop (k_gl_fused_any: the vmcnt countdown at the loop top)
    for (int g = 0; g < g_max; g++) {
        const bool active = live && g < d.G;
        const int gg = g < d.G ? g : d.G - 1;
        MagRawD mraw;
        load_mag_rows_d<G>(mraw, t, mag_p + (int64_t)gg * G::C);
        mraw.nq = nyq_p[gg];
        v2d v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = wl[T * j] * xin[j];
        // slide the input window by hop samples: through LDS for what stays, from memory for the new samples (in flight during the FFTs)
        const bool more = g + 1 < d.G;
        const int base2 = (g + 1) * h2;
        sync_before_write(sync);
#pragma unroll
        for (int j = 0; j < 16; j++) my[t + T * j] = xin[j];
        sync();
        const unsigned nb2 = (unsigned)(more ? base2 : 0);
        const unsigned nbf = (unsigned)(more ? (g + 1) * hop : 0);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int i = t + T * j;
            if constexpr (ODD) {
                const int f0 = 2 * i + hop;                               // the doubles that move down to this slot: f0, f0 + 1, if inside the window
                const v2d a = *reinterpret_cast<const v2d_a8*>(myf + (f0 < N ? f0 : 0));
                if (j >= 16 - NR) {
                    const v2d gl = *reinterpret_cast<const v2d_a8*>(xin_f + nbf + 2u * (unsigned)i);
                    xin[j] = v2d{f0 < N ? a.x : gl.x, f0 + 1 < N ? a.y : gl.y};
                } else xin[j] = a;
            } else {
                const v2d a = my[i < keep2 ? i + h2 : 0];
                if (j >= 16 - NR) {
                    const v2d gl = xin_p[nb2 + (unsigned)i];
                    xin[j] = i < keep2 ? a : gl;
                } else xin[j] = a;
            }
        }
        sync_reads_done(sync);   // the exchange area goes to the FFT now (its first write burst waits: sync_before_write)
        fft_forward<G, g_paired_d<G>()>(v, my, t, tw, sync);
        {
            MagRegsD mr;
            finish_mags_d(mr, mraw, t == 0);
            if constexpr (WN_LDS) {
#pragma unroll
                for (int s8 = 0; s8 < 8; s8++) wn[s8] = wn_l[s8 * T + t];
            }
            spectral_apply<G>(v, t, mr, wn);
        }
        fft_inverse<G, g_paired_d<G>()>(v, my, t, tw, sync);
#pragma unroll
        for (int j = 0; j < 16; j++) asm volatile("" : "+v"(xin[j]));   // the new samples are taken in BEFORE this frame's stores are issued
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; j++) acc[j] += v[j] * wl[T * j];
        if (active) {
            v2d* xo = xout_p + g * h2;
            double* xof = xout_f + g * hop;
#pragma unroll
            for (int j = 0; j < NR; j++) {                                 // rows 0 .. NR-1 hold every finished sample (host: NR covers them)
                const int i = t + T * j;
                if constexpr (ODD) {
                    if (2 * i + 1 < hop) *reinterpret_cast<v2d_a8*>(xof + 2 * i) = acc[j];
                    else if (2 * i < hop) xof[2 * i] = acc[j].x;            // the odd hop's last sample
                } else {
                    if (i < h2) xo[i] = acc[j];
                }
            }
        }
        // (no barrier here: fft_inverse ends with one behind its last reads of the exchange area -- only butterflies follow it)
#pragma unroll
        for (int j = 0; j < 16; j++) my[t + T * j] = acc[j];
        sync();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int i = t + T * j;
            if constexpr (ODD) {
                const int f0 = 2 * i + hop;
                const v2d a = *reinterpret_cast<const v2d_a8*>(myf + (f0 < N ? f0 : 0));
                acc[j] = v2d{f0 < N ? a.x : 0.0, f0 + 1 < N ? a.y : 0.0};
            } else {
                const v2d a = my[i < keep2 ? i + h2 : 0];
                acc[j] = i < keep2 ? a : v2d{0.0, 0.0};
            }
        }
        sync_reads_done(sync);   // ... before the next frame's window goes through the same area
        if (active && g == d.G - 1) {
            v2d* dst = d.has_next ? tout : xout_p + owned2;
            double* dstf = d.has_next ? tout_f : xout_f + d.G * hop;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int i = t + T * j;
                if constexpr (ODD) {
                    if (2 * i + 1 < keepf) *reinterpret_cast<v2d_a8*>(dstf + 2 * i) = acc[j];
                    else if (2 * i < keepf) dstf[2 * i] = acc[j].x;
                } else {
                    if (i < keep2) dst[i] = acc[j];
                }
            }
        }
    }
}
template <int N> static constexpr size_t fused_d_lds_bytes() {
    typedef Geo<N> G;
    return ((size_t)(kFusedThreads / G::T) * G::LDS_SLOTS + G::TW1_N + G::TW2_N + G::C + ((GL_D_WN_LDS && N >= GL_D_WN_LDS_MINN) ? 8 * G::T : 0)) * 16 + (size_t)(kFusedThreads / G::T) * 12 + 16;   // + g_sh[STREAMS] + the event counters [STREAMS][2]
}
// out = x + tails, both fp64 (the fp64 fused path's last step)
__global__ __launch_bounds__(256) void k_add_tails_d(const SegDesc* __restrict__ segs, const double* __restrict__ tail, double* __restrict__ out,
                                                     int tail_len, int hop) {
    const SegDesc d = segs[blockIdx.x];
    if (!d.has_next) return;
    double* o = out + d.s0 + (int64_t)d.G * hop;
    const double* tp = tail + (int64_t)blockIdx.x * tail_len;
    for (int i = threadIdx.x; i < tail_len; i += blockDim.x) o[i] += tp[i];
}

// out += tails (fp64), after k_finalize has converted the owned samples
// (x and the tail are both fp32: the sum is formed in fp32 first when the output is fp32, exactly as the next iteration would
// have read it, and in fp64 when the output is fp64)
template <class OutT>
__global__ __launch_bounds__(256) void k_add_tails(const SegDesc* __restrict__ segs, const float* __restrict__ tail, OutT* __restrict__ out,
                                                   int tail_len, int hop) {
    const SegDesc d = segs[blockIdx.x];
    if (!d.has_next) return;
    OutT* o = out + d.s0 + (int64_t)d.G * hop;
    const float* tp = tail + (int64_t)blockIdx.x * tail_len;
    for (int i = threadIdx.x; i < tail_len; i += blockDim.x) o[i] += (OutT)tp[i];
}


// =============================================================================================== ToMel (forward path), fp64
// gossp STFT (window + fft.FFTReal, mel/mel.go:52) -> |S[j]|, j in [0, N/2]  (mel/mel.go:55-63 pairs |S[j]| with
// |S[N-1-j]| = |S[j+1]|, so one array of N/2+1 magnitudes serves both channels).  fp64 because ln() of bins near the
// 1e-5 floor needs more dynamic range than an fp32 FFT keeps.
template <int N>
__global__ __launch_bounds__(128) void k_stft_mag(const double* __restrict__ wave, const int64_t* __restrict__ frame_off, double* __restrict__ smag,
                                                  int64_t n_frames, int hop, DevTables tb) {
    typedef Geo<N> G;
    constexpr int STREAMS = 128 / G::T;
    __shared__ v2d lds[STREAMS * G::LDS_SLOTS];
    const int stream = threadIdx.x / G::T, t = threadIdx.x % G::T;
    int64_t frame = (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < n_frames;
    if (!live) frame = n_frames - 1;
    v2d* my = lds + stream * G::LDS_SLOTS;
    const Twiddles<v2d> tw{tb.tw1d, tb.tw2d, tb.wnd};
    auto sync = [] { __syncthreads(); };
    const double* xs = wave + (frame_off ? frame_off[frame] : frame * hop);   // batched ToMel: frames of many utterances, offsets from the host
    v2d v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int p = 2 * (t + G::T * j);
        v[j] = v2d{xs[p] * tb.windowd[p], xs[p + 1] * tb.windowd[p + 1]};
    }
    fft_forward<G>(v, my, t, tw, sync);
    if (!live) return;
    double* out = smag + frame * (G::C + 1);
    const bool is0 = (t == 0);
    const v2d dc = v[0];
    lane0_permute<G>(v, is0);
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int k = slot_bin<G>(t, s);
        const v2d a = v[s], b = v[15 - s], wk = tb.wnd[k];
        const v2d bc = v2d{b.x, -b.y};
        const v2d p = a + bc, d = a - bc;
        const v2d q = cmul(wk, v2d{d.y, -d.x});
        const v2d xk = p + q, xc = p - q;  // 2 X[k], 2 conj(X[C-k])
        out[k] = 0.5 * hypot(xk.x, xk.y);
        out[G::C - k] = 0.5 * hypot(xc.x, xc.y);
    }
    if (is0) { out[0] = fabs(dc.x + dc.y); out[G::C] = fabs(dc.x - dc.y); }
}

// domel (mel/impl.go:310-345) + spectral_normalize (mel/impl.go:410-419): one block per frame.  The frame's N/2+1 magnitudes are
// staged in LDS with coalesced loads first (a band's row walks a run of bins: straight from global memory every thread strode
// through its own run -- 1.0 TB/s; from LDS the kernel is bound by the 16 KB it reads and the 3 KB it writes per frame).
// (staged = 0: frame lengths above 8192, whose magnitudes outgrow the default LDS allowance, read them from global memory)
__global__ __launch_bounds__(256) void k_domel_log(const double* __restrict__ smag, const BinRow* __restrict__ dom, double* __restrict__ out,
                                                   int K, int H, int staged = 1) {
    extern __shared__ double m_sh[];                      // [H + 1]
    const int64_t f = blockIdx.x;
    const double* m = smag + f * (H + 1);
    if (staged) {
        for (int k = threadIdx.x; k <= H; k += blockDim.x) m_sh[k] = m[k];
        __syncthreads();
    }
    const double* src = staged ? m_sh : m;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const BinRow r = dom[i];
        double2 o;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            double v = apply_row(r, src + c);  // channel 1 is the same spectrum one bin up: |S[N-1-j]| = |S[j+1]|
            if (v < 1e-5) v = 1e-5;
            (c ? o.y : o.x) = log(v);
        }
        reinterpret_cast<double2*>(out)[f * K + i] = o;
    }
}


// =============================================================================================== package phase (fp64)
// (*Phase).ToPhase, phase/phase.go:41-72: gossp STFT, then per bin j < N/2 the pair (Im S[j+1], Re S[N-1-j]) -- for a real
// frame Re S[N-1-j] = Re S[j+1] -- and `shrink` to the first NumFreqs pairs of every frame (phase/impl.go:383-391).
template <int N>
__global__ __launch_bounds__(128) void k_phase_stft(const double* __restrict__ wave, double* __restrict__ out, int64_t n_frames, int hop, int nf,
                                                    DevTables tb) {
    typedef Geo<N> G;
    constexpr int STREAMS = 128 / G::T;
    __shared__ v2d lds[STREAMS * G::LDS_SLOTS];
    const int stream = threadIdx.x / G::T, t = threadIdx.x % G::T;
    int64_t frame = (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < n_frames;
    if (!live) frame = n_frames - 1;
    v2d* my = lds + stream * G::LDS_SLOTS;
    const Twiddles<v2d> tw{tb.tw1d, tb.tw2d, tb.wnd};
    auto sync = [] { __syncthreads(); };
    const double* xs = wave + frame * hop;
    v2d v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int p = 2 * (t + G::T * j);
        v[j] = v2d{xs[p] * tb.windowd[p], xs[p + 1] * tb.windowd[p + 1]};
    }
    fft_forward<G>(v, my, t, tw, sync);
    if (!live) return;
    v2d* o = reinterpret_cast<v2d*>(out) + frame * nf;
    const bool is0 = (t == 0);
    const v2d dc = v[0];
    lane0_permute<G>(v, is0);
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int k = slot_bin<G>(t, s);
        const v2d a = v[s], b = v[15 - s], wk = tb.wnd[k];
        const v2d bc = v2d{b.x, -b.y};
        const v2d p = a + bc, d = a - bc;
        const v2d q = cmul(wk, v2d{d.y, -d.x});
        const v2d xk = p + q, xc = p - q;              // 2 S[k], 2 conj(S[C-k])
        if (k - 1 < nf) o[k - 1] = v2d{0.5 * xk.y, 0.5 * xk.x};
        if (G::C - k - 1 < nf) o[G::C - k - 1] = v2d{-0.5 * xc.y, 0.5 * xc.x};
    }
    if (is0 && G::C - 1 < nf) o[G::C - 1] = v2d{0.0, dc.x - dc.y};   // bin N/2 is real
}

// (*Phase).FromPhase up to the overlap-add, phase/phase.go:137-147 + 74-94 + the IFFT of ISTFT (:104): `grow` repeats a frame's
// last pair up to N/2 bins (phase/impl.go:392-404), bin k = j+1 gets realm0 + i realn1, bin 0 stays 0, bin N/2 is real.
template <int N>
__global__ __launch_bounds__(128) void k_phase_istft(const double* __restrict__ pairs, double* __restrict__ yw, int64_t n_frames, int nf,
                                                     DevTables tb) {
    typedef Geo<N> G;
    constexpr int STREAMS = 128 / G::T;
    __shared__ v2d lds[STREAMS * G::LDS_SLOTS];
    const int stream = threadIdx.x / G::T, t = threadIdx.x % G::T;
    int64_t frame = (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < n_frames;
    if (!live) frame = n_frames - 1;
    v2d* my = lds + stream * G::LDS_SLOTS;
    const Twiddles<v2d> tw{tb.tw1d, tb.tw2d, tb.wnd};
    auto sync = [] { __syncthreads(); };
    const v2d* in = reinterpret_cast<const v2d*>(pairs) + frame * nf;
    const bool is0 = (t == 0);
    const double sc = 1.0 / (double)N;
    auto bin = [&](int kk) {                         // spectrum value of bin kk in [1, C-1] (or the real part for kk = C)
        const int j = kk - 1;
        const v2d pr = in[j < nf ? j : nf - 1];       // (realn1, realm0)
        return v2d{pr.y * sc, pr.x * sc};
    };
    v2d v[16];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int k = slot_bin<G>(t, s);
        const v2d u = bin(k), y2 = bin(G::C - k), wk = tb.wnd[k];
        const v2d vv = v2d{y2.x, -y2.y};
        const v2d e = u + vv, o = cmulc(u - vv, wk);
        v[s] = v2d{e.x - o.y, e.y + o.x};
        v[15 - s] = v2d{e.x + o.y, o.x - e.y};
    }
    v2d dc = v2d{0.0, 0.0};
    if (is0) { const double yn = bin(G::C).x; dc = v2d{yn, -yn}; }   // Y[0] = 0, Y[N/2] = realm0
    lane0_unpermute<G>(v, is0, dc);
    fft_inverse<G>(v, my, t, tw, sync);
    if (live) {
        v2d* ys = reinterpret_cast<v2d*>(yw + frame * N);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int p = 2 * (t + G::T * j);
            ys[t + G::T * j] = v2d{v[j].x * tb.windowd[p], v[j].y * tb.windowd[p + 1]};
        }
    }
}

// Overlap-add with the window-square normalisation of phase.ISTFT (phase/phase.go:103-131): divide where the window sum
// exceeds `thr` (half its maximum), fade proportionally below, leave samples with no window energy; then VolumeBoost.
__global__ __launch_bounds__(256) void k_phase_ola(const double* __restrict__ yw, double* __restrict__ out, int64_t len, int frames, int N, int hop,
                                                   const double* __restrict__ window, double thr, double boost) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < len; n += (int64_t)gridDim.x * blockDim.x) {
        int64_t k0 = n >= N ? (n - N) / hop + 1 : 0;
        int64_t k1 = n / hop;
        if (k1 > frames - 1) k1 = frames - 1;
        double acc = 0.0, ws = 0.0;
        for (int64_t k = k0; k <= k1; k++) {
            const int j = (int)(n - k * hop);
            acc += yw[k * N + j];
            ws += window[j] * window[j];
        }
        if (ws > thr) acc /= ws;
        else if (ws > 1e-21) acc = acc / ws * (ws / thr);
        if (boost != 0.0) acc *= boost;
        out[n] = acc;
    }
}

// =============================================================================================== ISTFT's side effect on its argument
// mel/mel.go:98-108: every iteration overwrites the caller's spectrogram[i][j] with Rect(Abs(spectrogram[i][j]), Phase(S[j])) and
// conjugate-fills the upper half, so after the call it holds the LAST iteration's phase estimate (of the signal BEFORE the last
// update) on the caller's magnitudes.  This kernel recomputes that from x_{I-1}: frame * Hann -> real FFT -> unit phasor (the
// reference's Phase(0) = 0 rule) times |magnitude| as the caller gave it (fp64), bins 0 ... N/2; the shim mirrors the rest.
template <class V> struct TabOf;
template <> struct TabOf<v2f> {
    typedef float S;
    static GL_DEV Twiddles<v2f> tw(const DevTables& tb) { return Twiddles<v2f>{tb.tw1, tb.tw2, tb.wn}; }
    static GL_DEV const v2f* win(const DevTables& tb) { return reinterpret_cast<const v2f*>(tb.window); }
};
template <> struct TabOf<v2d> {
    typedef double S;
    static GL_DEV Twiddles<v2d> tw(const DevTables& tb) { return Twiddles<v2d>{tb.tw1d, tb.tw2d, tb.wnd}; }
    static GL_DEV const v2d* win(const DevTables& tb) { return reinterpret_cast<const v2d*>(tb.windowd); }
};

template <int N, class V>
__global__ __launch_bounds__(128) void k_spec_writeback(const typename TabOf<V>::S* __restrict__ x, const int64_t* __restrict__ frame_xoff,
                                                        const double* __restrict__ mags, double* __restrict__ spec, int64_t n_frames, DevTables tb) {
    typedef Geo<N> G;
    typedef typename TabOf<V>::S S;
    constexpr int STREAMS = 128 / G::T;
    __shared__ V lds[STREAMS * G::LDS_SLOTS];
    const int stream = threadIdx.x / G::T, t = threadIdx.x % G::T;
    int64_t frame = (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < n_frames;
    if (!live) frame = n_frames - 1;
    V* my = lds + stream * G::LDS_SLOTS;
    const Twiddles<V> tw = TabOf<V>::tw(tb);
    auto sync = [] { __syncthreads(); };
    typedef V Va __attribute__((aligned(sizeof(S))));
    const Va* xs = reinterpret_cast<const Va*>(x + frame_xoff[frame]);
    const V* w2 = TabOf<V>::win(tb);
    V v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = xs[t + G::T * j] * w2[t + G::T * j];
    fft_forward<G>(v, my, t, tw, sync);
    if (!live) return;
    const double* m = mags + frame * (G::C + 1);
    v2d* out = reinterpret_cast<v2d*>(spec) + frame * (G::C + 1);
    auto put = [&](int k, double re, double im) {       // Rect(|m[k]|, atan2(im, re)) with Phase(0) = 0
        const double a = fabs(m[k]), big = fmax(fabs(re), fabs(im));
        if (!(big > 0.0)) { out[k] = v2d{a, 0.0}; return; }
        const double sc = 1.0 / big, yr = re * sc, yi = im * sc, r = 1.0 / sqrt(yr * yr + yi * yi);
        out[k] = v2d{a * (yr * r), a * (yi * r)};
    };
    const bool is0 = (t == 0);
    const V dc = v[0];
    lane0_permute<G>(v, is0);
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int k = slot_bin<G>(t, s);
        const V a = v[s], b = v[15 - s], wk = tw.wn[k];
        const V bc = V{b.x, -b.y};
        const V p = a + bc, d = a - bc;
        const V q = cmul(wk, V{d.y, -d.x});
        const V xk = p + q, xc = p - q;                // 2 X[k], 2 conj(X[C-k])
        put(k, (double)xk.x, (double)xk.y);
        put(G::C - k, (double)xc.x, -(double)xc.y);
    }
    if (is0) {   // bins 0 and N/2 are real: phase 0 or pi; Go's Rect(m, pi) = (-m, m sin(pi)) with sin(pi) = 1.2246e-16 in fp64
        const double x0 = (double)dc.x + (double)dc.y, xn = (double)dc.x - (double)dc.y, a0 = fabs(m[0]), an = fabs(m[G::C]);
        out[0] = x0 < 0.0 ? v2d{-a0, a0 * 1.2246467991473532e-16} : v2d{a0, 0.0};
        out[G::C] = xn < 0.0 ? v2d{-an, an * 1.2246467991473532e-16} : v2d{an, 0.0};
    }
}

// =============================================================================================== test hooks: plain rfft / irfft
template <int N>
__global__ __launch_bounds__(128) void k_rfft(const float* __restrict__ frames, float* __restrict__ spec, int64_t n_frames, DevTables tb) {
    typedef Geo<N> G;
    constexpr int STREAMS = 128 / G::T;
    __shared__ v2f lds[STREAMS * G::LDS_SLOTS];
    const int stream = threadIdx.x / G::T, t = threadIdx.x % G::T;
    int64_t frame = (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < n_frames;
    if (!live) frame = n_frames - 1;
    v2f* my = lds + stream * G::LDS_SLOTS;
    const Twiddles<v2f> tw{tb.tw1, tb.tw2, tb.wn};
    auto sync = [] { __syncthreads(); };
    const v2f* xs = reinterpret_cast<const v2f*>(frames + frame * N);
    v2f v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = xs[t + G::T * j];
    fft_forward<G>(v, my, t, tw, sync);
    if (!live) return;
    v2f* out = reinterpret_cast<v2f*>(spec + frame * (N + 2));
    const bool is0 = (t == 0);
    v2f dc = v[0];
    lane0_permute<G>(v, is0);
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int k = slot_bin<G>(t, s);
        v2f a = v[s], b = v[15 - s], wk = tb.wn[k];
        v2f bc = v2f{b.x, -b.y};
        v2f p = a + bc, d = a - bc;
        v2f q = cmul(wk, v2f{d.y, -d.x});
        v2f xk = (p + q) * 0.5f, xc = (p - q) * 0.5f;
        out[k] = xk;
        out[G::C - k] = v2f{xc.x, -xc.y};
    }
    if (is0) { out[0] = v2f{dc.x + dc.y, 0.f}; out[G::C] = v2f{dc.x - dc.y, 0.f}; }
}

template <int N>
__global__ __launch_bounds__(128) void k_irfft(const float* __restrict__ spec, float* __restrict__ frames, int64_t n_frames, DevTables tb) {
    typedef Geo<N> G;
    constexpr int STREAMS = 128 / G::T;
    __shared__ v2f lds[STREAMS * G::LDS_SLOTS];
    const int stream = threadIdx.x / G::T, t = threadIdx.x % G::T;
    int64_t frame = (int64_t)blockIdx.x * STREAMS + stream;
    const bool live = frame < n_frames;
    if (!live) frame = n_frames - 1;
    v2f* my = lds + stream * G::LDS_SLOTS;
    const Twiddles<v2f> tw{tb.tw1, tb.tw2, tb.wn};
    auto sync = [] { __syncthreads(); };
    const v2f* in = reinterpret_cast<const v2f*>(spec + frame * (N + 2));
    const bool is0 = (t == 0);
    const float sc = 1.0f / (float)N;
    v2f v[16];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int k = slot_bin<G>(t, s);
        v2f u = in[k] * sc, y2 = in[G::C - k] * sc, wk = tb.wn[k];
        v2f vv = v2f{y2.x, -y2.y};
        v2f e = u + vv, o = cmulc(u - vv, wk);
        v[s] = v2f{e.x - o.y, e.y + o.x};
        v[15 - s] = v2f{e.x + o.y, o.x - e.y};
    }
    v2f dc = v2f{0.f, 0.f};
    if (is0) { float y0 = in[0].x * sc, yn = in[G::C].x * sc; dc = v2f{y0 + yn, y0 - yn}; }
    lane0_unpermute<G>(v, is0, dc);
    fft_inverse<G>(v, my, t, tw, sync);
    if (live) {
        v2f* ys = reinterpret_cast<v2f*>(frames + frame * N);
#pragma unroll
        for (int j = 0; j < 16; j++) ys[t + G::T * j] = v[j];
    }
}
