// gomel_hip.hip -- libgomel_hip.so: host side and C ABI (include/gomel_hip.h) of gomel's FromMel / ToMel path on
// MI355X (gfx950): table building, batch planning (segments), launches, timing.  Kernels: gl_kernels.hpp; FFT core:
// gl_fft.hpp.  See DESIGN.md for the data layout and the roofline of each kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/gomel_hip.h"
#include "gomel_host.hpp"
#include "gl_kernels.hpp"

// =============================================================================================== host side
static thread_local int g_last_hip = 0;
#define HIPCHK(expr)                                         \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) { g_last_hip = (int)e_; return GOMEL_ERR_HIP; } \
    } while (0)

// hipSetDevice is per-thread state that outlives the call: a process driving several GPUs (torch tensors on cuda:0, a handle on
// device 3) must find its current device unchanged after every entry point.  Entry points switch through this guard.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        if (prev != dev) { err = hipSetDevice(dev); switched = (err == hipSuccess); }
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

enum { kPrecAuto = 0, kPrecFp32 = 32, kPrecFp64 = 64, kPrecSchedule = 1 };

static void hip_pool_hooks(DevPool& p) {   // the allocator behind a handle's device-buffer cache
    p.alloc_fn = [](void** q, size_t n) { return (int)hipMalloc(q, n); };
    p.free_fn = [](void* q) { return (int)hipFree(q); };
    p.forget_error_fn = [] { (void)hipGetLastError(); };
}

struct gomel_handle {
    gomel_handle() { hip_pool_hooks(pool); }
    gomel_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;       // compute (and H2D)
    hipStream_t stream2 = nullptr;      // second compute stream: the chunks of a host-to-host call alternate between the two (lazy)
    hipStream_t copy_stream = nullptr;  // D2H of finished chunks, overlapped with the next chunk's compute
    hipStream_t up_stream = nullptr;    // H2D of the next chunk, overlapped with this chunk's compute
    std::mutex mu;
    std::atomic<int> jobs_outstanding{0};  // submitted jobs not yet waited for: gomel_destroy refuses while there are any
    std::atomic<int> call_mode{0};         // gomel_set_call_mode: 0 = a synchronous batch call is chunked, 1 = it is one whole batch (submit + wait)
    int path_mode = 0;
    // Arithmetic of the Griffin-Lim iterations (gomel_set_precision / gomel_set_precision_schedule): kPrecAuto (default) = the
    // calibrated schedule gomel_auto_fp64_iterations(cfg, I) -- what a caller gets without a knob agrees with the reference's float64
    // arithmetic within 1e-4 RMS; kPrecFp32 = fp32 kernels throughout (the opt-in fast mode); kPrecFp64 = fp64 throughout;
    // kPrecSchedule = the first precise_iters iterations in fp64, the rest in fp32.
    int precision = kPrecAuto;
    int precise_iters = -1;       // kPrecSchedule: iterations in fp64 before the fp32 kernels take over
    int dom_status = GOMEL_OK;    // GOMEL_ERR_TABLE_RANGE when domel would index past the bins (the reference panics in ToMel)
    bool generic = false;         // Resolut outside the register FFT cores: gl_any.hpp kernels (fp64, any batch size)
    bool big = false;             // ... above 8192: gl_big.hpp (every DFT a Bluestein convolution through HBM)
    // persistent small-batch kernel (k_gl_persist): resident-workgroup census, sticky off switch after a timeout, and a pinned
    // slab of timeout words (one per live batch: the kernel's *err comes down with the waveforms, no extra round trip)
    int persist_census[3] = {-1, -1, -1};   // workgroups of k_gl_persist<Resolut, 1 | 2> / k_gl_persist_d<Resolut> the device holds at once (-1: not asked yet)
    // a launch timed out (another context held the CUs): per-iteration kernels for the next persist_backoff batches, then one more
    // try; every further timeout doubles the pause (64 ... 4096 batches), so a transient neighbour costs little and a permanent one
    // costs one 20 ms attempt in thousands of calls
    uint64_t batches_seen = 0, persist_retry_at = 0;
    uint32_t persist_backoff = 64;
    unsigned* err_slab = nullptr; // [kErrSlots + 1] pinned; the last word: a fused kernel's bounded partner wait gave up (DevTables::sync_err)
    unsigned sync_cap = kSyncCapDefault;   // polls such a wait may spend (GOMEL_SYNC_CAP at gomel_create: the tests force a give-up with 0)
    std::vector<int> err_free;
    int persist_nap = 2;          // s_sleep(1) units between two looks at a slot that still shows the sentinel (GOMEL_PERSIST_NAP)
    unsigned persist_ticks = kPersistSpinTicks;   // spin limit in 100 MHz ticks (GOMEL_PERSIST_TICKS: the tests force a timeout with 0)
    std::vector<double> window;   // [N]
    std::vector<BinRow> undo;     // [N/2]
    std::vector<BinRow> dom;      // [K]
    void* d_blob = nullptr;       // one device allocation for all tables
    DevPool pool;                 // working buffers of batches and of the one-shot entry points (hipMalloc / hipFree behind it: hip_pool_hooks)
    DevTables dt{};
};

constexpr int kErrSlots = 256;

// fp64 iterations of a call of I iterations on this handle: 0 = fp32 throughout, >= I = fp64 throughout.  Caller holds h->mu.
static int resolve_fp64_iterations(const gomel_handle* h, int I) {
    if (h->generic) return I;                              // the generic-length kernels are fp64
    switch (h->precision) {
        case kPrecFp32: return 0;
        case kPrecFp64: return I;
        case kPrecSchedule: return std::min(I, std::max(0, h->precise_iters));
        default: return auto_fp64_iterations(h->cfg, I);
    }
}

template <class V> static void fill_twiddles(int N, int R1, int R2, std::vector<V>& tw1, std::vector<V>& tw2, std::vector<V>& wn) {
    typedef typename ScalarOf<V>::type S;
    const int C = N / 2, C1 = C / R1;
    tw1.resize(C); tw2.resize(R2 * 8); wn.resize(C);
    for (int q = 0; q < R1; q++)
        for (int t1 = 0; t1 < C1; t1++) {
            double a = -2.0 * M_PI * (double)((int64_t)t1 * q % C) / (double)C;
            tw1[q * C1 + t1] = V{(S)std::cos(a), (S)std::sin(a)};
        }
    for (int q = 0; q < R2; q++)
        for (int t2 = 0; t2 < 8; t2++) {
            double a = -2.0 * M_PI * (double)(t2 * q % C1) / (double)C1;
            tw2[q * 8 + t2] = V{(S)std::cos(a), (S)std::sin(a)};
        }
    for (int k = 0; k < C; k++) {
        double a = -2.0 * M_PI * (double)k / (double)N;
        wn[k] = V{(S)std::cos(a), (S)std::sin(a)};
    }
}

// Tables of the generic-length path: window (fp64 + fp32), mel rows, radix-2 twiddles of the M-point transform and, when N is
// not a power of two, Bluestein's chirp and the transform of its filter (fp64 on the host, in the order fft_dif leaves its output).
// Frame lengths above 8192 (gl_big.hpp): window, mel rows, the twiddles of the 8192-point LDS transform, Bluestein's chirp and the
// transform of its filter over Mc >= 2N-1 points -- in NATURAL order and pre-divided by Mc (fp64 on the host).
static int upload_tables_big(gomel_handle* h) {
    const int N = h->cfg.resolut, C = N / 2, K = h->cfg.num_mels;
    int Mc = glbig::kSub * 4, logMc = glbig::kLogSub + 2;
    while (Mc < 2 * N - 1) { Mc <<= 1; logMc++; }
    std::vector<v2d> tw((size_t)glbig::kSub / 2), chirp((size_t)N), b((size_t)Mc, v2d{0.0, 0.0}), twc((size_t)Mc / 2), bhat((size_t)Mc);
    for (int j = 0; j < glbig::kSub / 2; j++) { const double a = -2.0 * M_PI * (double)j / (double)glbig::kSub; tw[j] = v2d{std::cos(a), std::sin(a)}; }
    for (int j = 0; j < Mc / 2; j++) { const double a = -2.0 * M_PI * (double)j / (double)Mc; twc[j] = v2d{std::cos(a), std::sin(a)}; }
    auto w = [&](int64_t n) { const double a = -M_PI * (double)((n * n) % (2 * (int64_t)N)) / (double)N; return v2d{std::cos(a), std::sin(a)}; };   // e^{-i pi n^2 / N}
    for (int n = 0; n < N; n++) { chirp[n] = w(n); b[n] = v2d{chirp[n].x, -chirp[n].y}; if (n) b[Mc - n] = b[n]; }
    for (int s = 0; s < logMc; s++) {   // FFT_Mc of b, decimation in frequency: bit-reversed order out
        const int half = Mc >> (s + 1);
        for (int idx = 0; idx < Mc / 2; idx++) {
            const int grp = idx >> (logMc - 1 - s), pos = idx & (half - 1), i0 = (grp << (logMc - s)) + pos, i1 = i0 + half;
            const v2d a = b[i0], c = b[i1], t = twc[(size_t)pos << s], d = v2d{a.x - c.x, a.y - c.y};
            b[i0] = v2d{a.x + c.x, a.y + c.y};
            b[i1] = v2d{d.x * t.x - d.y * t.y, d.x * t.y + d.y * t.x};
        }
    }
    const double sc = 1.0 / (double)Mc;
    for (int k = 0; k < Mc; k++) {
        unsigned rk = 0;
        for (int bit = 0; bit < logMc; bit++) rk |= (((unsigned)k >> bit) & 1u) << (logMc - 1 - bit);
        bhat[k] = v2d{b[rk].x * sc, b[rk].y * sc};
    }
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_wind = carve((size_t)N * 8), o_undo = carve((size_t)C * sizeof(BinRow)), o_dom = carve((size_t)K * sizeof(BinRow)), o_tw = carve(tw.size() * 16),
                 o_ch = carve(chirp.size() * 16), o_bh = carve(bhat.size() * 16);
    std::vector<char> host(off, 0);
    memcpy(&host[o_wind], h->window.data(), (size_t)N * 8);
    memcpy(&host[o_undo], h->undo.data(), (size_t)C * sizeof(BinRow)); memcpy(&host[o_dom], h->dom.data(), (size_t)K * sizeof(BinRow));
    memcpy(&host[o_tw], tw.data(), tw.size() * 16); memcpy(&host[o_ch], chirp.data(), chirp.size() * 16); memcpy(&host[o_bh], bhat.data(), bhat.size() * 16);
    if (!h->d_blob) HIPCHK(hipMalloc(&h->d_blob, off));
    HIPCHK(hipMemcpy(h->d_blob, host.data(), off, hipMemcpyHostToDevice));
    char* d = (char*)h->d_blob;
    h->dt = DevTables{};
    h->dt.windowd = (const double*)(d + o_wind);
    h->dt.undo = (const BinRow*)(d + o_undo); h->dt.dom = (const BinRow*)(d + o_dom);
    h->dt.any.tw = (const v2d*)(d + o_tw); h->dt.any.M = glbig::kSub; h->dt.any.logM = glbig::kLogSub; h->dt.any.bluestein = 0;
    h->dt.big.chirp = (const v2d*)(d + o_ch); h->dt.big.bhat = (const v2d*)(d + o_bh); h->dt.big.Mc = Mc; h->dt.big.R = Mc / glbig::kSub;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&glbig::k_big_fft), hipFuncAttributeMaxDynamicSharedMemorySize, glbig::kSub * 16));
    return GOMEL_OK;
}

// One forward DFT_N per frame of a chunk, as Bluestein's convolution: A (chirped, zero-padded input) -> A = the convolution C, X[k] = chirp[k] C[k].
// Uses B as the intermediate.  nf frames, buffers [nf][Mc].
static void big_convolve(gomel_handle* h, hipStream_t s, v2d* A, v2d* B, int64_t nf) {
    const glbig::BigTables& bt = h->dt.big;
    const unsigned blocks = (unsigned)(nf * bt.R);
    const size_t lds = (size_t)glbig::kSub * 16;
    hipLaunchKernelGGL(glbig::k_big_fft, dim3(blocks), dim3(glany::kAnyThreads), lds, s, (const v2d*)A, B, bt, h->dt.any, bt.bhat, 0, 1);   // conj(FFT(u) . Bhat / Mc)
    hipLaunchKernelGGL(glbig::k_big_fft, dim3(blocks), dim3(glany::kAnyThreads), lds, s, (const v2d*)B, A, bt, h->dt.any, (const v2d*)nullptr, 0, 1);   // conj(FFT(conj V)) = the convolution
}
// Frames a chunk of the big-length path takes at once: two scratch buffers of [frames][Mc] complex doubles, ~1 GiB each at most
static int64_t big_chunk_frames(const gomel_handle* h, int64_t total_frames) {
    const int64_t per = std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)h->dt.big.Mc * 16));
    return std::min<int64_t>(total_frames, per);
}
// mode 0: one Griffin-Lim iteration x -> yw (windowed frames, [total_frames][N]); 1: ISTFT's write-back from x -> spec; 2: ToMel's |S[k]| -> smag
static void big_frames(gomel_handle* h, hipStream_t s, int mode, const double* x, const int64_t* d_frame_xoff, int64_t total_frames, v2d* A, v2d* B,
                       const double* mag, const double* nyq, const double* mid, double* out) {
    const int N = h->cfg.resolut;
    const glbig::BigTables& bt = h->dt.big;
    const int64_t step = big_chunk_frames(h, total_frames);
    const unsigned gx = (unsigned)std::min<int64_t>((bt.Mc + 255) / 256, 64);
    for (int64_t f0 = 0; f0 < total_frames; f0 += step) {
        const int64_t nf = std::min(step, total_frames - f0);
        hipLaunchKernelGGL(glbig::k_big_prep, dim3(gx, (unsigned)nf), dim3(256), 0, s, x, d_frame_xoff, f0, h->dt.windowd, A, N, bt);
        big_convolve(h, s, A, B, nf);
        if (mode == 2) { hipLaunchKernelGGL(glbig::k_big_absmag, dim3(gx, (unsigned)nf), dim3(256), 0, s, (const v2d*)A, f0, out, N, bt); continue; }
        if (mode == 1) { hipLaunchKernelGGL(glbig::k_big_clamp<1>, dim3(gx, (unsigned)nf), dim3(256), 0, s, (const v2d*)A, B, f0, mag, nyq, mid, out, N, bt); continue; }
        hipLaunchKernelGGL(glbig::k_big_clamp<0>, dim3(gx, (unsigned)nf), dim3(256), 0, s, (const v2d*)A, B, f0, mag, nyq, mid, (double*)nullptr, N, bt);
        big_convolve(h, s, B, A, nf);
        hipLaunchKernelGGL(glbig::k_big_finish, dim3(gx, (unsigned)nf), dim3(256), 0, s, (const v2d*)B, f0, h->dt.windowd, out, N, bt);
    }
}

static int upload_tables_generic(gomel_handle* h) {
    if (h->big) return upload_tables_big(h);
    const int N = h->cfg.resolut, C = N / 2, K = h->cfg.num_mels;
    const bool pow2 = (N & (N - 1)) == 0;
    int Mc = 1, logMc = 0;                              // the convolution length of Bluestein: >= 2N - 1
    while (Mc < (pow2 ? N : 2 * N - 1)) { Mc <<= 1; logMc++; }
    const bool split = !pow2 && Mc > 8192;              // 16384 points do not fit a CU's LDS: two 8192-point halves (dft_bluestein_split)
    const int M = split ? Mc / 2 : Mc, logM = split ? logMc - 1 : logMc;   // what the device's FFT routines run on
    std::vector<v2d> tw((size_t)M / 2), chirp(pow2 ? 0 : (size_t)N), bhat(pow2 ? 0 : (size_t)Mc), omega(split ? (size_t)M : 0);
    for (int j = 0; j < M / 2; j++) { const double a = -2.0 * M_PI * (double)j / (double)M; tw[j] = v2d{std::cos(a), std::sin(a)}; }
    for (size_t n = 0; n < omega.size(); n++) { const double a = -M_PI * (double)n / (double)M; omega[n] = v2d{std::cos(a), std::sin(a)}; }
    if (!pow2) {
        auto w = [&](int64_t n) { const double a = -M_PI * (double)((n * n) % (2 * (int64_t)N)) / (double)N; return v2d{std::cos(a), std::sin(a)}; };   // e^{-i pi n^2 / N}
        for (int n = 0; n < N; n++) chirp[n] = w(n);
        std::vector<v2d> b((size_t)Mc, v2d{0.0, 0.0}), twc((size_t)Mc / 2);
        for (int j = 0; j < Mc / 2; j++) { const double a = -2.0 * M_PI * (double)j / (double)Mc; twc[j] = v2d{std::cos(a), std::sin(a)}; }
        for (int n = 0; n < N; n++) { const v2d c = w(n); b[n] = v2d{c.x, -c.y}; if (n) b[Mc - n] = b[n]; }
        // host FFT_Mc of b, decimation in frequency (the device's fft_dif): the result is already in its bit-reversed order
        for (int s = 0; s < logMc; s++) {
            const int half = Mc >> (s + 1);
            for (int idx = 0; idx < Mc / 2; idx++) {
                const int grp = idx >> (logMc - 1 - s), pos = idx & (half - 1), i0 = (grp << (logMc - s)) + pos, i1 = i0 + half;
                const v2d a = b[i0], c = b[i1], t = twc[(size_t)pos << s], d = v2d{a.x - c.x, a.y - c.y};
                b[i0] = v2d{a.x + c.x, a.y + c.y};
                b[i1] = v2d{d.x * t.x - d.y * t.y, d.x * t.y + d.y * t.x};
            }
        }
        if (!split) bhat = b;
        else {
            // b[bitrev_Mc(k)] = Bhat[k].  Bin k = 2r + p (p = 0 even, 1 odd) sits at bitrev_Mc(k) = p * M + bitrev_M(r): the lower half of the
            // bit-reversed array IS the even bins in the order an M-point fft_dif leaves them, the upper half the odd bins likewise.
            bhat = b;
        }
    }
    std::vector<float> wf(N);
    for (int i = 0; i < N; i++) wf[i] = (float)h->window[i];
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_win = carve((size_t)N * 4), o_wind = carve((size_t)N * 8), o_undo = carve((size_t)C * sizeof(BinRow)), o_dom = carve((size_t)K * sizeof(BinRow)),
                 o_tw = carve(tw.size() * 16), o_ch = carve(chirp.size() * 16 + 16), o_bh = carve(bhat.size() * 16 + 16), o_om = carve(omega.size() * 16 + 16);
    std::vector<char> host(off, 0);
    memcpy(&host[o_win], wf.data(), (size_t)N * 4); memcpy(&host[o_wind], h->window.data(), (size_t)N * 8);
    memcpy(&host[o_undo], h->undo.data(), (size_t)C * sizeof(BinRow)); memcpy(&host[o_dom], h->dom.data(), (size_t)K * sizeof(BinRow));
    memcpy(&host[o_tw], tw.data(), tw.size() * 16);
    if (!pow2) { memcpy(&host[o_ch], chirp.data(), chirp.size() * 16); memcpy(&host[o_bh], bhat.data(), bhat.size() * 16); }
    if (split) memcpy(&host[o_om], omega.data(), omega.size() * 16);
    if (!h->d_blob) HIPCHK(hipMalloc(&h->d_blob, off));
    HIPCHK(hipMemcpy(h->d_blob, host.data(), off, hipMemcpyHostToDevice));
    char* d = (char*)h->d_blob;
    h->dt = DevTables{};
    h->dt.window = (const float*)(d + o_win); h->dt.windowd = (const double*)(d + o_wind);
    h->dt.undo = (const BinRow*)(d + o_undo); h->dt.dom = (const BinRow*)(d + o_dom);
    h->dt.any.tw = (const v2d*)(d + o_tw); h->dt.any.chirp = (const v2d*)(d + o_ch); h->dt.any.bhat = (const v2d*)(d + o_bh);
    h->dt.any.omega = (const v2d*)(d + o_om);
    h->dt.any.M = M; h->dt.any.logM = logM; h->dt.any.bluestein = pow2 ? 0 : split ? 2 : 1;
    if ((size_t)M * 16 > 64 * 1024) {   // the frame lives in LDS: up to 128 KiB of the CU's 160
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&glany::k_any_frames<glany::kAnyGl>), hipFuncAttributeMaxDynamicSharedMemorySize, M * 16));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&glany::k_any_frames<glany::kAnyToMel>), hipFuncAttributeMaxDynamicSharedMemorySize, M * 16));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&glany::k_any_frames<glany::kAnySpec>), hipFuncAttributeMaxDynamicSharedMemorySize, M * 16));
    }
    return GOMEL_OK;
}

static int upload_tables(gomel_handle* h) {
    if (h->generic) return upload_tables_generic(h);
    const int N = h->cfg.resolut, C = N / 2, K = h->cfg.num_mels;
    int R1 = 0, R2 = 0;
    GOMEL_DISPATCH_N(N, R1 = Geo<NN>::R1; R2 = Geo<NN>::R2);
    std::vector<v2f> tw1, tw2, wn;
    std::vector<v2d> tw1d, tw2d, wnd;
    fill_twiddles<v2f>(N, R1, R2, tw1, tw2, wn);
    fill_twiddles<v2d>(N, R1, R2, tw1d, tw2d, wnd);
    std::vector<float> wf(N);
    for (int i = 0; i < N; i++) wf[i] = (float)h->window[i];
    // carve one allocation, every piece 256-byte aligned
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    size_t o_win = carve(N * 4), o_tw1 = carve(C * 8), o_tw2 = carve(R2 * 8 * 8), o_wn = carve(C * 8);
    size_t o_tw1d = carve(C * 16), o_tw2d = carve(R2 * 8 * 16), o_wnd = carve(C * 16), o_wind = carve(N * 8);
    size_t o_undo = carve(C * sizeof(BinRow)), o_dom = carve(K * sizeof(BinRow));
    std::vector<char> host(off, 0);
    memcpy(&host[o_win], wf.data(), N * 4);
    memcpy(&host[o_tw1], tw1.data(), C * 8); memcpy(&host[o_tw2], tw2.data(), R2 * 8 * 8); memcpy(&host[o_wn], wn.data(), C * 8);
    memcpy(&host[o_tw1d], tw1d.data(), C * 16); memcpy(&host[o_tw2d], tw2d.data(), R2 * 8 * 16); memcpy(&host[o_wnd], wnd.data(), C * 16);
    memcpy(&host[o_wind], h->window.data(), N * 8);
    memcpy(&host[o_undo], h->undo.data(), C * sizeof(BinRow)); memcpy(&host[o_dom], h->dom.data(), K * sizeof(BinRow));
    if (!h->d_blob) HIPCHK(hipMalloc(&h->d_blob, off));
    HIPCHK(hipMemcpy(h->d_blob, host.data(), off, hipMemcpyHostToDevice));
    char* d = (char*)h->d_blob;
    h->dt.window = (const float*)(d + o_win);
    h->dt.tw1 = (const v2f*)(d + o_tw1); h->dt.tw2 = (const v2f*)(d + o_tw2); h->dt.wn = (const v2f*)(d + o_wn);
    h->dt.tw1d = (const v2d*)(d + o_tw1d); h->dt.tw2d = (const v2d*)(d + o_tw2d); h->dt.wnd = (const v2d*)(d + o_wnd);
    h->dt.windowd = (const double*)(d + o_wind);
    h->dt.undo = (const BinRow*)(d + o_undo); h->dt.dom = (const BinRow*)(d + o_dom);
    h->dt.sync_err = h->err_slab ? h->err_slab + kErrSlots : nullptr;   // (hipHostMalloc memory: the device writes it in place)
    h->dt.sync_cap = h->sync_cap;
    return GOMEL_OK;
}

// A fused kernel's bounded wait for its partner wave gave up since the last look (StreamFlagSync, gl_kernels.hpp): whatever that launch
// wrote is garbage.  Called wherever the host has just synchronised with the handle's work: the call in hand comes back as GOMEL_ERR_HIP
// (gomel_last_hip_error() = hipErrorLaunchTimeOut), the word is cleared and the handle stays usable -- a wave only stops waiting when
// its partner has stopped running, so recomputing is the caller's decision, not something to retry silently.
static int sync_err_check(gomel_handle* h) {
    if (!h->err_slab || !h->err_slab[kErrSlots]) return GOMEL_OK;
    h->err_slab[kErrSlots] = 0;
    g_last_hip = (int)hipErrorLaunchTimeOut;
    return GOMEL_ERR_HIP;
}

static int create_common(const gomel_config* cfg, const void* blob, size_t blob_bytes, gomel_handle** out) {
    if (!cfg || !out) return GOMEL_ERR_NULL;
    *out = nullptr;
    int rc = validate_config(cfg);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOMEL_ERR_NO_DEVICE;
    gomel_handle* h = new (std::nothrow) gomel_handle();
    if (!h) return GOMEL_ERR_NO_MEMORY;
    h->cfg = *cfg;
    h->generic = !gomel_supported_n(cfg->resolut);
    h->big = gomel_big_n(cfg->resolut);
    if (const char* e = getenv("GOMEL_POOL_MB")) h->pool.cap = (size_t)std::max(0L, atol(e)) << 20;
    if (cfg->device >= 0) {
        if (cfg->device >= ndev) { delete h; return GOMEL_ERR_NO_DEVICE; }
        h->device = cfg->device;
    } else if (hipGetDevice(&h->device) != hipSuccess) { delete h; return GOMEL_ERR_NO_DEVICE; }
    const int N = cfg->resolut, C = N / 2, K = cfg->num_mels;
    if (blob) {
        const size_t need = sizeof(TableHeader) + (size_t)N * 8 + ((size_t)C + K) * sizeof(BinRow);
        TableHeader th;
        if (blob_bytes != need) { delete h; return GOMEL_ERR_BAD_TABLES; }
        memcpy(&th, blob, sizeof th);
        if (th.magic != kMagic || (th.version != kTableFormat && th.version != 5 /* what the ABI-5 build stamped: same layout */) || th.n != N || th.hop != cfg->window || th.k != K ||
            th.fmin != cfg->mel_fmin || th.fmax != cfg->mel_fmax) { delete h; return GOMEL_ERR_BAD_TABLES; }
        const char* p = (const char*)blob + sizeof th;
        h->window.resize(N); memcpy(h->window.data(), p, (size_t)N * 8); p += (size_t)N * 8;
        h->undo.resize(C); memcpy(h->undo.data(), p, (size_t)C * sizeof(BinRow)); p += (size_t)C * sizeof(BinRow);
        h->dom.resize(K); memcpy(h->dom.data(), p, (size_t)K * sizeof(BinRow));
        h->cfg.hann_periodic = th.hann_periodic;
        h->dom_status = th.dom_status;
        // The rows become device-side indices (k_mel_to_mag, k_domel_log): a truncated / corrupted broadcast or a blob from a
        // build with other table logic must not reach the GPU.  Same bounds as classify_row enforces when building.
        if (!rows_in_range(h->undo, K, true) || (th.dom_status == GOMEL_OK && !rows_in_range(h->dom, C, false))) { delete h; return GOMEL_ERR_BAD_TABLES; }
        for (double w : h->window) if (!std::isfinite(w)) { delete h; return GOMEL_ERR_BAD_TABLES; }
    } else {
        rc = build_host_tables(*cfg, h->window, h->undo, h->dom, &h->dom_status);
        if (rc) { delete h; return rc; }
    }
    DeviceGuard dg(h->device);
    if (dg.err != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking) != hipSuccess) {
        if (h->stream) hipStreamDestroy(h->stream);
        if (h->copy_stream) hipStreamDestroy(h->copy_stream);
        delete h; return GOMEL_ERR_HIP;
    }
    if (hipHostMalloc((void**)&h->err_slab, (kErrSlots + 1) * sizeof(unsigned), hipHostMallocDefault) == hipSuccess) {
        memset(h->err_slab, 0, (kErrSlots + 1) * sizeof(unsigned));
        for (int i = kErrSlots - 1; i >= 0; i--) h->err_free.push_back(i);
    } else { h->err_slab = nullptr; (void)hipGetLastError(); }   // no slab: the persistent path is simply never chosen
    // GOMEL_PRECISION = auto | 32 | 64: the mode a handle STARTS in (default auto), for deployments that choose the fast mode -- or
    // fp64 throughout -- without touching the host program; gomel_set_precision / gomel_set_precision_schedule override it
    if (const char* e = getenv("GOMEL_PRECISION")) h->precision = !strcmp(e, "32") ? kPrecFp32 : !strcmp(e, "64") ? kPrecFp64 : kPrecAuto;
    if (const char* e = getenv("GOMEL_PERSIST_TICKS")) h->persist_ticks = (unsigned)strtoul(e, nullptr, 10);
    if (const char* e = getenv("GOMEL_PERSIST_NAP")) h->persist_nap = atoi(e);
    if (const char* e = getenv("GOMEL_SYNC_CAP")) h->sync_cap = (unsigned)strtoul(e, nullptr, 10);
    rc = upload_tables(h);
    if (rc) { if (h->err_slab) hipHostFree(h->err_slab); hipStreamDestroy(h->stream); hipStreamDestroy(h->copy_stream); hipStreamDestroy(h->up_stream); if (h->d_blob) hipFree(h->d_blob); delete h; return rc; }
    *out = h;
    return GOMEL_OK;
}

// Pins caller-owned host buffers for the duration of a call (hipHostRegister: ~8 ms per GiB, then PCIe runs at full rate; a
// buffer the driver has never seen copies at a tenth of that).  The runtime does not reference-count registrations, and the
// same buffer (or a view into it) may be passed for several utterances or by several threads / handles at once, so every range
// pinned by this library goes through one process-wide reference-counted INTERVAL registry: a request that lies inside a live
// registration takes a reference on it (so the owner cannot unregister under a DMA that is still in flight).  A refused
// registration (range pinned by somebody else, partial overlap, memory that cannot be locked, ...) leaves that buffer pageable:
// the runtime then stages the copy itself.  gomel_host_register / gomel_host_alloc (below) let a caller that reuses its
// buffers pay the registration once instead of per call.
static bool hip_pin_foreign(void* p, uintptr_t* base, size_t* size) {
    hipPointerAttribute_t at;
    *base = 0; *size = 0;
    if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type != hipMemoryTypeUnregistered) {      // device / pinned by the caller
        hipDeviceptr_t b = nullptr; size_t n = 0;
        if (hipMemGetAddressRange(&b, &n, (hipDeviceptr_t)p) == hipSuccess && b && n) { *base = (uintptr_t)b; *size = n; }
        (void)hipGetLastError();
        return true;
    }
    (void)hipGetLastError();
    return false;
}
static bool hip_pin_reg(void* p, size_t bytes) {
    const bool ok = hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();   // a refusal must not linger as the thread's "last error" (launches are checked with it)
    return ok;
}
static void hip_pin_unreg(void* p) { (void)hipHostUnregister(p); (void)hipGetLastError(); }
static const PinBackend kHipPins = {hip_pin_foreign, hip_pin_reg, hip_pin_unreg};
static const bool g_hip_pins_installed = (g_pins.be = g_pins.product_be = &kHipPins, true);   // at load time: the product's backend

// The buffers of ONE call.  want() only records (base, longest extent asked for) -- prefix views buf[:a], buf[:b] of one base
// become one registration of max(a, b) bytes -- and commit() registers.  The destructor first drains the streams the call
// copied on (every early error return still has copies queued), then drops the references.
struct PinnedScope : PinWants {
    hipStream_t drain[2] = {nullptr, nullptr}, extra = nullptr;
    void commit(hipStream_t s0 = nullptr, hipStream_t s1 = nullptr) {
        drain[0] = s0; drain[1] = s1;
        PinWants::commit();
    }
    void drain_too(hipStream_t s) { extra = s; }
    ~PinnedScope() {
        if (held.empty()) return;
        for (hipStream_t s : {drain[0], drain[1], extra}) if (s) { (void)hipStreamSynchronize(s); (void)hipGetLastError(); }
        release_all();
    }
};

// ----------------------------------------------------------------------------------------------- batch
struct gomel_batch {
    gomel_handle* h = nullptr;
    size_t n_utt = 0;
    std::vector<UttDesc> utts;
    std::vector<size_t> n_pairs;
    int64_t total_frames = 0, total_len = 0, total_pairs = 0, max_len = 0;
    bool any_x0 = false, uploaded = false;
    int fast = 0;
    int direct = 0;            // gomel_istft: d_mel holds linear magnitudes [total_frames][N/2+1] instead of log-mel pairs
    int iters = -1;            // >= 0: overrides cfg.griffin_lim_iterations (ISTFT's numIterations argument)
    int io32 = 0;              // the caller's buffers (log-mel, x0, waveforms) are float instead of double (gomel_from_mel_batch_f32)
    size_t esz = 8;            // bytes per element of d_mel / d_x0 / d_out
    int want_spec = 0;         // gomel_istft_spec: keep x_{I-1} reachable (generic kernels) and fill d_spec after the last iteration
    // device
    void* d_mel = nullptr;     // [total_pairs][2] (or, direct: [total_frames][N/2+1]) of esz-byte elements
    void* d_x0 = nullptr;      // [total_len] (only when some x0 is given)
    void* d_out = nullptr;     // [total_len]
    double* d_spec = nullptr;  // [total_frames][N/2+1][2]: ISTFT's side effect on its spectrogram argument (mel/mel.go:98-108)
    float* d_mag = nullptr;    // [total_frames][N/2]
    float* d_nyq = nullptr;    // [total_frames]
    float* d_x[2] = {nullptr, nullptr};  // ping-pong signal
    float* d_yw = nullptr;     // generic path: [total_frames][N]
    int precise = 0;           // fp64 mode: the four buffers below replace d_mag / d_nyq / d_x / d_yw
    hipStream_t rs = nullptr;  // the stream this batch's kernels run on (null: the handle's compute stream)
    int mixed_k = 0;           // > 0: precision schedule on the fused path -- the first mixed_k iterations k_gl_fused_d, the rest k_gl_fused (both buffer sets exist)
    double* d_magd = nullptr;
    double* d_nyqd = nullptr;
    double* d_xd[2] = {nullptr, nullptr};
    double* d_ywd = nullptr;
    v2d* d_bigA = nullptr;      // frame lengths above 8192: two scratch buffers of [chunk frames][Mc] complex doubles (gl_big.hpp)
    v2d* d_bigB = nullptr;
    double* d_midd = nullptr;   // exported ISTFT at an odd frame length: |spectrogram[f][N/2]| (bin N/2+1's magnitude sits in d_nyqd)
    int64_t* d_frame_xoff = nullptr;
    UttDesc* d_utts = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> dom_ev;  // 2 per iteration, created on the first timed run
    // fused path
    std::vector<SegDesc> segs;
    SegDesc* d_segs = nullptr;
    float* d_tail[2] = {nullptr, nullptr};  // [n_segs][N - hop]
    double* d_taild[2] = {nullptr, nullptr}; // the same for the fp64 fused kernel
    int g_max = 0;
    // persistent path (k_gl_persist): d_yw holds its three output buffers, d_ctl = {err, pad[3]}
    int persist = 0;
    int pmix_k = 0, pmix_occ = 0;   // > 0: precision schedule on the persistent path -- k_gl_persist_d for the first pmix_k iterations, then k_gl_persist<.., pmix_occ, ..> from its result
    int n_wg = 0;
    std::vector<std::pair<int64_t, int64_t>> pgroups;   // k_gl_persist_d only: the frame ranges (whole utterances) of its launches -- one, or two for a batch of up to twice the chip's capacity
    int err_slot = -1;
    int persist_units = 0;     // share of the device this batch's queued persistent launch is accounted with (persist_take)
    hipEvent_t gate_ev = nullptr;   // recorded behind every run of this batch: how the device gate sees its work finish
    int32_t* d_frame_utt = nullptr;
    unsigned* d_ctl = nullptr;
    size_t ctl_bytes = 0;
};

static inline hipStream_t run_stream(const gomel_batch* b) { return b->rs ? b->rs : b->h->stream; }
static void persist_release(gomel_batch* b);
static void gate_forget(gomel_batch* b);
static void batch_free(gomel_batch* b) {
    if (!b) return;
    DeviceGuard dg(b->h->device);
    gate_forget(b);
    if (b->gate_ev) hipEventDestroy(b->gate_ev);
    void* ptrs[] = {b->d_spec, b->d_mel, b->d_x0, b->d_out, b->d_mag, b->d_nyq, b->d_x[0], b->d_x[1], b->d_yw, b->d_frame_xoff, b->d_utts,
                    b->d_segs, b->d_tail[0], b->d_tail[1], b->d_taild[0], b->d_taild[1], b->d_magd, b->d_nyqd, b->d_midd, b->d_xd[0], b->d_xd[1], b->d_ywd, b->d_bigA, b->d_bigB,
                    b->d_frame_utt, b->d_ctl};
    for (void* p : ptrs) if (p) b->h->pool.release(p);
    persist_release(b);
    if (b->err_slot >= 0) b->h->err_free.push_back(b->err_slot);
    for (auto& e : b->ev) if (e) hipEventDestroy(e);
    for (auto& e : b->dom_ev) if (e) hipEventDestroy(e);
    delete b;
}

struct BatchGuard {   // frees a half-built batch on every early exit, exceptions included
    gomel_batch* b;
    ~BatchGuard() { if (b) batch_free(b); }
};

template <int N, int OCC> static int persist_census_n(int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gl_persist<N, OCC, (OCC == 1 ? 4 : 2)>, 128, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return per_cu * cus;
}
template <int N> static int persist_census_d(int device) {
    int per_cu = 0, cus = 0;
    constexpr size_t lds = persist_d_lds_bytes<N>();
    if ((lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gl_persist_d<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gl_persist_d<N>, 128, lds) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return per_cu * cus;
}
static int persist_census(gomel_handle* h, int occ) {   // occ 1 / 2: the fp32 kernel's builds; 3: the fp64 kernel.  Caller holds a DeviceGuard
    int& c = h->persist_census[occ - 1];
    if (c < 0) {
        int n = 0;
        if (occ == 1) { GOMEL_DISPATCH_N(h->cfg.resolut, n = (persist_census_n<NN, 1>(h->device))); }
        else if (occ == 2) { GOMEL_DISPATCH_N(h->cfg.resolut, n = (persist_census_n<NN, 2>(h->device))); }
        else { GOMEL_DISPATCH_N(h->cfg.resolut, n = persist_census_d<NN>(h->device)); }
        c = n;
    }
    return c;
}

template <int N, int SH> static void launch_fused_t(gomel_batch* b, int cur) {
    typedef Geo<N> G;
    constexpr int STREAMS = kFusedThreads / G::T;
    const int n = (int)b->segs.size();
    hipLaunchKernelGGL((k_gl_fused<N, SH>), dim3((unsigned)((n + STREAMS - 1) / STREAMS)), dim3(kFusedThreads), 0, run_stream(b), b->d_x[cur], b->d_tail[cur],
                       b->d_x[cur ^ 1], b->d_tail[cur ^ 1], b->d_segs, n, b->g_max, b->d_mag, b->d_nyq, b->h->dt);
}
template <int N> static void launch_fused_n(gomel_batch* b, int cur, int sh) {
    switch (sh) {
        case 1: launch_fused_t<N, 1>(b, cur); break;
        case 2: launch_fused_t<N, 2>(b, cur); break;
        case 4: launch_fused_t<N, 4>(b, cur); break;
        case 5: launch_fused_t<N, 5>(b, cur); break;
        default: launch_fused_t<N, 8>(b, cur); break;
    }
}
template <int N, int SH> static hipError_t launch_fused_d_t(gomel_batch* b, int cur) {
    typedef Geo<N> G;
    constexpr int STREAMS = kFusedThreads / G::T;
    constexpr size_t lds = fused_d_lds_bytes<N>();
    if (lds > 64 * 1024) {   // per device and cheap: set on every launch rather than remembered per process
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gl_fused_d<N, SH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int n = (int)b->segs.size();
    hipLaunchKernelGGL((k_gl_fused_d<N, SH>), dim3((unsigned)((n + STREAMS - 1) / STREAMS)), dim3(kFusedThreads), lds, run_stream(b), b->d_xd[cur], b->d_taild[cur],
                       b->d_xd[cur ^ 1], b->d_taild[cur ^ 1], b->d_segs, n, b->g_max, b->d_magd, b->d_nyqd, b->h->dt);
    return hipSuccess;
}
template <int N> static hipError_t launch_fused_d_n(gomel_batch* b, int cur, int sh) {
    switch (sh) {
        case 1: return launch_fused_d_t<N, 1>(b, cur);
        case 2: return launch_fused_d_t<N, 2>(b, cur);
        case 4: return launch_fused_d_t<N, 4>(b, cur);
        case 5: return launch_fused_d_t<N, 5>(b, cur);
        default: return launch_fused_d_t<N, 8>(b, cur);
    }
}
// the any-hop form in fp64 (k_gl_fused_any_d) + the fold of the tails into x, in place (as launch_fused_any)
template <int N> static hipError_t launch_fused_any_d(gomel_batch* b, int cur) {
    typedef Geo<N> G;
    constexpr int STREAMS = kFusedThreads / G::T;
    constexpr size_t lds = fused_d_lds_bytes<N>();
    const int n = (int)b->segs.size();
    const int hop = b->h->cfg.window;
    const int rows_new = std::max(16 - ((N - hop) / 2) / G::T, ((hop - 1) / 2) / G::T + 1);
    hipError_t e = hipSuccess;
#define GOMEL_ANY_D_LAUNCH(NR, ODD)                                                                                                                                  \
    do {                                                                                                                                                             \
        if (lds > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gl_fused_any_d<N, NR, ODD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess)                                                                                                                                          \
            hipLaunchKernelGGL((k_gl_fused_any_d<N, NR, ODD>), dim3((unsigned)((n + STREAMS - 1) / STREAMS)), dim3(kFusedThreads), lds, run_stream(b), b->d_xd[cur],  \
                               b->d_xd[cur ^ 1], b->d_taild[0], b->d_segs, n, b->g_max, b->d_magd, b->d_nyqd, hop, b->h->dt);                                        \
    } while (0)
    if (hop & 1) { if (rows_new <= 8) GOMEL_ANY_D_LAUNCH(8, true); else GOMEL_ANY_D_LAUNCH(16, true); }
    else { if (rows_new <= 8) GOMEL_ANY_D_LAUNCH(8, false); else GOMEL_ANY_D_LAUNCH(16, false); }
#undef GOMEL_ANY_D_LAUNCH
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_add_tails_d, dim3((unsigned)n), dim3(256), 0, run_stream(b), b->d_segs, b->d_taild[0], b->d_xd[cur ^ 1], N - hop, hop);
    return hipSuccess;
}
static hipError_t launch_fused_d(gomel_batch* b, int cur) {
    const int N = b->h->cfg.resolut, sh = fused_shift(b->h->cfg);
    hipError_t e = hipSuccess;
    if (b->fast == 2) { GOMEL_DISPATCH_N(N, e = launch_fused_any_d<NN>(b, cur)); return e; }
    GOMEL_DISPATCH_N(N, e = launch_fused_d_n<NN>(b, cur, sh));
    return e;
}
template <int N> static void launch_fused_any(gomel_batch* b, int cur) {
    typedef Geo<N> G;
    constexpr int STREAMS = kFusedThreads / G::T;
    const int n = (int)b->segs.size();
    // rows of the window that take new samples (the last ones of the 16: pairs from floor((N - hop) / 2) on) and rows that hold
    // finished samples (the first ones: pairs up to (hop - 1) / 2): NR must cover both
    const int hop = b->h->cfg.window;
    const int rows_new = std::max(16 - ((N - hop) / 2) / G::T, ((hop - 1) / 2) / G::T + 1);
#define GOMEL_ANY_LAUNCH(NR, ODD)                                                                                                                       \
    hipLaunchKernelGGL((k_gl_fused_any<N, NR, ODD>), dim3((unsigned)((n + STREAMS - 1) / STREAMS)), dim3(kFusedThreads), 0, run_stream(b), b->d_x[cur], \
                       b->d_x[cur ^ 1], b->d_tail[0], b->d_segs, n, b->g_max, b->d_mag, b->d_nyq, hop, b->h->dt)
    if (hop & 1) { if (rows_new <= 4) GOMEL_ANY_LAUNCH(4, true); else if (rows_new <= 8) GOMEL_ANY_LAUNCH(8, true); else GOMEL_ANY_LAUNCH(16, true); }
    else { if (rows_new <= 4) GOMEL_ANY_LAUNCH(4, false); else if (rows_new <= 8) GOMEL_ANY_LAUNCH(8, false); else GOMEL_ANY_LAUNCH(16, false); }
#undef GOMEL_ANY_LAUNCH
    // fold what the segments added past their own samples into x, in place: the next iteration (and k_finalize) read x alone
    hipLaunchKernelGGL(k_add_tails<float>, dim3((unsigned)n), dim3(256), 0, run_stream(b), b->d_segs, b->d_tail[0], b->d_x[cur ^ 1], N - b->h->cfg.window, b->h->cfg.window);
}
static void launch_fused(gomel_batch* b, int cur) {
    const int N = b->h->cfg.resolut, sh = fused_shift(b->h->cfg);
    if (b->fast == 2) { GOMEL_DISPATCH_N(N, launch_fused_any<NN>(b, cur)); return; }
    GOMEL_DISPATCH_N(N, launch_fused_n<NN>(b, cur, sh));
}

template <int N> static void launch_frames(gomel_batch* b, const float* x) {
    typedef Geo<N> G;
    constexpr int STREAMS = 128 / G::T;
    const int64_t blocks = (b->total_frames + STREAMS - 1) / STREAMS;
    hipLaunchKernelGGL(k_gl_frames<N>, dim3((unsigned)blocks), dim3(128), 0, run_stream(b), x, b->d_frame_xoff, b->d_mag, b->d_nyq, b->d_yw,
                       b->total_frames, b->h->dt);
}

static int batch_run_impl(gomel_batch* b, gomel_timing* tm);

// THE DEVICE GATE.  A persistent launch (k_gl_persist) needs every one of its workgroups resident at once -- a 512-register workgroup
// wants a whole SIMD to itself -- and waits inside the launch for its neighbours.  Two things of THIS process can keep it from that
// (another process on the same GPU is what the kernel's own time limit is for):
//   * other persistent launches: two half-resident ones would wait for each other until both give up.  Launches are accounted in
//     1/1024ths of the device (n_wg against the census of the build in use) from the moment they are queued until the host has seen
//     them finish (sync / download / destroy); a run that finds no room goes through the per-iteration kernels (same bits);
//   * ordinary kernels of other handles (per-iteration kernels, fused batches, ToMel): a stream of them from three threads kept one
//     workgroup of a persistent launch from its SIMD beyond the 20 ms spin limit in round 3 (tests/test_gpu_persist.py, the
//     four-handle test).  So a batch takes the persistent path only while NO other kernel work of this process is in flight on the
//     device -- counted from the moment its enqueueing begins until the event recorded behind it has completed -- and whatever is
//     enqueued while persistent launches are in flight first makes its stream wait for them (hipStreamWaitEvent: <= 0.6 ms of
//     device-side ordering, nothing blocks on the host).  The gate's mutex is held while a persistent launch is being enqueued
//     (~0.1 ms), so that "nothing else in flight" still holds when its kernel reaches the queue.
struct HipEvOps {
    typedef hipEvent_t Event;
    static int query(hipEvent_t ev) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipErrorNotReady) return 1;
        if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
        return 0;
    }
};
typedef DeviceGateT<HipEvOps> DeviceGate;
static DeviceGate g_gate[64];
static inline DeviceGate& gate_of(const gomel_handle* h) { return g_gate[h->device & 63]; }

// Ordinary (non-persistent) kernel work on `s`: enter before the first launch, leave behind the last one.  leave(ev): ev has just been
// recorded behind the work; leave(nullptr): the work has been synchronised with, nothing stays in flight.
struct GateOrdinary {
    DeviceGate* g = nullptr;
    void enter(const gomel_handle* h, hipStream_t s) {
        g = &gate_of(h);
        g->enter_ordinary([s](hipEvent_t ev) { if (hipStreamWaitEvent(s, ev, 0) != hipSuccess) (void)hipGetLastError(); });
    }
    void leave(hipEvent_t ev, const gomel_batch* owner) {
        if (!g) return;
        g->leave_ordinary(ev ? &ev : nullptr, owner);
        g = nullptr;
    }
    ~GateOrdinary() { leave(nullptr, nullptr); }   // an early error return: whatever was queued is drained by the caller's error path
};

static int persist_units(gomel_batch* b) {
    const int census = std::max(1, persist_census(b->h, b->persist));
    return (int)std::min<int64_t>(1024, ((int64_t)b->n_wg * 1024 + census - 1) / census);
}
// A persistent run: takes the gate (DeviceGateT::try_take_persistent keeps its mutex) when the device is free of other work and has
// room; `locked` then stays set until leave() has recorded the launch.  Caller holds h->mu and a DeviceGuard.
struct GatePersist {
    DeviceGate* g = nullptr;
    bool locked = false;
    bool take(gomel_batch* b) {
        g = &gate_of(b->h);
        const int u = b->persist_units ? 0 : persist_units(b);            // an earlier run of this batch still accounted: same stream, same share
        if (!g->try_take_persistent(u, b)) { g = nullptr; return false; }
        if (u) b->persist_units = u;
        locked = true;
        return true;
    }
    void leave(gomel_batch* b, hipEvent_t ev) {   // ev recorded behind the launch (null: nothing was launched after all)
        if (!locked) return;
        locked = false;
        g->record_persistent(ev ? &ev : nullptr, b);
    }
    ~GatePersist() { if (locked) { locked = false; g->record_persistent(nullptr, nullptr); } }
};
static void persist_release(gomel_batch* b) {   // the batch's launches are known to have finished (or never ran)
    if (!b->persist_units) return;
    gate_of(b->h).release_persistent(b->persist_units, b);
    b->persist_units = 0;
}
static void gate_forget(gomel_batch* b) { gate_of(b->h).forget(b); }   // the batch (and its event) is going away

// A persistent launch gave up (k_gl_persist: a workgroup waited 20 ms for a neighbour -- some other context held the CUs it
// needed): nothing of its output is used.  x0 is still in d_x[0] (the kernel only reads it), so the batch is recomputed with the
// per-iteration kernels, and the handle pauses the persistent path (persist_backoff batches).  Caller holds h->mu and has synchronised the stream.
static int persist_redo(gomel_batch* b, gomel_timing* tm) {
    gomel_handle* h = b->h;
    h->persist_retry_at = h->batches_seen + h->persist_backoff;
    h->persist_backoff = std::min<uint32_t>(h->persist_backoff * 2, 4096);
    h->err_slab[b->err_slot] = 0;
    persist_release(b);
    b->persist = 0;
    int rc = batch_run_impl(b, tm);
    if (!rc && hipStreamSynchronize(run_stream(b)) != hipSuccess) rc = GOMEL_ERR_HIP;
    return rc;
}
// after the stream has drained: did the batch's persistent launch complete?
static int persist_verify(gomel_batch* b) {
    persist_release(b);
    if (!b->persist) return GOMEL_OK;
    if (!b->h->err_slab[b->err_slot]) { b->h->persist_backoff = 64; return GOMEL_OK; }   // a launch that ran through ends the escalation: the next timeout pauses for 64 batches again
    return persist_redo(b, nullptr);
}

static int batch_run_impl(gomel_batch* b, gomel_timing* tm) {
    gomel_handle* h = b->h;
    const gomel_config& c = h->cfg;
    const int N = c.resolut, K = c.num_mels, I = b->iters >= 0 ? b->iters : c.griffin_lim_iterations;
    hipStream_t s = run_stream(b);
    if (!b->uploaded) return GOMEL_ERR_NULL;
    // this run's path: the batch's persistent kernel only while the device is free of this process's other work and has room for it
    // next to the other handles' persistent launches (the device gate); everything else waits for persistent launches in flight
    GatePersist gate_p;
    GateOrdinary gate_o;
    const int persist = (b->persist && I > 0 && gate_p.take(b)) ? b->persist : 0;
    if (!persist) gate_o.enter(h, s);
    if (tm) {
        while ((int)b->dom_ev.size() < 2 * I) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); b->dom_ev.push_back(e); }
        HIPCHK(hipEventRecord(b->ev[0], s));
    }
    const unsigned gx = (unsigned)std::min<int64_t>((b->max_len + 255) / 256, 4096), gy = (unsigned)std::min<size_t>(b->n_utt, 65535);
    const double* mel64 = (const double*)b->d_mel; const float* mel32 = (const float*)b->d_mel;
    const double* x064 = (const double*)b->d_x0;   const float* x032 = (const float*)b->d_x0;
    double* out64 = (double*)b->d_out;             float* out32 = (float*)b->d_out;
    if (I == 0) {
        // "I = 0 returns the random init" (mel/mel.go:85 never runs): keep it in the caller's precision
        if (b->io32) hipLaunchKernelGGL((k_init_x<float, float>), dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, x032, out32);
        else hipLaunchKernelGGL((k_init_x<double, double>), dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, x064, out64);
        if (tm) { HIPCHK(hipEventRecord(b->ev[1], s)); HIPCHK(hipEventRecord(b->ev[2], s)); }
    } else {
        if (b->precise) {   // fp64 generic kernels end to end
            if (h->generic) {
                if (b->direct) hipLaunchKernelGGL(k_mag_direct_nat, dim3((unsigned)b->total_frames), dim3(256), 0, s, mel64, b->d_magd, b->d_nyqd, b->d_midd, N / 2, N & 1);
                else if (b->io32) hipLaunchKernelGGL(k_mel_to_mag_nat<float>, dim3((unsigned)b->total_frames), dim3(256), 0, s, mel32, h->dt.undo, b->d_magd, b->d_nyqd, K, N / 2, c.tune_add, c.tune_mul);
                else hipLaunchKernelGGL(k_mel_to_mag_nat<double>, dim3((unsigned)b->total_frames), dim3(256), 0, s, mel64, h->dt.undo, b->d_magd, b->d_nyqd, K, N / 2, c.tune_add, c.tune_mul);
            }
            // (a precision schedule's fp32 magnitudes -- the fp64 rows rounded once -- leave in the same pass; from linear magnitudes they are narrowed below)
            const bool want32 = !h->generic && (b->mixed_k > 0 || b->pmix_k > 0);
            float* m32 = want32 && !b->direct ? b->d_mag : nullptr;
            float* n32 = want32 && !b->direct ? b->d_nyq : nullptr;
            if (h->generic) {}
            else if (b->direct) launch_mag_direct<double>(N, b->total_frames, s, mel64, b->d_magd, b->d_nyqd);
            else if (b->io32) HIPCHK(launch_mel_to_mag<double>(N, b->total_frames, s, mel32, h->dt.undo, b->d_magd, b->d_nyqd, K, c.tune_add, c.tune_mul, m32, n32));
            else HIPCHK(launch_mel_to_mag<double>(N, b->total_frames, s, mel64, h->dt.undo, b->d_magd, b->d_nyqd, K, c.tune_add, c.tune_mul, m32, n32));
            if (b->io32) hipLaunchKernelGGL((k_init_x<double, float>), dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, x032, b->d_xd[0]);   // float host buffers
            else hipLaunchKernelGGL((k_init_x<double, double>), dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, x064, b->d_xd[0]);
            if (b->mixed_k && b->direct) {
                // (from linear magnitudes, gomel_istft:) the fp32 kernel's magnitudes are the fp64 rows rounded once, in the same lane-slot order
                const int64_t nm = b->total_frames * (int64_t)(N / 2), nn = (b->total_frames + 3) & ~(int64_t)3;
                hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((nm / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_magd, b->d_mag, nm);
                hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((nn / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_nyqd, b->d_nyq, nn);
            }
            if (tm) HIPCHK(hipEventRecord(b->ev[1], s));
            int curd = 0;
            const int64_t spb = h->generic ? 1 : 128 / (N / 32), blocks = (b->total_frames + spb - 1) / spb;   // streams per 128-thread block
            if (b->fast) HIPCHK(hipMemsetAsync(b->d_taild[0], 0, b->segs.size() * (size_t)(N - c.window) * 8, s));
            if (b->mixed_k) {
                // Precision schedule (gomel_set_precision_schedule): the first mixed_k iterations in fp64 -- where a rounding-level
                // perturbation would still be amplified by every iteration that follows -- then x (tails folded in, rounded to
                // fp32 once) goes on through the fp32 kernel.  DESIGN.md section 2.
                for (int it = 0; it < b->mixed_k; it++) {
                    if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it], s));
                    HIPCHK(launch_fused_d(b, curd));
                    if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it + 1], s));
                    curd ^= 1;
                }
                if (b->fast == 1)   // (the any-hop form folds its tails into x behind every launch)
                    hipLaunchKernelGGL(k_add_tails_d, dim3((unsigned)b->segs.size()), dim3(256), 0, s, b->d_segs, b->d_taild[curd], b->d_xd[curd], N - c.window, c.window);
                const unsigned gn = (unsigned)std::min<int64_t>((b->total_len / 4 + 255) / 256, 65536);
                hipLaunchKernelGGL(k_narrow, dim3(gn), dim3(256), 0, s, b->d_xd[curd], b->d_x[0], b->total_len);
                HIPCHK(hipMemsetAsync(b->d_tail[0], 0, b->segs.size() * (size_t)(N - c.window) * 4, s));
                int cur = 0;
                for (int it = b->mixed_k; it < I; it++) {
                    if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it], s));
                    launch_fused(b, cur);
                    if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it + 1], s));
                    cur ^= 1;
                }
                if (tm) HIPCHK(hipEventRecord(b->ev[2], s));
                const unsigned gf = (unsigned)std::min<int64_t>((b->total_len + 255) / 256, 65536);
                if (b->io32) {
                    hipLaunchKernelGGL(k_finalize<float>, dim3(gf), dim3(256), 0, s, b->d_x[cur], out32, b->total_len);
                    if (b->fast == 1) hipLaunchKernelGGL(k_add_tails<float>, dim3((unsigned)b->segs.size()), dim3(256), 0, s, b->d_segs, b->d_tail[cur], out32, N - c.window, c.window);
                } else {
                    hipLaunchKernelGGL(k_finalize<double>, dim3(gf), dim3(256), 0, s, b->d_x[cur], out64, b->total_len);
                    if (b->fast == 1) hipLaunchKernelGGL(k_add_tails<double>, dim3((unsigned)b->segs.size()), dim3(256), 0, s, b->d_segs, b->d_tail[cur], out64, N - c.window, c.window);
                }
            } else {
            if (persist) {   // 64-bit mode, a batch that fits the chip: every iteration in one launch (k_gl_persist_d), then x_I = overlap-add of the last frames
                const size_t buf_doubles = (size_t)b->total_frames * N;
                const int I64 = b->pmix_k ? b->pmix_k : I;          // precision schedule: the fp64 launch covers the first pmix_k iterations only
                HIPCHK(hipMemsetAsync(b->d_ctl, 0, b->ctl_bytes, s));
                HIPCHK(hipMemsetAsync(b->d_ywd, 0xff, buf_doubles * 8, s));
                if (b->pmix_k) {   // the fp32 launch: its magnitudes left k_mel_to_mag beside the fp64 rows (linear magnitudes: narrowed here); its exchange buffer starts as "not stored yet"
                    if (b->direct) {
                        const int64_t nm = b->total_frames * (int64_t)(N / 2), nn = (b->total_frames + 3) & ~(int64_t)3;
                        hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((nm / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_magd, b->d_mag, nm);
                        hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((nn / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_nyqd, b->d_nyq, nn);
                    }
                    HIPCHK(hipMemsetAsync(b->d_yw, 0xff, buf_doubles * 4, s));
                }
                if (tm) HIPCHK(hipEventRecord(b->dom_ev[0], s));
                // (dynamic LDS: exchange areas + window + twiddles; the size limit was raised -- and the census taken with it -- in persist_census_d)
                for (const std::pair<int64_t, int64_t>& gr : b->pgroups) {   // (one launch; two, one after the other, for a batch of up to twice the chip's capacity)
                    const int spw_d = 128 / (N / 32);
                    const unsigned grid = (unsigned)((gr.second - gr.first + spw_d - 1) / spw_d);
                    GOMEL_DISPATCH_N(N, hipLaunchKernelGGL(k_gl_persist_d<NN>, dim3(grid), dim3(128), persist_d_lds_bytes<NN>(), s, b->d_xd[0], b->d_utts, b->d_frame_utt, b->d_magd, b->d_nyqd,
                                                           b->d_ywd, b->d_ctl, b->total_frames, I64, c.window, h->persist_ticks, h->persist_nap, h->dt, gr.first, gr.second));
                }
                if (b->pmix_k) {
                    // x_k = the overlap-add of the fp64 launch's last frames (k_ola's order), rounded to fp32 once; then the remaining
                    // iterations in the fp32 launch, and x_I = the overlap-add of ITS last frames.  A timeout in either launch sets the
                    // same word: the host recomputes the whole batch with the per-iteration kernels.
                    hipLaunchKernelGGL(k_ola<double>, dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, b->d_ywd + (size_t)((I64 - 1) % 3) * buf_doubles, b->d_xd[1], N, c.window);
                    hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((b->total_len / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_xd[1], b->d_x[0], b->total_len);
#define GOMEL_PERSIST_LAUNCH(OCC, DEPTH)                                                                                                                     \
    GOMEL_DISPATCH_N(N, hipLaunchKernelGGL((k_gl_persist<NN, OCC, DEPTH>), dim3((unsigned)b->n_wg), dim3(128), 0, s, b->d_x[0], b->d_utts, b->d_frame_utt, b->d_mag, \
                                           b->d_nyq, b->d_yw, b->d_ctl, b->total_frames, I - I64, c.window, h->persist_ticks, h->persist_nap, h->dt, (int64_t)0, b->total_frames))
                    if (b->pmix_occ == 2) { GOMEL_PERSIST_LAUNCH(2, 2); } else { GOMEL_PERSIST_LAUNCH(1, 4); }
#undef GOMEL_PERSIST_LAUNCH
                    if (tm) HIPCHK(hipEventRecord(b->dom_ev[1], s));
                    const float* y_last = b->d_yw + (size_t)((I - I64 - 1) % 3) * buf_doubles;
                    if (b->io32) hipLaunchKernelGGL(k_ola_out<float>, dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, y_last, out32, N, c.window);
                    else hipLaunchKernelGGL(k_ola_out<double>, dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, y_last, out64, N, c.window);
                } else {
                if (tm) HIPCHK(hipEventRecord(b->dom_ev[1], s));
                // x_I = the overlap-add of the last iteration's frames (float host buffers: summed in fp64 into the idle second signal buffer, narrowed below)
                hipLaunchKernelGGL(k_ola<double>, dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, b->d_ywd + (size_t)((I - 1) % 3) * buf_doubles, b->io32 ? b->d_xd[1] : out64, N, c.window);
                if (b->io32) hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((b->total_len / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_xd[1], out32, b->total_len);
                }
            }
            const bool gmix = !persist && b->pmix_k > 0;           // precision schedule on the generic kernels (also the recomputation of a persistent launch that gave up: the same bits)
            const int I64g = gmix ? b->pmix_k : I;
            for (int it = 0; it < I64g && !persist; it++) {
                if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it], s));
                if (b->fast) { HIPCHK(launch_fused_d(b, curd)); if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it + 1], s)); curd ^= 1; continue; }
                if (h->big)
                    big_frames(h, s, 0, b->d_xd[curd], b->d_frame_xoff, b->total_frames, b->d_bigA, b->d_bigB, b->d_magd, b->d_nyqd, b->d_midd, b->d_ywd);
                else if (h->generic)
                    hipLaunchKernelGGL(glany::k_any_frames<glany::kAnyGl>, dim3((unsigned)b->total_frames), dim3(glany::kAnyThreads), (size_t)h->dt.any.M * 16, s,
                                       b->d_xd[curd], b->d_frame_xoff, b->d_magd, b->d_nyqd, b->d_ywd, b->total_frames, N, h->dt.windowd, h->dt.any, b->d_midd);
                else
                GOMEL_DISPATCH_N(N, hipLaunchKernelGGL(k_gl_frames_d<NN>, dim3((unsigned)blocks), dim3(128), 0, s, b->d_xd[curd], b->d_frame_xoff, b->d_magd, b->d_nyqd, b->d_ywd, b->total_frames, h->dt));
                if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it + 1], s));
                hipLaunchKernelGGL(k_ola<double>, dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, b->d_ywd, b->d_xd[curd ^ 1], N, c.window);
                curd ^= 1;
            }
            if (gmix) {   // hand-over: the signal rounded to fp32 once (the fp32 magnitudes left k_mel_to_mag beside the fp64 rows; linear magnitudes: narrowed here); then the fp32 generic pair
                if (b->direct) {
                    const int64_t nm = b->total_frames * (int64_t)(N / 2), nn = (b->total_frames + 3) & ~(int64_t)3;
                    hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((nm / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_magd, b->d_mag, nm);
                    hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((nn / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_nyqd, b->d_nyq, nn);
                }
                hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((b->total_len / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_xd[curd], b->d_x[0], b->total_len);
                int cur = 0;
                for (int it = I64g; it < I; it++) {
                    if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it], s));
                    GOMEL_DISPATCH_N(N, launch_frames<NN>(b, b->d_x[cur]));
                    if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it + 1], s));
                    hipLaunchKernelGGL(k_ola<float>, dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, b->d_yw, b->d_x[cur ^ 1], N, c.window);
                    cur ^= 1;
                }
                if (tm) HIPCHK(hipEventRecord(b->ev[2], s));
                const unsigned gf = (unsigned)std::min<int64_t>((b->total_len + 255) / 256, 65536);
                if (b->io32) hipLaunchKernelGGL(k_finalize<float>, dim3(gf), dim3(256), 0, s, b->d_x[cur], out32, b->total_len);
                else hipLaunchKernelGGL(k_finalize<double>, dim3(gf), dim3(256), 0, s, b->d_x[cur], out64, b->total_len);
            } else {
            if (tm) HIPCHK(hipEventRecord(b->ev[2], s));
            if (b->want_spec && h->big)
                big_frames(h, s, 1, b->d_xd[curd ^ 1], b->d_frame_xoff, b->total_frames, b->d_bigA, b->d_bigB, b->d_magd, b->d_nyqd, b->d_midd, b->d_spec);
            else if (b->want_spec && h->generic)   // the literal form's clamp step on x_{I-1}, bins 0 ... N/2 (d_magd holds |spectrogram| as the caller gave it)
                hipLaunchKernelGGL(glany::k_any_frames<glany::kAnySpec>, dim3((unsigned)b->total_frames), dim3(glany::kAnyThreads), (size_t)h->dt.any.M * 16, s,
                                   b->d_xd[curd ^ 1], b->d_frame_xoff, b->d_magd, b->d_nyqd, b->d_spec, b->total_frames, N, h->dt.windowd, h->dt.any, b->d_midd);
            else if (b->want_spec)   // d_xd[curd ^ 1] is x_{I-1}: the signal whose phases the last iteration imposed on the spectrogram
                GOMEL_DISPATCH_N(N, hipLaunchKernelGGL((k_spec_writeback<NN, v2d>), dim3((unsigned)blocks), dim3(128), 0, s, b->d_xd[curd ^ 1], b->d_frame_xoff, mel64, b->d_spec, b->total_frames, h->dt));
            // the fused kernel's tails are folded into x in place, in fp64 (the any-hop form folds its tails itself), then x leaves in the caller's type
            if (b->fast == 1) hipLaunchKernelGGL(k_add_tails_d, dim3((unsigned)b->segs.size()), dim3(256), 0, s, b->d_segs, b->d_taild[curd], b->d_xd[curd], N - c.window, c.window);
            if (!persist && b->io32) hipLaunchKernelGGL(k_narrow, dim3((unsigned)std::min<int64_t>((b->total_len / 4 + 255) / 256, 65536)), dim3(256), 0, s, b->d_xd[curd], out32, b->total_len);
            else if (!persist) HIPCHK(hipMemcpyAsync(b->d_out, b->d_xd[curd], (size_t)b->total_len * 8, hipMemcpyDeviceToDevice, s));
            }   // !gmix
            }   // !mixed_k
        } else {
        if (b->direct) launch_mag_direct<float>(N, b->total_frames, s, mel64, b->d_mag, b->d_nyq);
        else if (b->io32) HIPCHK(launch_mel_to_mag<float>(N, b->total_frames, s, mel32, h->dt.undo, b->d_mag, b->d_nyq, K, c.tune_add, c.tune_mul));
        else HIPCHK(launch_mel_to_mag<float>(N, b->total_frames, s, mel64, h->dt.undo, b->d_mag, b->d_nyq, K, c.tune_add, c.tune_mul));
        if (b->io32) hipLaunchKernelGGL((k_init_x<float, float>), dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, x032, b->d_x[0]);
        else hipLaunchKernelGGL((k_init_x<float, double>), dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, x064, b->d_x[0]);
        if (tm) HIPCHK(hipEventRecord(b->ev[1], s));
        int cur = 0;
        const size_t tail_bytes = b->segs.size() * (size_t)(N - c.window) * 4;
        if (b->fast) HIPCHK(hipMemsetAsync(b->d_tail[0], 0, tail_bytes, s));  // x0 has no tail contributions
        if (persist) {   // all I iterations in one launch; x_I = overlap-add of the last iteration's frames
            const size_t buf_floats = (size_t)b->total_frames * N;
            HIPCHK(hipMemsetAsync(b->d_ctl, 0, b->ctl_bytes, s));
            // buf[0] starts as "not stored yet" (kPersistSentinel); the streams themselves reset their slots of buf[1] and buf[2] in
            // iterations 0 and 1, and drain that before anything a reader of those slots could have waited for
            HIPCHK(hipMemsetAsync(b->d_yw, 0xff, buf_floats * 4, s));
            if (tm) HIPCHK(hipEventRecord(b->dom_ev[0], s));
#define GOMEL_PERSIST_LAUNCH(OCC, DEPTH)                                                                                                                     \
    GOMEL_DISPATCH_N(N, hipLaunchKernelGGL((k_gl_persist<NN, OCC, DEPTH>), dim3(grid), dim3(128), 0, s, b->d_x[0], b->d_utts, b->d_frame_utt, b->d_mag, \
                                           b->d_nyq, b->d_yw, b->d_ctl, b->total_frames, I, c.window, h->persist_ticks, h->persist_nap, h->dt, gr.first, gr.second))
            for (const std::pair<int64_t, int64_t>& gr : b->pgroups) {   // (one launch; two, one after the other, for a batch of up to two chip-fulls)
                const int spw_f = 128 / (N / 32);
                const unsigned grid = (unsigned)((gr.second - gr.first + spw_f - 1) / spw_f);
                if ((int)grid > persist_census(h, 1)) { GOMEL_PERSIST_LAUNCH(2, 2); }   // the kernel's OCC: one wave per SIMD while the launch fits that way
                else { GOMEL_PERSIST_LAUNCH(1, 4); }   // (eight covering frames in flight, for hops below N/4, measured slower than two rounds of four: its registers come out of the tables')
            }
#undef GOMEL_PERSIST_LAUNCH
            if (tm) HIPCHK(hipEventRecord(b->dom_ev[1], s));
            const float* y_last = b->d_yw + (size_t)((I - 1) % 3) * buf_floats;
            if (b->io32) hipLaunchKernelGGL(k_ola_out<float>, dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, y_last, out32, N, c.window);
            else hipLaunchKernelGGL(k_ola_out<double>, dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, y_last, out64, N, c.window);
        }
        for (int it = 0; it < I && !persist; it++) {
            if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it], s));
            if (b->fast) launch_fused(b, cur);
            else { GOMEL_DISPATCH_N(N, launch_frames<NN>(b, b->d_x[cur])); }
            if (tm) HIPCHK(hipEventRecord(b->dom_ev[2 * it + 1], s));
            if (!b->fast) hipLaunchKernelGGL(k_ola<float>, dim3(gx, gy), dim3(256), 0, s, b->d_utts, (int)b->n_utt, b->d_yw, b->d_x[cur ^ 1], N, c.window);
            cur ^= 1;
        }
        if (tm) HIPCHK(hipEventRecord(b->ev[2], s));
        if (b->want_spec) {   // generic kernels were forced: d_x[cur ^ 1] is x_{I-1}, contiguous
            constexpr int dummy = 0; (void)dummy;
            GOMEL_DISPATCH_N(N, { constexpr int SPB = 128 / Geo<NN>::T;
                                  hipLaunchKernelGGL((k_spec_writeback<NN, v2f>), dim3((unsigned)((b->total_frames + SPB - 1) / SPB)), dim3(128), 0, s, b->d_x[cur ^ 1], b->d_frame_xoff, mel64, b->d_spec, b->total_frames, h->dt); });
        }
        const unsigned gf = (unsigned)std::min<int64_t>((b->total_len + 255) / 256, 65536);
        if (persist) {
            // k_ola_out has written the waveforms
        } else if (b->io32) {
            hipLaunchKernelGGL(k_finalize<float>, dim3(gf), dim3(256), 0, s, b->d_x[cur], out32, b->total_len);
            if (b->fast == 1) hipLaunchKernelGGL(k_add_tails<float>, dim3((unsigned)b->segs.size()), dim3(256), 0, s, b->d_segs, b->d_tail[cur], out32, N - c.window, c.window);
        } else {
            hipLaunchKernelGGL(k_finalize<double>, dim3(gf), dim3(256), 0, s, b->d_x[cur], out64, b->total_len);
            if (b->fast == 1) hipLaunchKernelGGL(k_add_tails<double>, dim3((unsigned)b->segs.size()), dim3(256), 0, s, b->d_segs, b->d_tail[cur], out64, N - c.window, c.window);
        }
        }
    }
    if (persist && I > 0) HIPCHK(hipMemcpyAsync(h->err_slab + b->err_slot, b->d_ctl, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b->gate_ev, s));
    gate_p.leave(b, b->gate_ev);
    gate_o.leave(b->gate_ev, b);
    if (tm) {
        HIPCHK(hipEventRecord(b->ev[3], s));
        HIPCHK(hipEventSynchronize(b->ev[3]));
        if (persist) persist_release(b);   // seen to finish
        if (int se = sync_err_check(h)) return se;
        if (persist && I > 0 && h->err_slab[b->err_slot]) return persist_redo(b, tm);
        if (persist && I > 0) h->persist_backoff = 64;
        HIPCHK(hipEventElapsedTime(&tm->total_ms, b->ev[0], b->ev[3]));
        HIPCHK(hipEventElapsedTime(&tm->iter_ms, b->ev[1], b->ev[2]));
        tm->dom_ms = 0.f; tm->dom64_ms = 0.f;
        const int n64 = persist ? 0 : b->mixed_k ? b->mixed_k : (b->precise && b->fast) ? I : (b->precise && b->pmix_k) ? b->pmix_k : 0;   // launches of the fp64 kernel among them
        for (int it = 0; it < (persist ? std::min(I, 1) : I); it++) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, b->dom_ev[2 * it], b->dom_ev[2 * it + 1]));
            tm->dom_ms += ms;
            if (it < n64) tm->dom64_ms += ms;
        }
        tm->iter64_launches = n64;
        tm->iter_launches = persist ? std::min(I, 1) : I;
        tm->fast_path = persist ? 2 : b->fast == 2 ? 3 : b->fast;
        tm->frames = b->total_frames;
    }
    return GOMEL_OK;
}

template <int N> static void launch_fft(gomel_handle* h, bool inverse, const float* in, float* out, int64_t n_frames) {
    typedef Geo<N> G;
    constexpr int STREAMS = 128 / G::T;
    const int64_t blocks = (n_frames + STREAMS - 1) / STREAMS;
    if (!inverse) hipLaunchKernelGGL(k_rfft<N>, dim3((unsigned)blocks), dim3(128), 0, h->stream, in, out, n_frames, h->dt);
    else hipLaunchKernelGGL(k_irfft<N>, dim3((unsigned)blocks), dim3(128), 0, h->stream, in, out, n_frames, h->dt);
}

static int fft_hook(gomel_handle* h, bool inverse, const float* in, size_t n_frames, float* out) {
    if (!h || !in || !out) return GOMEL_ERR_NULL;
    if (h->generic) return GOMEL_ERR_UNSUPPORTED_N;   // register FFT cores only (Resolut 256 ... 4096, powers of two)
    if (n_frames == 0) return GOMEL_ERR_EMPTY;
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device); HIPCHK(dg.err);
    GateOrdinary gate_o; gate_o.enter(h, h->stream);   // synchronous entry point: the gate is left (destructor) after its streams have drained
    const int N = h->cfg.resolut;
    const size_t in_b = n_frames * (inverse ? (size_t)(N + 2) : (size_t)N) * 4, out_b = n_frames * (inverse ? (size_t)N : (size_t)(N + 2)) * 4;
    float *d_in = nullptr, *d_out = nullptr;
    HIPCHK((hipError_t)h->pool.alloc((void**)&d_in, in_b));
    if (h->pool.alloc((void**)&d_out, out_b) != hipSuccess) { h->pool.release(d_in); return GOMEL_ERR_HIP; }
    int rc = GOMEL_OK;
    if (hipMemcpyAsync(d_in, in, in_b, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = GOMEL_ERR_HIP;
    if (!rc) {
        GOMEL_DISPATCH_N(N, launch_fft<NN>(h, inverse, d_in, d_out, (int64_t)n_frames));
        if (hipMemcpyAsync(out, d_out, out_b, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
            g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP;
        }
    }
    h->pool.release(d_in); h->pool.release(d_out);
    return rc;
}

// =============================================================================================== C ABI
extern "C" {

int gomel_last_hip_error(void) { return g_last_hip; }

int gomel_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gomel_device_synchronize(int32_t device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError(); return GOMEL_ERR_NO_DEVICE; }
    if (device >= n) return GOMEL_ERR_BAD_CONFIG;
    if (device < 0) { HIPCHK(hipDeviceSynchronize()); return GOMEL_OK; }
    DeviceGuard dg(device); HIPCHK(dg.err);
    HIPCHK(hipDeviceSynchronize());
    return GOMEL_OK;
}

int gomel_create(const gomel_config* cfg, gomel_handle** out) { GOMEL_TRY return create_common(cfg, nullptr, 0, out); GOMEL_CATCH }
int gomel_create_from_tables(const gomel_config* cfg, const void* blob, size_t n, gomel_handle** out) {
    if (!blob) return GOMEL_ERR_NULL;
    GOMEL_TRY return create_common(cfg, blob, n, out); GOMEL_CATCH
}

size_t gomel_tables_bytes(const gomel_handle* h) {
    if (!h) return 0;
    return sizeof(TableHeader) + (size_t)h->cfg.resolut * 8 + ((size_t)h->cfg.resolut / 2 + h->cfg.num_mels) * sizeof(BinRow);
}

int gomel_tables_export(const gomel_handle* h, void* blob, size_t cap) {
    if (!h || !blob) return GOMEL_ERR_NULL;
    if (cap < gomel_tables_bytes(h)) return GOMEL_ERR_BUFFER_TOO_SMALL;
    std::lock_guard<std::mutex> lk(const_cast<gomel_handle*>(h)->mu);   // gomel_set_window may be replacing the window
    pack_tables(h->cfg, h->dom_status, h->window, h->undo, h->dom, blob);
    return GOMEL_OK;
}

static int set_window_guarded(gomel_handle* h, const double* w, size_t n) {
    if (!h || !w) return GOMEL_ERR_NULL;
    if (n != (size_t)h->cfg.resolut) return GOMEL_ERR_BAD_LENGTH;
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device); HIPCHK(dg.err);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->window.assign(w, w + n);
    return upload_tables(h);
}

int gomel_get_window(const gomel_handle* h, double* w, size_t cap) {
    if (!h || !w) return GOMEL_ERR_NULL;
    if (cap < h->window.size()) return GOMEL_ERR_BUFFER_TOO_SMALL;
    std::lock_guard<std::mutex> lk(const_cast<gomel_handle*>(h)->mu);
    memcpy(w, h->window.data(), h->window.size() * 8);
    return GOMEL_OK;
}

int gomel_set_path(gomel_handle* h, int mode) {
    if (!h) return GOMEL_ERR_NULL;
    if (mode < 0 || mode > 3) return GOMEL_ERR_BAD_CONFIG;
    if (mode == 2 && !fused_applicable(h->cfg)) return GOMEL_ERR_BAD_CONFIG;
    if (mode == 3 && !persist_geometry_ok(h->cfg)) return GOMEL_ERR_BAD_CONFIG;
    std::lock_guard<std::mutex> lk(h->mu);
    h->path_mode = mode;
    return GOMEL_OK;
}

int gomel_set_call_mode(gomel_handle* h, int mode) {
    if (!h) return GOMEL_ERR_NULL;
    if (mode != 0 && mode != 1) return GOMEL_ERR_BAD_CONFIG;
    h->call_mode.store(mode);
    return GOMEL_OK;
}
int gomel_set_precision(gomel_handle* h, int bits) {
    if (!h) return GOMEL_ERR_NULL;
    if (bits != 0 && bits != 32 && bits != 64) return GOMEL_ERR_BAD_CONFIG;
    std::lock_guard<std::mutex> lk(h->mu);
    h->precision = bits == 0 ? kPrecAuto : bits == 32 ? kPrecFp32 : kPrecFp64;
    h->precise_iters = -1;
    return GOMEL_OK;
}
int gomel_set_precision_schedule(gomel_handle* h, int32_t fp64_iterations) {
    if (!h) return GOMEL_ERR_NULL;
    std::lock_guard<std::mutex> lk(h->mu);
    if (fp64_iterations == GOMEL_SCHEDULE_AUTO) { h->precision = kPrecAuto; h->precise_iters = -1; }
    else if (fp64_iterations == 0) { h->precision = kPrecFp32; h->precise_iters = -1; }
    else if (fp64_iterations < 0) { h->precision = kPrecFp64; h->precise_iters = -1; }
    else { h->precision = kPrecSchedule; h->precise_iters = fp64_iterations; }
    return GOMEL_OK;
}
int gomel_get_precision(const gomel_handle* h, int32_t iterations, int32_t* fp64_iterations) {
    if (!h || !fp64_iterations) return GOMEL_ERR_NULL;
    if (iterations < 0) iterations = h->cfg.griffin_lim_iterations;
    std::lock_guard<std::mutex> lk(const_cast<gomel_handle*>(h)->mu);
    *fp64_iterations = resolve_fp64_iterations(h, iterations);
    return GOMEL_OK;
}

int gomel_destroy(gomel_handle* h) {
    if (!h) return GOMEL_OK;
    if (h->jobs_outstanding.load() != 0) return GOMEL_ERR_BUSY;   // a submitted job has not been waited for (its buffers live in this handle)
    {
    DeviceGuard dg(h->device);
    if (h->stream) { hipStreamSynchronize(h->stream); hipStreamDestroy(h->stream); }
    if (h->copy_stream) { hipStreamSynchronize(h->copy_stream); hipStreamDestroy(h->copy_stream); }
    if (h->up_stream) { hipStreamSynchronize(h->up_stream); hipStreamDestroy(h->up_stream); }
    if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
    h->pool.clear();
    if (h->d_blob) hipFree(h->d_blob);
    if (h->err_slab) hipHostFree(h->err_slab);
    }
    delete h;
    return GOMEL_OK;
}

int gomel_device_name(const gomel_handle* h, char* buf, size_t cap) {
    if (!h || !buf || cap == 0) return GOMEL_ERR_NULL;
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, h->device));
    snprintf(buf, cap, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return GOMEL_OK;
}

// One plan for both kinds of input: log-mel pairs (n_pairs[u] = frames * NumMels) or, for gomel_istft, linear magnitudes
// (`direct`: n_pairs[u] = frames, rows of N/2+1 doubles).  `iters` >= 0 overrides the configured iteration count.
enum { BF_DIRECT = 1, BF_IO32 = 2, BF_SPEC = 4, BF_NOPERSIST = 8 };   // batch_create_common flags (NOPERSIST: a submitted job -- a persistent launch wants a verifying wait)
// init_stream: where the batch's descriptor copies and zero-fills are queued -- the stream its upload will use (every reader waits for
// the upload): the handle's compute stream for the batch API, the upload stream inside the chunk pipeline, where anything queued on
// the compute stream would sit (and, coming from pageable memory, hold the host) behind the previous chunks' kernels.
static int batch_create_common(gomel_handle* h, size_t n_utt, const size_t* n_pairs, int flags, int iters, gomel_batch** out, hipStream_t init_stream = nullptr) {
    const int direct = (flags & BF_DIRECT) ? 1 : 0;
    if (!h || !n_pairs || !out) return GOMEL_ERR_NULL;
    *out = nullptr;
    if (n_utt == 0) return GOMEL_ERR_EMPTY;
    const gomel_config& c = h->cfg;
    const int N = c.resolut, K = direct ? 1 : c.num_mels, hop = c.window;
    for (size_t u = 0; u < n_utt; u++) {
        if (n_pairs[u] == 0) return GOMEL_ERR_EMPTY;
        if (n_pairs[u] % (size_t)K) return GOMEL_ERR_BAD_LENGTH;
    }
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device); HIPCHK(dg.err);
    gomel_batch* b = new (std::nothrow) gomel_batch();
    if (!b) return GOMEL_ERR_NO_MEMORY;
    BatchGuard guard{b};
    b->h = h; b->n_utt = n_utt; b->direct = direct; b->iters = iters;
    h->batches_seen++;
    b->io32 = (flags & BF_IO32) ? 1 : 0; b->esz = b->io32 ? 4 : 8; b->want_spec = (flags & BF_SPEC) ? 1 : 0;
    if (b->io32 && direct) return GOMEL_ERR_BAD_CONFIG;   // fp32 host I/O: the log-mel entry points (any precision mode: float at the boundary, the mode's arithmetic inside)
    b->utts.resize(n_utt); b->n_pairs.assign(n_pairs, n_pairs + n_utt);
    for (size_t u = 0; u < n_utt; u++) {
        UttDesc& d = b->utts[u];
        d.frames = (int32_t)(n_pairs[u] / (size_t)K);
        d.len = (int64_t)N + (int64_t)(d.frames - 1) * hop;
        d.xoff = b->total_len; d.fbase = b->total_frames; d.has_x0 = 0; d.seed = u;
        b->total_len += (d.len + 3) & ~(int64_t)3;  // every utterance starts 16-byte aligned (float4 row pairs in k_gl_fused)
        b->total_frames += d.frames; b->total_pairs += (int64_t)n_pairs[u];
        b->max_len = std::max(b->max_len, d.len);
    }
    // Path choice.  A batch whose frames all fit the chip at once (one utterance, a few short ones) takes the persistent kernel:
    // every iteration in one launch, ~9 us per iteration whatever the frame count.  Beyond that the fused kernel, which walks
    // consecutive frames per stream and so only pays once there are enough frames to fill the chip (fused_min_frames); in
    // between (and for hops neither takes) every frame in parallel, two launches per iteration (generic kernels).
    const int I = iters >= 0 ? iters : c.griffin_lim_iterations;
    const int k64 = resolve_fp64_iterations(h, I);            // fp64 iterations of this call (0: fp32 throughout; I: fp64 throughout)
    b->precise = (k64 > 0 || h->generic) ? 1 : 0;
    // (every frame gathers its whole window from the covering frames, N/hop times the bytes of k_ola's single pass: beyond eight
    // covering frames that traffic costs more than the launches it saves -- measured at N 2048: hop 256 0.76 ms against 1.05 ms
    // generic, hop 128 1.25 against 1.14 -- so the library picks the kernel down to hop = N/8 only; gomel_set_path(h, 3) still runs it)
    const bool persist_pays = (N - 1) / hop <= kPersistAutoNb || h->path_mode == 3;
    if (!h->generic && !b->want_spec && !(flags & BF_NOPERSIST) && (h->path_mode == 0 || h->path_mode == 3) && persist_geometry_ok(c) && persist_pays && I >= 2) {
        const int spw = 128 / (N / 32);                                  // streams (frames) per workgroup
        const int64_t n_wg = (b->total_frames + spw - 1) / spw;
        int max_frames = 0;
        for (const UttDesc& d : b->utts) max_frames = std::max(max_frames, d.frames);
        const bool room = h->batches_seen >= h->persist_retry_at && h->err_slab && !h->err_free.empty() && b->max_len < ((int64_t)1 << 30) && max_frames < (1 << 20);
        // utterances packed, in order, into launches of at most `cap` frames each (empty: some utterance does not fit a launch alone)
        auto pack = [&](int64_t cap) {
            std::vector<std::pair<int64_t, int64_t>> groups;
            int64_t g0 = 0, gf = 0;
            for (const UttDesc& d : b->utts) {
                if (d.frames > cap) return std::vector<std::pair<int64_t, int64_t>>();
                if (gf + d.frames > cap) { groups.emplace_back(g0, g0 + gf); g0 += gf; gf = 0; }
                gf += d.frames;
            }
            if (gf) groups.emplace_back(g0, g0 + gf);
            return groups;
        };
        if (room && !b->precise && b->total_frames * (int64_t)N < ((int64_t)1 << 28)) {
            // fp32: one launch while the batch fits the chip (OCC 1 or 2); up to TWO chip-fulls as two launches one after the other, each over
            // whole utterances (round 4: 3-4 x 342 frames at N 4096, 6-10 x 400 at N 2048 took the generic pair before)
            // (two launches only at up to four covering frames per sample, as for the fp64 kernel below: N 4096 / hop 1280 3-4 x 342 frames 1.76-1.98 ->
            // 1.37-1.60 ms, N 1024 / hop 256 4800-6400 frames 1.60-1.89 (fused) -> 1.39-1.69, but N 2048 / hop 256 2400-3200 frames 1.90-2.07 -> 1.95-2.25)
            const std::vector<std::pair<int64_t, int64_t>> groups = pack((int64_t)persist_census(h, 2) * spw);
            if (groups.size() == 1 || (groups.size() == 2 && (N - 1) / hop <= 3)) {
                b->persist = (groups.size() == 1 && n_wg <= persist_census(h, 1)) ? 1 : 2;   // (what the census / the gate's share is taken with)
                b->pgroups = groups;
            }
        } else if (room && b->precise && b->total_frames * (int64_t)N < ((int64_t)1 << 27)) {
            // 64-bit mode: k_gl_persist_d (three buffers of doubles: < 2^32 bytes).  Its census is half the fp32 kernel's (one wave per SIMD), so a
            // batch of up to TWO chip-fulls runs as two launches one after the other, each over whole utterances (round 4: 2 x 342 frames at
            // N 4096, 6-10 x 400 at N 1024 took the generic pair before, 1.25-1.3 x the time); more launches than two stop paying against the
            // generic kernels, which run every frame of the batch in parallel.
            const std::vector<std::pair<int64_t, int64_t>> groups = pack((int64_t)persist_census(h, 3) * spw);
            const bool fits = !groups.empty();
            const bool mix = k64 > 0 && k64 < I;
            // two launches only where the fp32 part of a schedule still runs as ONE launch of the fp32 kernel (or there is no fp32 part), and only
            // at up to four covering frames per sample ((N-1)/hop <= 3: one round of gather loads per iteration) -- measured (scripts/size_sweep.py,
            // profiles/r04/size_sweep_*.txt): N 4096 / hop 1280 2 x 342 frames 1.93 -> 1.47 ms, N 1024 / hop 256 2400-4000 frames 1.70-1.91 -> 1.33-1.49 ms,
            // but N 2048 / hop 256 (eight covering frames) 1200-1600 frames 1.79-1.95 -> 1.85-2.00 ms: there the generic pair stays
            const bool two_pay = (N - 1) / hop <= 3;
            if (fits && (groups.size() == 1 || (groups.size() == 2 && two_pay && (!mix || n_wg <= persist_census(h, 2))))) {
                b->persist = 3;
                b->pgroups = groups;
                // a precision schedule on this path too (round 4): the fp64 launch(es) for the first k64 iterations, the fp32 launch for the rest
                if (mix && n_wg <= persist_census(h, 2)) { b->pmix_k = k64; b->pmix_occ = n_wg <= persist_census(h, 1) ? 1 : 2; }
            }
        }
        if (b->persist) {
            b->n_wg = (int)n_wg;
            b->err_slot = h->err_free.back(); h->err_free.pop_back();
        }
    }
    const int kind = fused_kind(c);   // (0 for generic lengths)
    b->fast = (!b->persist && !b->want_spec && h->path_mode != 1 && h->path_mode != 3 && kind &&
               (h->path_mode == 2 || b->total_frames >= fused_min_frames(c, k64 > 0))) ? kind : 0;
    if (h->path_mode == 3 && !b->persist) return GOMEL_ERR_BAD_CONFIG;   // the persistent kernel was required and this batch does not fit it
    // Precision schedule: on the fused kernel's batches and (pmix_k, above) on the persistent kernel's.  What the generic kernels take
    // (a handful of utterances that fit neither) and the generic lengths stay fp64 for the whole call: correct, and cheap at that size.
    if (b->precise && !h->generic && b->fast && k64 > 0 && k64 < I) b->mixed_k = k64;
    if (b->precise && !h->generic && !b->fast && !b->persist && !b->want_spec && k64 > 0 && k64 < I) { b->pmix_k = k64; b->pmix_occ = 0; }   // the generic pair: fp64 launches, then fp32 ones
    int rc = GOMEL_OK;
    const hipStream_t ist = init_stream ? init_stream : h->stream;
    auto alloc = [&](void** p, size_t bytes) { if (!rc && h->pool.alloc(p, bytes) != hipSuccess) { g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP; } };
    alloc((void**)&b->d_mel, direct ? (size_t)b->total_frames * istft_row(N) * 8 : (size_t)b->total_pairs * 2 * b->esz);
    alloc((void**)&b->d_out, (size_t)b->total_len * b->esz);
    if (b->want_spec && I > 0) alloc((void**)&b->d_spec, (size_t)b->total_frames * istft_row(N) * 16);
    if (direct && (N & 1) && I > 0) alloc((void**)&b->d_midd, (size_t)b->total_frames * 8);
    alloc((void**)&b->d_utts, n_utt * sizeof(UttDesc));
    if (I > 0 && b->precise) {
        alloc((void**)&b->d_magd, (size_t)b->total_frames * (N / 2) * 8);
        alloc((void**)&b->d_nyqd, ((size_t)b->total_frames + 4) * 8);   // + 4: narrowed four at a time into d_nyq (precision schedule)
        alloc((void**)&b->d_xd[0], (size_t)b->total_len * 8);
        alloc((void**)&b->d_xd[1], (size_t)b->total_len * 8);
        if (b->fast) {   // fp64 fused kernel: segments and tails as in the fp32 case, no per-frame output buffer
            plan_segments(b->h->cfg, b->utts, b->total_frames, b->segs, b->g_max);
            alloc((void**)&b->d_segs, b->segs.size() * sizeof(SegDesc));
            alloc((void**)&b->d_taild[0], b->segs.size() * (size_t)(N - hop) * 8);
            alloc((void**)&b->d_taild[1], b->segs.size() * (size_t)(N - hop) * 8);
            if (!rc && hipMemcpyAsync(b->d_segs, b->segs.data(), b->segs.size() * sizeof(SegDesc), hipMemcpyHostToDevice, ist) != hipSuccess) rc = GOMEL_ERR_HIP;   // b->segs outlives it
            for (int i = 0; i < 2; i++)
                if (!rc && hipMemsetAsync(b->d_taild[i], 0, b->segs.size() * (size_t)(N - hop) * 8, ist) != hipSuccess) rc = GOMEL_ERR_HIP;
            if (b->mixed_k) {   // the fp32 kernel's buffers beside the fp64 ones (same segments)
                alloc((void**)&b->d_mag, (size_t)b->total_frames * (N / 2) * 4);
                alloc((void**)&b->d_nyq, ((size_t)b->total_frames + 4) * 4);
                alloc((void**)&b->d_x[0], (size_t)b->total_len * 4);
                alloc((void**)&b->d_x[1], (size_t)b->total_len * 4);
                alloc((void**)&b->d_tail[0], b->segs.size() * (size_t)(N - hop) * 4);
                alloc((void**)&b->d_tail[1], b->segs.size() * (size_t)(N - hop) * 4);
                for (int i = 0; i < 2; i++)
                    if (!rc && hipMemsetAsync(b->d_tail[i], 0, b->segs.size() * (size_t)(N - hop) * 4, ist) != hipSuccess) rc = GOMEL_ERR_HIP;
            }
        } else {
            alloc((void**)&b->d_ywd, (size_t)b->total_frames * N * 8 * (b->persist ? 3 : 1));
            alloc((void**)&b->d_frame_xoff, (size_t)b->total_frames * 8);
            if (h->big) {
                const size_t sc = (size_t)big_chunk_frames(h, b->total_frames) * (size_t)h->dt.big.Mc * 16;
                alloc((void**)&b->d_bigA, sc);
                alloc((void**)&b->d_bigB, sc);
            }
            if (b->persist) {
                b->ctl_bytes = 16;
                alloc((void**)&b->d_ctl, b->ctl_bytes);
                alloc((void**)&b->d_frame_utt, (size_t)b->total_frames * 4);
            }
            if (b->pmix_k) {   // the fp32 launches' buffers beside the fp64 ones (both signal buffers: a persistent launch that gives up is recomputed by the generic pair)
                alloc((void**)&b->d_mag, (size_t)b->total_frames * (N / 2) * 4);
                alloc((void**)&b->d_nyq, ((size_t)b->total_frames + 4) * 4);
                alloc((void**)&b->d_x[0], (size_t)b->total_len * 4);
                alloc((void**)&b->d_x[1], (size_t)b->total_len * 4);
                alloc((void**)&b->d_yw, (size_t)b->total_frames * N * 4 * (b->persist ? 3 : 1));
            }
        }
    } else if (I > 0) {
        alloc((void**)&b->d_mag, (size_t)b->total_frames * (N / 2) * 4);
        alloc((void**)&b->d_nyq, ((size_t)b->total_frames + 4) * 4);   // + 4: lane t reads nyq[frame + (t & 3)] (load_mags_raw)
        alloc((void**)&b->d_x[0], (size_t)b->total_len * 4);
        alloc((void**)&b->d_x[1], (size_t)b->total_len * 4);
        if (!b->fast) alloc((void**)&b->d_frame_xoff, (size_t)b->total_frames * 8);
        if (!b->fast) alloc((void**)&b->d_yw, (size_t)b->total_frames * N * 4 * (b->persist ? 3 : 1));
        if (b->persist) {
            b->ctl_bytes = 16;
            alloc((void**)&b->d_ctl, b->ctl_bytes);
            alloc((void**)&b->d_frame_utt, (size_t)b->total_frames * 4);
        }
        if (b->fast) {
            plan_segments(b->h->cfg, b->utts, b->total_frames, b->segs, b->g_max);
            alloc((void**)&b->d_segs, b->segs.size() * sizeof(SegDesc));
            alloc((void**)&b->d_tail[0], b->segs.size() * (size_t)(N - hop) * 4);
            alloc((void**)&b->d_tail[1], b->segs.size() * (size_t)(N - hop) * 4);
            if (!rc && hipMemcpyAsync(b->d_segs, b->segs.data(), b->segs.size() * sizeof(SegDesc), hipMemcpyHostToDevice, ist) != hipSuccess) rc = GOMEL_ERR_HIP;   // b->segs outlives it
            for (int i = 0; i < 2; i++)  // tails of segments without a successor are never written: they must read as zeros
                if (!rc && hipMemsetAsync(b->d_tail[i], 0, b->segs.size() * (size_t)(N - hop) * 4, ist) != hipSuccess) rc = GOMEL_ERR_HIP;
        }
    }
    for (auto& e : b->ev) if (!rc && hipEventCreate(&e) != hipSuccess) rc = GOMEL_ERR_HIP;
    if (!rc && hipEventCreateWithFlags(&b->gate_ev, hipEventDisableTiming) != hipSuccess) rc = GOMEL_ERR_HIP;
    // (the per-frame index tables d_frame_xoff / d_frame_utt are filled on the device, behind the utterance descriptors: batch_upload_impl)
    if (rc) return rc;
    guard.b = nullptr;
    *out = b;
    return GOMEL_OK;
}

static int batch_create_guarded(gomel_handle* h, size_t n_utt, const size_t* n_pairs, gomel_batch** out) {
    return batch_create_common(h, n_utt, n_pairs, 0, -1, out);
}
static int batch_create_f32_guarded(gomel_handle* h, size_t n_utt, const size_t* n_pairs, gomel_batch** out) {
    return batch_create_common(h, n_utt, n_pairs, BF_IO32, -1, out);
}

// H2D of log-mel and x0 on `st` (asynchronous when the host buffers are pinned).  Pointers are to esz-byte elements.
static int batch_upload_impl(gomel_batch* b, const void* const* mel, const void* const* x0, const uint64_t* seeds, bool pin_here, bool wait,
                             hipStream_t st) {
    gomel_handle* h = b->h;
    const size_t esz = b->esz;
    b->any_x0 = false;
    for (size_t u = 0; u < b->n_utt; u++) {
        if (!mel[u]) return GOMEL_ERR_NULL;
        b->utts[u].has_x0 = (x0 && x0[u]) ? 1 : 0;
        b->utts[u].seed = seeds ? seeds[u] : (uint64_t)u;
        b->any_x0 |= b->utts[u].has_x0 != 0;
    }
    if (b->any_x0 && !b->d_x0) HIPCHK((hipError_t)h->pool.alloc((void**)&b->d_x0, (size_t)b->total_len * esz));
    PinnedScope pins;
    const size_t row = b->direct ? istft_row(h->cfg.resolut) : 2;   // elements per input unit (magnitude row / pair)
    if (pin_here)
        for (size_t u = 0; u < b->n_utt; u++) {
            pins.want(mel[u], b->n_pairs[u] * row * esz);
            if (b->utts[u].has_x0) pins.want(x0[u], (size_t)b->utts[u].len * esz);
        }
    pins.commit(st);
    int64_t poff = 0;
    for (size_t u = 0; u < b->n_utt; u++) {
        HIPCHK(hipMemcpyAsync((char*)b->d_mel + (size_t)poff * row * esz, mel[u], b->n_pairs[u] * row * esz, hipMemcpyDefault, st));
        poff += (int64_t)b->n_pairs[u];
        if (b->utts[u].has_x0)
            HIPCHK(hipMemcpyAsync((char*)b->d_x0 + (size_t)b->utts[u].xoff * esz, x0[u], (size_t)b->utts[u].len * esz, hipMemcpyDefault, st));
    }
    HIPCHK(hipMemcpyAsync(b->d_utts, b->utts.data(), b->n_utt * sizeof(UttDesc), hipMemcpyHostToDevice, st));
    if (b->d_frame_xoff || b->d_frame_utt)
        hipLaunchKernelGGL(k_frame_index, dim3((unsigned)((b->total_frames + 255) / 256)), dim3(256), 0, st, b->d_utts, (int)b->n_utt, b->total_frames, b->h->cfg.window,
                           b->d_frame_xoff, b->d_frame_utt);
    if (wait || pin_here) HIPCHK(hipStreamSynchronize(st));   // the local pins must outlive the copies
    b->uploaded = true;
    return GOMEL_OK;
}

static int batch_upload_guarded(gomel_batch* b, const void* const* mel, const void* const* x0, const uint64_t* seeds, int io32) {
    if (!b || !mel) return GOMEL_ERR_NULL;
    if (b->io32 != io32) return GOMEL_ERR_BAD_CONFIG;   // a batch created for one element type was handed the other
    std::lock_guard<std::mutex> lk(b->h->mu);
    DeviceGuard dg(b->h->device); HIPCHK(dg.err);
    return batch_upload_impl(b, mel, x0, seeds, true, true, b->h->stream);
}

static int batch_run_guarded(gomel_batch* b) {
    if (!b) return GOMEL_ERR_NULL;
    std::lock_guard<std::mutex> lk(b->h->mu);
    DeviceGuard dg(b->h->device); HIPCHK(dg.err);
    return batch_run_impl(b, nullptr);
}

static int batch_run_timed_guarded(gomel_batch* b, gomel_timing* t) {
    if (!b || !t) return GOMEL_ERR_NULL;
    std::lock_guard<std::mutex> lk(b->h->mu);
    DeviceGuard dg(b->h->device); HIPCHK(dg.err);
    return batch_run_impl(b, t);
}

int gomel_batch_sync(gomel_batch* b) {
    if (!b) return GOMEL_ERR_NULL;
    std::lock_guard<std::mutex> lk(b->h->mu);
    DeviceGuard dg(b->h->device); HIPCHK(dg.err);
    HIPCHK(hipStreamSynchronize(b->h->stream));
    const int rv = persist_verify(b);
    return rv ? rv : sync_err_check(b->h);
}

static int batch_download_guarded(gomel_batch* b, void* const* out, const size_t* out_cap, int io32) {
    if (!b || !out) return GOMEL_ERR_NULL;
    if (b->io32 != io32) return GOMEL_ERR_BAD_CONFIG;
    gomel_handle* h = b->h;
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device); HIPCHK(dg.err);
    for (size_t u = 0; u < b->n_utt; u++) {
        if (!out[u]) return GOMEL_ERR_NULL;
        if (out_cap && out_cap[u] < (size_t)b->utts[u].len) return GOMEL_ERR_BUFFER_TOO_SMALL;
    }
    if (b->persist) { HIPCHK(hipStreamSynchronize(h->stream)); int rv = persist_verify(b); if (rv) return rv; }
    PinnedScope pins;
    for (size_t u = 0; u < b->n_utt; u++) pins.want(out[u], (size_t)b->utts[u].len * b->esz);
    pins.commit(h->stream);
    for (size_t u = 0; u < b->n_utt; u++)
        HIPCHK(hipMemcpyAsync(out[u], (char*)b->d_out + (size_t)b->utts[u].xoff * b->esz, (size_t)b->utts[u].len * b->esz, hipMemcpyDefault, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return sync_err_check(h);
}

int gomel_batch_destroy(gomel_batch* b) {
    if (!b) return GOMEL_OK;
    std::lock_guard<std::mutex> lk(b->h->mu);
    DeviceGuard dg(b->h->device);
    hipStreamSynchronize(b->h->stream);
    if (b->rs) hipStreamSynchronize(b->rs);
    batch_free(b);
    return GOMEL_OK;
}

// How a host-to-host call is cut into chunks: bounds[0] = 0 < bounds[1] < ... < bounds[n_chunks] = n_utt (utterance indices).  Host
// logic only (exported as gomel_plan_chunks, tested without a GPU).  Lengths must already be validated (multiples of NumMels, > 0).
// `heavy`: the call runs fp64 iterations (precision schedule / 64-bit mode) -- its kernels take 1.75-2.4x as long per frame while its copies
// stay the same, so chunks twice the size still hide them and pay half the chunk-sized launches: 75.8 ms against 79.0 per 256 x 400-frame
// call in the default mode (scripts/chunk_sweep_default.py, profiles/r04/chunk_sweep_default.txt).
static int from_mel_batch_chunked(gomel_handle* h, size_t n_utt, const void* const* mel, const size_t* n_pairs, const void* const* x0,
                                  const uint64_t* seeds, void* const* out, const size_t* out_cap, int io32) {
    if (!h || !mel || !n_pairs || !out) return GOMEL_ERR_NULL;
    if (n_utt == 0) return GOMEL_ERR_EMPTY;
    const int K = h->cfg.num_mels;
    const size_t esz = io32 ? 4 : 8;
    int64_t total_frames = 0;
    for (size_t u = 0; u < n_utt; u++) {
        if (n_pairs[u] == 0) return GOMEL_ERR_EMPTY;
        if (n_pairs[u] % (size_t)K) return GOMEL_ERR_BAD_LENGTH;
        if (!mel[u] || !out[u]) return GOMEL_ERR_NULL;
        const size_t L = gomel_out_len(&h->cfg, n_pairs[u]);
        if (out_cap && out_cap[u] < L) return GOMEL_ERR_BUFFER_TOO_SMALL;
        total_frames += (int64_t)(n_pairs[u] / (size_t)K);
    }
    bool heavy;
    { std::lock_guard<std::mutex> lk(h->mu); heavy = resolve_fp64_iterations(h, h->cfg.griffin_lim_iterations) > 0; }
    std::vector<size_t> bounds = plan_chunks(h->cfg, n_utt, n_pairs, heavy);
    int n_chunks = (int)bounds.size() - 1;
    struct ChunkList {
        std::vector<gomel_batch*> v;
        ~ChunkList() { for (gomel_batch* b : v) if (b) gomel_batch_destroy(b); }
    } chunk_list;
    std::vector<gomel_batch*>& chunks = chunk_list.v;
    chunks.assign((size_t)n_chunks, nullptr);
    // At most three chunks hold device buffers at a time (uploading c+1, computing c, downloading c-1): chunk c-2's batch is
    // destroyed -- its buffers go back to the handle's pool -- before chunk c+1's is created, so a call of any size cycles
    // through three sets of buffers instead of allocating one per chunk.
    // Two compute streams, the chunks alternating between them, so that the tail of one chunk's launch is filled by the head of the
    // neighbour's: built and MEASURED in round 3 (profiles/r03/compute_streams.txt: 2.365 / 2.363 M frames/s against 2.379 / 2.390 M
    // on one stream, config 3, interleaved runs on one box) -- the two kernels share the chip's LDS and registers rather than its idle
    // tail.  One stream stays the default; GOMEL_COMPUTE_STREAMS=2 is the A/B knob.
    static const int n_cs = [] { const char* e = getenv("GOMEL_COMPUTE_STREAMS"); return e && atoi(e) == 2 ? 2 : 1; }();
    if (n_cs == 2 && n_chunks > 1 && !h->stream2) {
        std::lock_guard<std::mutex> lk(h->mu);
        DeviceGuard dg(h->device);
        if (!h->stream2 && hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess) { h->stream2 = nullptr; (void)hipGetLastError(); }
    }
    auto cstream = [&](int c) { return (n_cs == 2 && h->stream2 && (c & 1)) ? h->stream2 : h->stream; };
    auto create = [&](int c) {
        // (a call cut into several chunks keeps them off the persistent path: its verifying wait would stall the copy / compute pipeline --
        // two-launch persistent batches reach chunk sizes at the small frame lengths; a call that is ONE chunk takes it as before)
        int r = batch_create_common(h, bounds[c + 1] - bounds[c], n_pairs + bounds[c], (io32 ? BF_IO32 : 0) | (n_chunks > 1 ? BF_NOPERSIST : 0), -1, &chunks[c], h->up_stream);
        if (!r) chunks[c]->rs = cstream(c);
        return r;
    };
    int rc = create(0);
    std::vector<hipEvent_t> done((size_t)n_chunks, nullptr), up((size_t)n_chunks, nullptr), down((size_t)n_chunks, nullptr);
    std::vector<std::vector<uint64_t>> gseeds((size_t)n_chunks);   // default seed = GLOBAL utterance index, whatever the chunking
    {
        PinnedScope pins;
        if (!rc) {
            std::lock_guard<std::mutex> lk(h->mu);
            DeviceGuard dg(h->device);
            if (dg.err != hipSuccess) rc = GOMEL_ERR_HIP;
            for (size_t u = 0; u < n_utt && !rc; u++) {
                pins.want(mel[u], n_pairs[u] * 2 * esz);
                pins.want(out[u], gomel_out_len(&h->cfg, n_pairs[u]) * esz);
                if (x0 && x0[u]) pins.want(x0[u], gomel_out_len(&h->cfg, n_pairs[u]) * esz);
            }
            pins.commit(h->stream, h->copy_stream);
            pins.drain_too(h->up_stream);
            for (int c = 0; c < n_chunks && !rc; c++)
                if (hipEventCreateWithFlags(&done[c], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&up[c], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&down[c], hipEventDisableTiming) != hipSuccess) rc = GOMEL_ERR_HIP;
        }
        auto upload = [&](int c) {   // caller holds h->mu
            gomel_batch* b = chunks[c];
            const size_t u0 = bounds[c];
            if (!seeds) { gseeds[c].resize(b->n_utt); for (size_t u = 0; u < b->n_utt; u++) gseeds[c][u] = (uint64_t)(u0 + u); }
            int r = batch_upload_impl(b, mel + u0, x0 ? x0 + u0 : nullptr, seeds ? seeds + u0 : gseeds[c].data(), false, false, h->up_stream);
            if (!r && hipEventRecord(up[c], h->up_stream) != hipSuccess) r = GOMEL_ERR_HIP;
            return r;
        };
        if (!rc) { std::lock_guard<std::mutex> lk(h->mu); DeviceGuard dg(h->device); rc = upload(0); }
        // Order matters for keeping the GPU fed: chunk c's kernels are queued FIRST (its upload was issued an iteration ago), and
        // only then does the host wait for chunk c-2's download, recycle its buffers into chunk c+1 and queue that upload -- all of
        // which may hold the host for a millisecond or two (the waveforms of c-2 are still coming down; descriptor copies from
        // pageable memory wait for the upload stream).  The other way round (round 2's first form) the host reached the launches of
        // chunk c only after chunk c-1 had finished: 0.5 ms of idle GPU per chunk boundary in the kernel trace.
        for (int c = 0; c < n_chunks && !rc; c++) {
            {
                std::lock_guard<std::mutex> lk(h->mu);
                DeviceGuard dg(h->device);
                gomel_batch* b = chunks[c];
                const size_t u0 = bounds[c];
                if (hipStreamWaitEvent(run_stream(b), up[c], 0) != hipSuccess) rc = GOMEL_ERR_HIP;
                if (!rc) rc = batch_run_impl(b, nullptr);
                if (!rc && b->persist) { if (hipStreamSynchronize(run_stream(b)) != hipSuccess) rc = GOMEL_ERR_HIP; else rc = persist_verify(b); }
                if (!rc && (hipEventRecord(done[c], run_stream(b)) != hipSuccess || hipStreamWaitEvent(h->copy_stream, done[c], 0) != hipSuccess)) rc = GOMEL_ERR_HIP;
                for (size_t u = 0; u < b->n_utt && !rc; u++)
                    if (hipMemcpyAsync(out[u0 + u], (char*)b->d_out + (size_t)b->utts[u].xoff * esz, (size_t)b->utts[u].len * esz, hipMemcpyDefault, h->copy_stream) != hipSuccess)
                        rc = GOMEL_ERR_HIP;
                if (!rc && hipEventRecord(down[c], h->copy_stream) != hipSuccess) rc = GOMEL_ERR_HIP;
            }
            if (rc || c + 1 >= n_chunks) continue;
            if (c >= 2) {   // chunk c-2 has computed; once its waveforms are down its buffers serve chunk c+1
                DeviceGuard dg(h->device);
                if (hipEventSynchronize(down[c - 2]) != hipSuccess) { rc = GOMEL_ERR_HIP; break; }
                // (not gomel_batch_destroy: that drains the compute stream, i.e. waits for chunks c-1 and c as well; everything that
                // ever touched this batch's buffers finished before its download did)
                std::lock_guard<std::mutex> lk(h->mu);
                batch_free(chunks[c - 2]); chunks[c - 2] = nullptr;
            }
            rc = create(c + 1);                                                       // (takes and drops h->mu itself)
            if (rc) break;
            std::lock_guard<std::mutex> lk(h->mu);
            DeviceGuard dg(h->device);
            rc = upload(c + 1);                                                        // runs beside chunk c's (and c-1's) kernels
        }
        {
            DeviceGuard dg(h->device);
            if (hipStreamSynchronize(h->up_stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess || (h->stream2 && hipStreamSynchronize(h->stream2) != hipSuccess) ||
                hipStreamSynchronize(h->copy_stream) != hipSuccess) rc = rc ? rc : GOMEL_ERR_HIP;
            if (rc == GOMEL_ERR_HIP) g_last_hip = (int)hipGetLastError();
            for (hipEvent_t e : done) if (e) hipEventDestroy(e);
            for (hipEvent_t e : up) if (e) hipEventDestroy(e);
            for (hipEvent_t e : down) if (e) hipEventDestroy(e);
        }
    }   // pins dropped here, after every stream has drained
    if (!rc) { std::lock_guard<std::mutex> lk(h->mu); rc = sync_err_check(h); }
    return rc;
}

// ----------------------------------------------------------------------------------------------- jobs: calls in flight
// A chunked call hides its copies behind its own kernels, but pays for that with chunk-sized launches (12 800 frames where the
// whole batch has 102 400: 0.365 instead of 0.40 of the HBM roofline) and still exposes its first upload and last download.  With
// MORE THAN ONE call in flight none of that is needed: every call is ONE whole batch -- upload on the upload stream, all its
// kernels in whole-batch launches on the compute stream, download on the copy stream -- and the copies of call n+1 / n-1 run
// beside the kernels of call n.  gomel_from_mel_batch_submit queues a call like that and returns; gomel_job_wait blocks until its
// waveforms are in the caller's buffers.  Two synchronous gomel_from_mel_batch calls that overlap on one handle (two goroutines)
// on a handle in whole-batch call mode (gomel_set_call_mode) are run the same way.
struct gomel_job {
    gomel_handle* h = nullptr;
    gomel_batch* b = nullptr;
    PinnedScope pins;                 // the caller's buffers stay pinned until the job has been waited for
    hipEvent_t up = nullptr, done = nullptr, down = nullptr;
    int status = GOMEL_OK;            // of a job that ran to completion inside submit (a batch too small for the fused kernels)
    bool completed = false;
};

static void job_drop(gomel_job* j) {  // caller has made sure nothing of the job is still in flight
    if (j->b) { std::lock_guard<std::mutex> lk(j->h->mu); batch_free(j->b); j->b = nullptr; }
    for (hipEvent_t e : {j->up, j->done, j->down}) if (e) hipEventDestroy(e);
    j->h->jobs_outstanding--;
    delete j;                         // releases the pins (no stream is drained: the caller has done what was needed)
}

static bool whole_batch_pays(const gomel_handle* h, size_t n_utt, const size_t* n_pairs) {
    if (!fused_applicable(h->cfg) || h->path_mode == 1 || h->path_mode == 3) return false;
    int64_t frames = 0;
    for (size_t u = 0; u < n_utt; u++) frames += (int64_t)(n_pairs[u] / (size_t)h->cfg.num_mels);
    return frames >= 2 * fused_min_frames(h->cfg, resolve_fp64_iterations(h, h->cfg.griffin_lim_iterations) > 0);   // below that a call is a handful of launches: nothing to pipeline
}

static int submit_guarded(gomel_handle* h, size_t n_utt, const void* const* mel, const size_t* n_pairs, const void* const* x0,
                          const uint64_t* seeds, void* const* out, const size_t* out_cap, int io32, gomel_job** job_out) {
    if (!h || !mel || !n_pairs || !out || !job_out) return GOMEL_ERR_NULL;
    *job_out = nullptr;
    if (n_utt == 0) return GOMEL_ERR_EMPTY;
    const int K = h->cfg.num_mels;
    const size_t esz = io32 ? 4 : 8;
    for (size_t u = 0; u < n_utt; u++) {
        if (n_pairs[u] == 0) return GOMEL_ERR_EMPTY;
        if (n_pairs[u] % (size_t)K) return GOMEL_ERR_BAD_LENGTH;
        if (!mel[u] || !out[u]) return GOMEL_ERR_NULL;
        if (out_cap && out_cap[u] < gomel_out_len(&h->cfg, n_pairs[u])) return GOMEL_ERR_BUFFER_TOO_SMALL;
    }
    gomel_job* j = new (std::nothrow) gomel_job();
    if (!j) return GOMEL_ERR_NO_MEMORY;
    j->h = h;
    h->jobs_outstanding++;
    if (!whole_batch_pays(h, n_utt, n_pairs)) {   // small: run it now, hand back a finished job
        j->status = from_mel_batch_chunked(h, n_utt, mel, n_pairs, x0, seeds, out, out_cap, io32);
        j->completed = true;
        *job_out = j;
        return GOMEL_OK;
    }
    int rc = batch_create_common(h, n_utt, n_pairs, (io32 ? BF_IO32 : 0) | BF_NOPERSIST, -1, &j->b, h->up_stream);   // (since two-launch persistent batches reach above the
    // fused kernels' threshold at the small frame lengths, the job says so itself: whole-batch fused / generic launches only)
    if (!rc && j->b->persist) rc = GOMEL_ERR_BAD_CONFIG;
    bool queued = false;
    if (!rc) {
        std::lock_guard<std::mutex> lk(h->mu);
        DeviceGuard dg(h->device);
        if (dg.err != hipSuccess) rc = GOMEL_ERR_HIP;
        gomel_batch* b = j->b;
        b->rs = h->stream;
        for (size_t u = 0; u < n_utt && !rc; u++) {
            j->pins.want(mel[u], n_pairs[u] * 2 * esz);
            j->pins.want(out[u], (size_t)b->utts[u].len * esz);
            if (x0 && x0[u]) j->pins.want(x0[u], (size_t)b->utts[u].len * esz);
        }
        if (!rc) j->pins.commit();   // no stream to drain at release: gomel_job_wait has seen the job's last copy finish
        for (hipEvent_t* e : {&j->up, &j->done, &j->down})
            if (!rc && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = GOMEL_ERR_HIP;
        if (!rc) { queued = true; rc = batch_upload_impl(b, mel, x0, seeds, false, false, h->up_stream); }   // default seeds: the utterance index within the call
        if (!rc && (hipEventRecord(j->up, h->up_stream) != hipSuccess || hipStreamWaitEvent(h->stream, j->up, 0) != hipSuccess)) rc = GOMEL_ERR_HIP;
        if (!rc) rc = batch_run_impl(b, nullptr);
        if (!rc && (hipEventRecord(j->done, h->stream) != hipSuccess || hipStreamWaitEvent(h->copy_stream, j->done, 0) != hipSuccess)) rc = GOMEL_ERR_HIP;
        for (size_t u = 0; u < n_utt && !rc; u++)
            if (hipMemcpyAsync(out[u], (char*)b->d_out + (size_t)b->utts[u].xoff * esz, (size_t)b->utts[u].len * esz, hipMemcpyDefault, h->copy_stream) != hipSuccess) rc = GOMEL_ERR_HIP;
        if (!rc && hipEventRecord(j->down, h->copy_stream) != hipSuccess) rc = GOMEL_ERR_HIP;
        if (rc == GOMEL_ERR_HIP) g_last_hip = (int)hipGetLastError();
        if (rc && queued) {   // something of this job may be in flight: drain before its buffers and pins go
            (void)hipStreamSynchronize(h->up_stream); (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->copy_stream);
            (void)hipGetLastError();
        }
    }
    if (rc) { job_drop(j); return rc; }
    *job_out = j;
    return GOMEL_OK;
}

// timeout_ms < 0: wait without bound (gomel_job_wait); otherwise poll until the deadline and hand the job back alive (GOMEL_ERR_BUSY)
static int job_wait_guarded(gomel_job* j, int64_t timeout_ms = -1) {
    if (!j) return GOMEL_ERR_NULL;
    int rc = j->status;
    if (!j->completed) {
        DeviceGuard dg(j->h->device);
        int q = 0;
        if (dg.err == hipSuccess && timeout_ms >= 0) {
            q = wait_event_bounded<HipEvOps>(j->down, timeout_ms);
            if (q == 1) return GOMEL_ERR_BUSY;                          // still running: nothing is freed, the job stays valid
        }
        if (dg.err != hipSuccess || q < 0 || (timeout_ms < 0 && hipEventSynchronize(j->down) != hipSuccess)) {
            g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP;
            (void)hipStreamSynchronize(j->h->up_stream); (void)hipStreamSynchronize(j->h->stream); (void)hipStreamSynchronize(j->h->copy_stream);
            (void)hipGetLastError();
        }
        if (!rc) { std::lock_guard<std::mutex> lk(j->h->mu); rc = sync_err_check(j->h); }   // (handle-wide: with several jobs in flight the first wait behind a give-up reports it)
    }
    job_drop(j);
    return rc;
}

// The synchronous call.  By default it is cut into chunks that hide their copies behind each other's kernels: the best a LONE caller
// can get.  A host that overlaps calls on one handle (two goroutines calling in a loop) switches the handle to whole-batch calls
// (gomel_set_call_mode(h, 1)): every call is then one submit + wait, and the copies of one call run beside the whole-batch kernels
// of the other.  The mode is the caller's choice, never guessed from what happens to be in flight: the two modes cut an utterance
// into different segments (other summation order of the overlap-add), so their results differ at rounding level, and what a call
// returns must not depend on timing.
static int from_mel_batch_guarded(gomel_handle* h, size_t n_utt, const void* const* mel, const size_t* n_pairs, const void* const* x0,
                                  const uint64_t* seeds, void* const* out, const size_t* out_cap, int io32) {
    if (!h || !mel || !n_pairs || !out) return GOMEL_ERR_NULL;
    if (h->call_mode.load() == 1) {
        gomel_job* j = nullptr;
        const int rc = submit_guarded(h, n_utt, mel, n_pairs, x0, seeds, out, out_cap, io32, &j);   // (a small call runs chunked inside)
        return rc ? rc : job_wait_guarded(j);
    }
    return from_mel_batch_chunked(h, n_utt, mel, n_pairs, x0, seeds, out, out_cap, io32);
}

// One call, several GPUs: contiguous utterance shards balanced by frame count, one host thread per handle (a handle = one
// device), no data-path collective.  What a Go host uses to drive the 8 GPUs of a node without Python; every
// handle must have been created from the same configuration (use gomel_create_from_tables for bit-identical tables).
// The calling host thread onto the CPUs of the NUMA node `device` hangs off (a fresh per-shard thread of gomel_from_mel_batch_multi;
// one process drives several GPUs there, so a process-wide binding as the Python ranks do it would be wrong).  No NUMA information,
// GOMEL_NUMA_BIND=0, or a node outside the thread's mask: the thread stays where it is.  GOMEL_SYSFS_ROOT: another sysfs tree (tests).
static void bind_thread_to_device_node(int device) {
    static const bool enabled = [] { const char* e = getenv("GOMEL_NUMA_BIND"); return !(e && *e == '0'); }();
    if (!enabled) return;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf - 1, device) != hipSuccess) { (void)hipGetLastError(); return; }
    (void)gomel_numa_bind_thread(getenv("GOMEL_SYSFS_ROOT"), bdf, nullptr, nullptr);
}
static int from_mel_batch_multi_guarded(gomel_handle* const* hs, size_t n_handles, size_t n_utt, const void* const* mel, const size_t* n_pairs,
                                        const void* const* x0, const uint64_t* seeds, void* const* out, const size_t* out_cap, int io32) {
    if (!hs || !mel || !n_pairs || !out) return GOMEL_ERR_NULL;
    if (n_handles == 0 || n_utt == 0) return GOMEL_ERR_EMPTY;
    for (size_t i = 0; i < n_handles; i++) {
        if (!hs[i]) return GOMEL_ERR_NULL;
        const gomel_config &a = hs[0]->cfg, &c = hs[i]->cfg;
        if (a.num_mels != c.num_mels || a.window != c.window || a.resolut != c.resolut || a.griffin_lim_iterations != c.griffin_lim_iterations ||
            a.mel_fmin != c.mel_fmin || a.mel_fmax != c.mel_fmax || a.tune_mul != c.tune_mul || a.tune_add != c.tune_add) return GOMEL_ERR_BAD_CONFIG;
    }
    const int K = hs[0]->cfg.num_mels;
    std::vector<int64_t> pre(n_utt + 1, 0);
    for (size_t u = 0; u < n_utt; u++) {
        if (n_pairs[u] == 0) return GOMEL_ERR_EMPTY;
        if (n_pairs[u] % (size_t)K) return GOMEL_ERR_BAD_LENGTH;
        pre[u + 1] = pre[u] + (int64_t)(n_pairs[u] / (size_t)K);
    }
    std::vector<size_t> bounds(n_handles + 1, 0);
    for (size_t r = 1; r < n_handles; r++) {   // greedy on the prefix sum, as gomel_amd.dist.shard_bounds
        const double target = (double)pre[n_utt] * (double)r / (double)n_handles;
        size_t i = (size_t)(std::lower_bound(pre.begin(), pre.end(), (int64_t)std::ceil(target)) - pre.begin());
        if (i > 0 && std::fabs((double)pre[i - 1] - target) <= std::fabs((double)pre[std::min(i, n_utt)] - target)) i--;
        bounds[r] = std::min(std::max(i, bounds[r - 1]), n_utt);
    }
    bounds[n_handles] = n_utt;
    std::vector<uint64_t> gseeds;
    if (!seeds) { gseeds.resize(n_utt); for (size_t u = 0; u < n_utt; u++) gseeds[u] = (uint64_t)u; seeds = gseeds.data(); }   // global index: sharding does not change results
    std::vector<int> rcs(n_handles, GOMEL_OK);
    std::vector<std::thread> th;
    for (size_t r = 0; r < n_handles; r++) {
        const size_t lo = bounds[r], hi = bounds[r + 1];
        if (hi <= lo) continue;
        th.emplace_back([&, r, lo, hi] {
            try {
                bind_thread_to_device_node(hs[r]->device);   // this shard's pinning and copy submission run beside its GPU's root port
                rcs[r] = from_mel_batch_guarded(hs[r], hi - lo, mel + lo, n_pairs + lo, x0 ? x0 + lo : nullptr, seeds + lo, out + lo, out_cap ? out_cap + lo : nullptr, io32);
            } catch (...) { rcs[r] = GOMEL_ERR_NO_MEMORY; }
        });
    }
    for (std::thread& t : th) t.join();
    for (int r : rcs) if (r) return r;
    return GOMEL_OK;
}

int gomel_from_mel(gomel_handle* h, const double* mel, size_t n_pairs, const double* x0, uint64_t seed, double* out, size_t out_cap,
                   size_t* out_len) {
    if (!h || !mel || !out) return GOMEL_ERR_NULL;
    if (n_pairs == 0) return GOMEL_ERR_EMPTY;
    if (n_pairs % (size_t)h->cfg.num_mels) return GOMEL_ERR_BAD_LENGTH;
    const size_t L = gomel_out_len(&h->cfg, n_pairs);
    if (out_cap < L) return GOMEL_ERR_BUFFER_TOO_SMALL;
    const double* mels[1] = {mel};
    const double* x0s[1] = {x0};
    double* outs[1] = {out};
    int rc = gomel_from_mel_batch(h, 1, mels, &n_pairs, x0s, &seed, outs, &out_cap);
    if (!rc && out_len) *out_len = L;
    return rc;
}

// ISTFT as an entry point of its own: Griffin-Lim from linear magnitudes, `iterations` = numIterations.  With `spec_out` the
// call also reproduces the reference's side effect on its spectrogram argument (mel/mel.go:98-108), bins 0 ... N/2.
static int istft_guarded(gomel_handle* h, const double* mags, size_t n_frames, int32_t iterations, const double* x0, uint64_t seed,
                         double* out, size_t out_cap, size_t* out_len, double* spec_out) {
    if (!h || !mags || !out) return GOMEL_ERR_NULL;
    if (n_frames == 0) return GOMEL_ERR_EMPTY;
    if (iterations < 0) return GOMEL_ERR_BAD_CONFIG;
    const size_t L = (size_t)h->cfg.resolut + (n_frames - 1) * (size_t)h->cfg.window;   // mel/mel.go:80
    if (out_cap < L) return GOMEL_ERR_BUFFER_TOO_SMALL;
    gomel_batch* b = nullptr;
    const bool spec = spec_out && iterations > 0;   // zero iterations: the loop never runs, the spectrogram is left as it was
    int rc = batch_create_common(h, 1, &n_frames, BF_DIRECT | (spec ? BF_SPEC : 0), iterations, &b);
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        DeviceGuard dg(h->device);
        if (dg.err != hipSuccess) rc = GOMEL_ERR_HIP;
        const void* ins[1] = {mags};
        const void* x0s[1] = {x0};
        if (!rc) rc = batch_upload_impl(b, ins, x0s, &seed, true, true, h->stream);
        if (!rc) rc = batch_run_impl(b, nullptr);
        if (!rc && spec) {
            const size_t bytes = n_frames * istft_row(h->cfg.resolut) * 16;
            PinnedScope pins;
            pins.want(spec_out, bytes); pins.commit(h->stream);
            if (hipMemcpyAsync(spec_out, b->d_spec, bytes, hipMemcpyDefault, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
                g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP;
            }
        }
    }
    double* outs[1] = {out};
    if (!rc) rc = gomel_batch_download(b, outs, &out_cap);
    gomel_batch_destroy(b);
    if (!rc && out_len) *out_len = L;
    return rc;
}

int gomel_rfft_frames(gomel_handle* h, const float* frames, size_t n_frames, float* spectrum) { return fft_hook(h, false, frames, n_frames, spectrum); }
int gomel_irfft_frames(gomel_handle* h, const float* spectrum, size_t n_frames, float* frames) { return fft_hook(h, true, spectrum, n_frames, frames); }

static int target_magnitudes_guarded(gomel_handle* h, const double* mel, size_t n_pairs, float* mag_out) {
    if (!h || !mel || !mag_out) return GOMEL_ERR_NULL;
    if (h->generic) return GOMEL_ERR_UNSUPPORTED_N;   // register FFT cores only (Resolut 256 ... 4096, powers of two)
    if (n_pairs == 0) return GOMEL_ERR_EMPTY;
    const int K = h->cfg.num_mels, N = h->cfg.resolut, H = N / 2;
    if (n_pairs % (size_t)K) return GOMEL_ERR_BAD_LENGTH;
    const size_t F = n_pairs / (size_t)K;  // (rows come back in lane-slot order and are un-permuted below)
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device); HIPCHK(dg.err);
    GateOrdinary gate_o; gate_o.enter(h, h->stream);   // synchronous entry point: the gate is left (destructor) after its streams have drained
    double* d_mel = nullptr; float *d_mag = nullptr, *d_nyq = nullptr;
    int rc = GOMEL_OK;
    if (h->pool.alloc((void**)&d_mel, n_pairs * 16) != hipSuccess || h->pool.alloc((void**)&d_mag, F * H * 4) != hipSuccess ||
        h->pool.alloc((void**)&d_nyq, F * 4) != hipSuccess) rc = GOMEL_ERR_HIP;
    if (!rc && hipMemcpyAsync(d_mel, mel, n_pairs * 16, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = GOMEL_ERR_HIP;
    if (!rc) {
        if (launch_mel_to_mag<float>(N, (int64_t)F, h->stream, d_mel, h->dt.undo, d_mag, d_nyq, K, h->cfg.tune_add, h->cfg.tune_mul) != hipSuccess) rc = GOMEL_ERR_HIP;
        std::vector<float> mg(F * H), ny(F);
        if (hipMemcpyAsync(mg.data(), d_mag, F * H * 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipMemcpyAsync(ny.data(), d_nyq, F * 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) rc = GOMEL_ERR_HIP;
        if (!rc)
            for (size_t f = 0; f < F; f++) {
                for (int k = 0; k < H; k++) {
                    int slot = 0;
                    GOMEL_DISPATCH_N(N, slot = mag_slot_index<Geo<NN>>(k));
                    mag_out[f * (H + 1) + k] = mg[f * H + slot];
                }
                mag_out[f * (H + 1) + H] = ny[f];
            }
    }
    if (d_mel) h->pool.release(d_mel);
    if (d_mag) h->pool.release(d_mag);
    if (d_nyq) h->pool.release(d_nyq);
    return rc;
}

static int to_mel_batch_guarded(gomel_handle* h, size_t n_utt, const double* const* wave, const size_t* n_samples, double* const* out_pairs,
                                const size_t* cap_pairs, size_t* n_pairs_out) {
    if (!h || !wave || !n_samples || !out_pairs) return GOMEL_ERR_NULL;
    if (n_utt == 0) return GOMEL_ERR_EMPTY;
    const gomel_config& c = h->cfg;
    const int N = c.resolut, H = N / 2, K = c.num_mels, hop = c.window;
    if (h->dom_status) return h->dom_status;
    std::vector<size_t> P(n_utt), F(n_utt), woff(n_utt), foff(n_utt);
    size_t tot_w = 0, tot_f = 0;
    for (size_t u = 0; u < n_utt; u++) {
        if (!wave[u] || !out_pairs[u]) return GOMEL_ERR_NULL;
        if (n_samples[u] == 0) return GOMEL_ERR_EMPTY;
        P[u] = gomel_pad_len(n_samples[u], hop);                                  // pad, mel/impl.go:429-455
        if (P[u] < (size_t)N) return GOMEL_ERR_TOO_SHORT;
        const size_t pairs = gomel_to_mel_pairs(&c, n_samples[u]);                // gossp NumFrames * NumMels
        F[u] = pairs / (size_t)K;
        if (cap_pairs && cap_pairs[u] < pairs) return GOMEL_ERR_BUFFER_TOO_SMALL;
        woff[u] = tot_w; foff[u] = tot_f;
        tot_w += (P[u] + 1) & ~(size_t)1;                                         // 16-byte aligned starts (double2 loads)
        tot_f += F[u];
    }
    std::vector<int64_t> fx(tot_f);
    for (size_t u = 0; u < n_utt; u++)
        for (size_t f = 0; f < F[u]; f++) fx[foff[u] + f] = (int64_t)(woff[u] + f * (size_t)hop);
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device); HIPCHK(dg.err);
    GateOrdinary gate_o; gate_o.enter(h, h->stream);   // synchronous entry point: the gate is left (destructor) after its streams have drained
    double *d_wave = nullptr, *d_smag = nullptr, *d_out = nullptr;
    v2d *d_bigA = nullptr, *d_bigB = nullptr;   // frame lengths above 8192: the convolution's scratch (gl_big.hpp)
    int64_t* d_fx = nullptr;
    int rc = GOMEL_OK;
    PinnedScope pins;                      // big caller buffers: pinned for the call (pageable copies run at a tenth of the rate)
    for (size_t u = 0; u < n_utt; u++) { pins.want(wave[u], n_samples[u] * 8); pins.want(out_pairs[u], F[u] * (size_t)K * 16); }
    pins.commit(h->stream, h->copy_stream);
    pins.drain_too(h->up_stream);
    if (h->pool.alloc((void**)&d_wave, tot_w * 8) != hipSuccess || h->pool.alloc((void**)&d_smag, tot_f * (size_t)(H + 1) * 8) != hipSuccess ||
        h->pool.alloc((void**)&d_out, tot_f * (size_t)K * 16) != hipSuccess || h->pool.alloc((void**)&d_fx, tot_f * 8) != hipSuccess) {
        g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP;
    }
    if (!rc && h->big) {
        const size_t sc = (size_t)big_chunk_frames(h, (int64_t)tot_f) * (size_t)h->dt.big.Mc * 16;
        if (h->pool.alloc((void**)&d_bigA, sc) != hipSuccess || h->pool.alloc((void**)&d_bigB, sc) != hipSuccess) { g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP; }
    }
    // Chunks of ~32 MB of samples, pipelined over the handle's three streams as in FromMel's batch call: chunk c+1's waveforms go up
    // while chunk c is transformed and chunk c-1's log-mel comes down (a 256 x 10 s call is H2D-bound: 0.9 GB up, 0.27 GB down,
    // 1.5 ms of kernels -- 28.7 ms one thing after the other, the upload's 18 ms plus a chunk's worth overlapped).  The device
    // buffers cover the whole call; only the streams are staged.
    std::vector<size_t> cb(1, 0);
    {
        size_t acc = 0;
        for (size_t u = 0; u < n_utt; u++) {
            acc += n_samples[u] * 8;
            if (acc >= ((size_t)32 << 20) && u + 1 < n_utt) { cb.push_back(u + 1); acc = 0; }
        }
        cb.push_back(n_utt);
    }
    const size_t n_ch = cb.size() - 1;
    std::vector<hipEvent_t> ev_up(n_ch, nullptr), ev_done(n_ch, nullptr);
    for (size_t cidx = 0; cidx < n_ch && !rc; cidx++)
        if (hipEventCreateWithFlags(&ev_up[cidx], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_done[cidx], hipEventDisableTiming) != hipSuccess) rc = GOMEL_ERR_HIP;
    // (the pins were committed against the compute stream: the copies of this call run on the other two, drained below before the pins go)
    if (!rc && hipMemsetAsync(d_wave, 0, tot_w * 8, h->up_stream) != hipSuccess) rc = GOMEL_ERR_HIP;          // pad's zeros (mel/impl.go:429-455)
    if (!rc && hipMemcpyAsync(d_fx, fx.data(), tot_f * 8, hipMemcpyHostToDevice, h->up_stream) != hipSuccess) rc = GOMEL_ERR_HIP;
    for (size_t cidx = 0; cidx < n_ch && !rc; cidx++) {
        const size_t u0 = cb[cidx], u1 = cb[cidx + 1];
        const size_t f0 = foff[u0], nf = (u1 < n_utt ? foff[u1] : tot_f) - f0;
        for (size_t u = u0; u < u1 && !rc; u++)
            if (hipMemcpyAsync(d_wave + woff[u], wave[u], n_samples[u] * 8, hipMemcpyDefault, h->up_stream) != hipSuccess) rc = GOMEL_ERR_HIP;
        if (!rc && (hipEventRecord(ev_up[cidx], h->up_stream) != hipSuccess || hipStreamWaitEvent(h->stream, ev_up[cidx], 0) != hipSuccess)) rc = GOMEL_ERR_HIP;
        if (rc) break;
        if (h->big)
            big_frames(h, h->stream, 2, d_wave, d_fx + f0, (int64_t)nf, d_bigA, d_bigB, nullptr, nullptr, nullptr, d_smag + f0 * (size_t)(H + 1));
        else if (h->generic)
            hipLaunchKernelGGL(glany::k_any_frames<glany::kAnyToMel>, dim3((unsigned)nf), dim3(glany::kAnyThreads), (size_t)h->dt.any.M * 16, h->stream,
                               d_wave, d_fx + f0, (const double*)nullptr, (const double*)nullptr, d_smag + f0 * (size_t)(H + 1), (int64_t)nf, N, h->dt.windowd, h->dt.any);
        else
        GOMEL_DISPATCH_N(N, { constexpr int SPB = 128 / Geo<NN>::T;   /* frame streams per 128-thread block */
                              hipLaunchKernelGGL(k_stft_mag<NN>, dim3((unsigned)((nf + SPB - 1) / SPB)), dim3(128), 0, h->stream, d_wave, d_fx + f0, d_smag + f0 * (size_t)(H + 1), (int64_t)nf, hop, h->dt); });
        if (h->big) hipLaunchKernelGGL(k_domel_log, dim3((unsigned)nf), dim3(256), 0, h->stream, d_smag + f0 * (size_t)(H + 1), h->dt.dom, d_out + f0 * (size_t)K * 2, K, H, 0);
        else hipLaunchKernelGGL(k_domel_log, dim3((unsigned)nf), dim3(256), (size_t)(H + 1) * 8, h->stream, d_smag + f0 * (size_t)(H + 1), h->dt.dom, d_out + f0 * (size_t)K * 2, K, H, 1);
        if (hipEventRecord(ev_done[cidx], h->stream) != hipSuccess || hipStreamWaitEvent(h->copy_stream, ev_done[cidx], 0) != hipSuccess) rc = GOMEL_ERR_HIP;
        for (size_t u = u0; u < u1 && !rc; u++)
            if (hipMemcpyAsync(out_pairs[u], d_out + foff[u] * (size_t)K * 2, F[u] * (size_t)K * 16, hipMemcpyDefault, h->copy_stream) != hipSuccess) rc = GOMEL_ERR_HIP;
    }
    if (hipStreamSynchronize(h->up_stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess || hipStreamSynchronize(h->copy_stream) != hipSuccess)
        rc = rc ? rc : GOMEL_ERR_HIP;   // also: fx must outlive its copy
    if (rc == GOMEL_ERR_HIP) g_last_hip = (int)hipGetLastError();
    for (hipEvent_t e : ev_up) if (e) hipEventDestroy(e);
    for (hipEvent_t e : ev_done) if (e) hipEventDestroy(e);
    if (d_wave) h->pool.release(d_wave);
    if (d_smag) h->pool.release(d_smag);
    if (d_out) h->pool.release(d_out);
    if (d_fx) h->pool.release(d_fx);
    if (d_bigA) h->pool.release(d_bigA);
    if (d_bigB) h->pool.release(d_bigB);
    if (!rc && n_pairs_out) for (size_t u = 0; u < n_utt; u++) n_pairs_out[u] = F[u] * (size_t)K;
    return rc;
}

static int to_mel_guarded(gomel_handle* h, const double* wave, size_t n, double* out_pairs, size_t cap_pairs, size_t* n_pairs) {
    if (!h || !wave || !out_pairs) return GOMEL_ERR_NULL;
    if (n == 0) return GOMEL_ERR_EMPTY;
    size_t got = 0;
    const int rc = to_mel_batch_guarded(h, 1, &wave, &n, &out_pairs, &cap_pairs, &got);
    if (!rc && n_pairs) *n_pairs = got;
    return rc;
}

// ----------------------------------------------------------------------------------------------- package phase
static int to_phase_guarded(gomel_handle* h, int32_t num_freqs, const double* wave, size_t n, double* out_pairs, size_t cap_pairs, size_t* n_pairs) {
    if (!h || !wave || !out_pairs) return GOMEL_ERR_NULL;
    if (h->generic) return GOMEL_ERR_UNSUPPORTED_N;   // register FFT cores only (Resolut 256 ... 4096, powers of two)
    const gomel_config& c = h->cfg;
    const int N = c.resolut, hop = c.window;
    if (n == 0) return GOMEL_ERR_EMPTY;
    if (num_freqs <= 0) return GOMEL_ERR_BAD_CONFIG;
    const size_t P = gomel_pad_len(n, hop);
    if (P < (size_t)N) return GOMEL_ERR_TOO_SHORT;
    const int nf = std::min(num_freqs, N / 2);                     // shrink keeps j < omels, at most all N/2 pairs
    const size_t pairs = gomel_phase_pairs(&c, num_freqs, n), F = pairs / (size_t)nf;
    if (cap_pairs < pairs) return GOMEL_ERR_BUFFER_TOO_SMALL;
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device); HIPCHK(dg.err);
    GateOrdinary gate_o; gate_o.enter(h, h->stream);   // synchronous entry point: the gate is left (destructor) after its streams have drained
    double *d_wave = nullptr, *d_out = nullptr;
    int rc = GOMEL_OK;
    PinnedScope pins;
    pins.want(wave, n * 8);
    pins.want(out_pairs, pairs * 16);
    pins.commit(h->stream);
    if (h->pool.alloc((void**)&d_wave, P * 8) != hipSuccess || h->pool.alloc((void**)&d_out, pairs * 16) != hipSuccess) { g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP; }
    if (!rc && (hipMemsetAsync(d_wave, 0, P * 8, h->stream) != hipSuccess ||
                hipMemcpyAsync(d_wave, wave, n * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess)) rc = GOMEL_ERR_HIP;
    if (!rc) {
        GOMEL_DISPATCH_N(N, { constexpr int SPB = 128 / Geo<NN>::T;   /* frame streams per 128-thread block */
                              hipLaunchKernelGGL(k_phase_stft<NN>, dim3((unsigned)((F + SPB - 1) / SPB)), dim3(128), 0, h->stream, d_wave, d_out, (int64_t)F, hop, nf, h->dt); });
        if (hipMemcpyAsync(out_pairs, d_out, pairs * 16, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP; }
    }
    if (d_wave) h->pool.release(d_wave);
    if (d_out) h->pool.release(d_out);
    if (!rc && n_pairs) *n_pairs = pairs;
    return rc;
}

static int from_phase_guarded(gomel_handle* h, int32_t num_freqs, double volume_boost, const double* pairs, size_t n_pairs, double* out, size_t out_cap,
                     size_t* out_len) {
    if (!h || !pairs || !out) return GOMEL_ERR_NULL;
    if (h->generic) return GOMEL_ERR_UNSUPPORTED_N;
    const gomel_config& c = h->cfg;
    const int N = c.resolut, hop = c.window;
    if (n_pairs == 0) return GOMEL_ERR_EMPTY;
    if (num_freqs <= 0 || num_freqs > N / 2) return GOMEL_ERR_BAD_CONFIG;
    if (n_pairs % (size_t)num_freqs) return GOMEL_ERR_BAD_LENGTH;
    const size_t F = n_pairs / (size_t)num_freqs, L = (size_t)N + (F - 1) * (size_t)hop;
    if (out_cap < L) return GOMEL_ERR_BUFFER_TOO_SMALL;
    // half the maximum of the window-square sum (phase/phase.go:117-123): data-independent, same accumulation order as the kernel
    double mx = 0.0;
    {
        std::vector<double> ws(L, 0.0);
        for (size_t i = 0; i < F; i++)
            for (int j = 0; j < N; j++) ws[i * hop + j] += h->window[j] * h->window[j];
        for (double v : ws) mx = std::max(mx, v);
    }
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard dg(h->device); HIPCHK(dg.err);
    GateOrdinary gate_o; gate_o.enter(h, h->stream);   // synchronous entry point: the gate is left (destructor) after its streams have drained
    double *d_in = nullptr, *d_yw = nullptr, *d_out = nullptr;
    int rc = GOMEL_OK;
    PinnedScope pins;
    pins.want(pairs, n_pairs * 16);
    pins.want(out, L * 8);
    pins.commit(h->stream);
    if (h->pool.alloc((void**)&d_in, n_pairs * 16) != hipSuccess || h->pool.alloc((void**)&d_yw, F * (size_t)N * 8) != hipSuccess ||
        h->pool.alloc((void**)&d_out, L * 8) != hipSuccess) { g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP; }
    if (!rc && hipMemcpyAsync(d_in, pairs, n_pairs * 16, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = GOMEL_ERR_HIP;
    if (!rc) {
        GOMEL_DISPATCH_N(N, { constexpr int SPB = 128 / Geo<NN>::T;   /* frame streams per 128-thread block */
                              hipLaunchKernelGGL(k_phase_istft<NN>, dim3((unsigned)((F + SPB - 1) / SPB)), dim3(128), 0, h->stream, d_in, d_yw, (int64_t)F, num_freqs, h->dt); });
        const unsigned g = (unsigned)std::min<size_t>((L + 255) / 256, 65536);
        hipLaunchKernelGGL(k_phase_ola, dim3(g), dim3(256), 0, h->stream, d_yw, d_out, (int64_t)L, (int)F, N, hop, h->dt.windowd, mx * 0.5, volume_boost);
        if (hipMemcpyAsync(out, d_out, L * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
            g_last_hip = (int)hipGetLastError(); rc = GOMEL_ERR_HIP;
        }
    }
    if (d_in) h->pool.release(d_in);
    if (d_yw) h->pool.release(d_yw);
    if (d_out) h->pool.release(d_out);
    if (!rc && out_len) *out_len = L;
    return rc;
}

// Diagnostic (not in the header; the tests reach it by name): the persistent path's pause state of a handle, and a way to change the
// spin limit of a LIVE handle (GOMEL_PERSIST_TICKS is read at creation) so that a test can force a timeout and then let launches
// complete again.  set_ticks < 0 leaves the limit alone.
int gomel_debug_persist_state(gomel_handle* h, long long set_ticks, unsigned* backoff, unsigned long long* paused_for) {
    if (!h) return GOMEL_ERR_NULL;
    std::lock_guard<std::mutex> lk(h->mu);
    if (set_ticks >= 0) h->persist_ticks = (unsigned)set_ticks;
    if (backoff) *backoff = h->persist_backoff;
    if (paused_for) *paused_for = h->persist_retry_at > h->batches_seen ? h->persist_retry_at - h->batches_seen : 0;
    return GOMEL_OK;
}
#ifdef GOMEL_STAMPS   // diagnostic build only: read (and clear) the per-phase cycle sums of k_gl_fused
int gomel_debug_stamps(unsigned long long* out21, int clear) {
    if (out21 && hipMemcpyFromSymbol(out21, HIP_SYMBOL(g_stamp_acc), 21 * sizeof(unsigned long long)) != hipSuccess) return GOMEL_ERR_HIP;
    if (clear) { unsigned long long z[21] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_acc), z, sizeof z) != hipSuccess) return GOMEL_ERR_HIP; }
    return GOMEL_OK;
}
#endif

// ---- host memory helpers: pinned buffers a bulk caller reuses across calls (no per-call hipHostRegister)
void* gomel_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
int gomel_host_free(void* p) {
    if (!p) return GOMEL_OK;
    HIPCHK(hipHostFree(p));
    return GOMEL_OK;
}
// ---- exception firewall: the entry points that allocate on the host
int gomel_set_window(gomel_handle* h, const double* w, size_t n) {
    GOMEL_TRY return set_window_guarded(h, w, n); GOMEL_CATCH
}
int gomel_batch_create(gomel_handle* h, size_t n_utt, const size_t* n_pairs, gomel_batch** out) {
    GOMEL_TRY return batch_create_guarded(h, n_utt, n_pairs, out); GOMEL_CATCH
}
int gomel_batch_create_f32(gomel_handle* h, size_t n_utt, const size_t* n_pairs, gomel_batch** out) {
    GOMEL_TRY return batch_create_f32_guarded(h, n_utt, n_pairs, out); GOMEL_CATCH
}
int gomel_batch_upload(gomel_batch* b, const double* const* mel, const double* const* x0, const uint64_t* seeds) {
    GOMEL_TRY return batch_upload_guarded(b, (const void* const*)mel, (const void* const*)x0, seeds, 0); GOMEL_CATCH
}
int gomel_batch_upload_f32(gomel_batch* b, const float* const* mel, const float* const* x0, const uint64_t* seeds) {
    GOMEL_TRY return batch_upload_guarded(b, (const void* const*)mel, (const void* const*)x0, seeds, 1); GOMEL_CATCH
}
int gomel_batch_run(gomel_batch* b) {
    GOMEL_TRY return batch_run_guarded(b); GOMEL_CATCH
}
int gomel_batch_run_timed(gomel_batch* b, gomel_timing* t) {
    GOMEL_TRY return batch_run_timed_guarded(b, t); GOMEL_CATCH
}
int gomel_batch_download(gomel_batch* b, double* const* out, const size_t* out_cap) {
    GOMEL_TRY return batch_download_guarded(b, (void* const*)out, out_cap, 0); GOMEL_CATCH
}
int gomel_batch_download_f32(gomel_batch* b, float* const* out, const size_t* out_cap) {
    GOMEL_TRY return batch_download_guarded(b, (void* const*)out, out_cap, 1); GOMEL_CATCH
}
int gomel_from_mel_batch(gomel_handle* h, size_t n_utt, const double* const* mel, const size_t* n_pairs, const double* const* x0,
                         const uint64_t* seeds, double* const* out, const size_t* out_cap) {
    GOMEL_TRY return from_mel_batch_guarded(h, n_utt, (const void* const*)mel, n_pairs, (const void* const*)x0, seeds, (void* const*)out, out_cap, 0); GOMEL_CATCH
}
int gomel_from_mel_batch_f32(gomel_handle* h, size_t n_utt, const float* const* mel, const size_t* n_pairs, const float* const* x0,
                             const uint64_t* seeds, float* const* out, const size_t* out_cap) {
    GOMEL_TRY return from_mel_batch_guarded(h, n_utt, (const void* const*)mel, n_pairs, (const void* const*)x0, seeds, (void* const*)out, out_cap, 1); GOMEL_CATCH
}
int gomel_from_mel_batch_submit(gomel_handle* h, size_t n_utt, const double* const* mel, const size_t* n_pairs, const double* const* x0,
                                const uint64_t* seeds, double* const* out, const size_t* out_cap, gomel_job** job) {
    GOMEL_TRY return submit_guarded(h, n_utt, (const void* const*)mel, n_pairs, (const void* const*)x0, seeds, (void* const*)out, out_cap, 0, job); GOMEL_CATCH
}
int gomel_from_mel_batch_submit_f32(gomel_handle* h, size_t n_utt, const float* const* mel, const size_t* n_pairs, const float* const* x0,
                                    const uint64_t* seeds, float* const* out, const size_t* out_cap, gomel_job** job) {
    GOMEL_TRY return submit_guarded(h, n_utt, (const void* const*)mel, n_pairs, (const void* const*)x0, seeds, (void* const*)out, out_cap, 1, job); GOMEL_CATCH
}
int gomel_job_wait(gomel_job* job) {
    GOMEL_TRY return job_wait_guarded(job); GOMEL_CATCH
}
int gomel_job_wait_timeout(gomel_job* job, int64_t timeout_ms) {
    GOMEL_TRY return job_wait_guarded(job, timeout_ms < 0 ? 0 : timeout_ms); GOMEL_CATCH
}
int gomel_job_poll(gomel_job* job, int* finished) {
    if (!job || !finished) return GOMEL_ERR_NULL;
    if (job->completed) { *finished = 1; return GOMEL_OK; }
    DeviceGuard dg(job->h->device); HIPCHK(dg.err);
    const hipError_t e = hipEventQuery(job->down);
    if (e != hipSuccess && e != hipErrorNotReady) { g_last_hip = (int)e; return GOMEL_ERR_HIP; }
    if (e == hipErrorNotReady) (void)hipGetLastError();
    *finished = e == hipSuccess;
    return GOMEL_OK;
}
int gomel_from_mel_batch_multi(gomel_handle* const* handles, size_t n_handles, size_t n_utt, const double* const* mel, const size_t* n_pairs,
                               const double* const* x0, const uint64_t* seeds, double* const* out, const size_t* out_cap) {
    GOMEL_TRY return from_mel_batch_multi_guarded(handles, n_handles, n_utt, (const void* const*)mel, n_pairs, (const void* const*)x0, seeds, (void* const*)out, out_cap, 0); GOMEL_CATCH
}
int gomel_target_magnitudes(gomel_handle* h, const double* mel, size_t n_pairs, float* mag_out) {
    GOMEL_TRY return target_magnitudes_guarded(h, mel, n_pairs, mag_out); GOMEL_CATCH
}
int gomel_from_phase(gomel_handle* h, int32_t num_freqs, double volume_boost, const double* pairs, size_t n_pairs, double* out, size_t out_cap,
                     size_t* out_len) {
    GOMEL_TRY return from_phase_guarded(h, num_freqs, volume_boost, pairs, n_pairs, out, out_cap, out_len); GOMEL_CATCH
}
int gomel_istft(gomel_handle* h, const double* mags, size_t n_frames, int32_t iterations, const double* x0, uint64_t seed, double* out,
                size_t out_cap, size_t* out_len) {
    GOMEL_TRY return istft_guarded(h, mags, n_frames, iterations, x0, seed, out, out_cap, out_len, nullptr); GOMEL_CATCH
}
int gomel_istft_spec(gomel_handle* h, const double* mags, size_t n_frames, int32_t iterations, const double* x0, uint64_t seed, double* out,
                     size_t out_cap, size_t* out_len, double* spec_out) {
    GOMEL_TRY return istft_guarded(h, mags, n_frames, iterations, x0, seed, out, out_cap, out_len, spec_out); GOMEL_CATCH
}
int gomel_to_mel(gomel_handle* h, const double* wave, size_t n, double* out_pairs, size_t cap_pairs, size_t* n_pairs) {
    GOMEL_TRY return to_mel_guarded(h, wave, n, out_pairs, cap_pairs, n_pairs); GOMEL_CATCH
}
int gomel_to_mel_batch(gomel_handle* h, size_t n_utt, const double* const* wave, const size_t* n_samples, double* const* out_pairs,
                       const size_t* cap_pairs, size_t* n_pairs_out) {
    GOMEL_TRY return to_mel_batch_guarded(h, n_utt, wave, n_samples, out_pairs, cap_pairs, n_pairs_out); GOMEL_CATCH
}
int gomel_to_phase(gomel_handle* h, int32_t num_freqs, const double* wave, size_t n, double* out_pairs, size_t cap_pairs, size_t* n_pairs) {
    GOMEL_TRY return to_phase_guarded(h, num_freqs, wave, n, out_pairs, cap_pairs, n_pairs); GOMEL_CATCH
}

}  // extern "C"
