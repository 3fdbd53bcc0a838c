// gomel_host.hpp -- the DEVICE-FREE host logic of libgomel_hip.so: table builder (hz_to_mel, undomel / domel rows, Hann), the automatic
// precision schedule, the geometry rules that pick a kernel path, segment and chunk planners, the device-buffer cache, the pinned-range
// registry, the device gate (templated on its event operations) and the NUMA lookup.  Nothing here includes a HIP header: this file and
// gomel_host.cpp compile as plain C++17 (g++ / clang++), which is how scripts/sanitize_host.sh builds them under ASan / UBSan / TSan.
// gomel_hip.hip (kernel launches, streams, the C ABI's device entry points) includes this header; gomel_host.cpp holds the C ABI's
// host-only entry points.  Reference lines cited are into neurlang/gomel (mel/impl.go: the mel <-> linear index math).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/gomel_hip.h"
#include "gl_types.hpp"

// No C++ exception crosses the C boundary: every entry point that can allocate on the host runs inside this pair.
#define GOMEL_TRY try {
#define GOMEL_CATCH                                              \
    } catch (const std::bad_alloc&) { return GOMEL_ERR_NO_MEMORY; } \
    catch (...) { return GOMEL_ERR_NO_MEMORY; }

struct TableHeader {
    uint32_t magic, version;
    int32_t n, hop, k, hann_periodic;
    int32_t dom_status, pad_;
    double fmin, fmax;
};
static const uint32_t kMagic = 0x474d454cu;  // "GMEL"
// The blob's OWN format number, stamped into TableHeader::version: it changes only when the blob's layout (TableHeader, window, BinRow
// rows) or the table builder's semantics change -- not with the C ABI (ABI 4 -> 5 -> 6 added entry points; blobs written by those
// builds stay interchangeable, which ranks running mixed library builds behind gomel_create_from_tables rely on).  4 = the value the
// ABI-4 and ABI-5 builds stamped.
static const uint32_t kTableFormat = 4;

// Device-buffer cache of one handle.  hipFree of a multi-megabyte buffer is synchronous and slow (0.6 ms of the 0.75 ms that
// one 342-frame, 2-iteration FromMel call took host-to-host went into freeing its working set), and the Go API is one call
// per utterance: freed buffers are kept and handed out again to requests of similar size.  Guarded by the handle's mutex.
struct DevPool {
    struct Block { void* p; size_t bytes; bool used; uint64_t stamp; };
    // the allocator behind the cache: the HIP runtime in the product; gomel_debug_pool_selftest (CPU, sanitizer builds) injects host malloc
    // (0 = success; the HIP TU installs hipMalloc / hipFree and a "forget the runtime's last error" hook in every handle it creates)
    int (*alloc_fn)(void**, size_t) = nullptr;
    int (*free_fn)(void*) = nullptr;
    void (*forget_error_fn)() = nullptr;
    std::vector<Block> blocks;
    size_t cached = 0, cap = (size_t)16 << 30;  // bytes kept while unused (GOMEL_POOL_MB; 0 = no caching); the device has 288 GB
    uint64_t clock = 0;
    int alloc(void** out, size_t bytes) {
        bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
        int best = -1;
        for (size_t i = 0; i < blocks.size(); i++) {
            const Block& k = blocks[i];
            if (!k.used && k.bytes >= bytes && k.bytes <= bytes + bytes / 2 + 65536 && (best < 0 || k.bytes < blocks[best].bytes)) best = (int)i;
        }
        if (best >= 0) { blocks[best].used = true; cached -= blocks[best].bytes; *out = blocks[best].p; return 0; }
        int e = alloc_fn(out, bytes);
        if (e != 0) { if (forget_error_fn) forget_error_fn(); trim(0); e = alloc_fn(out, bytes); }   // give the cache back and retry
        if (e == 0) blocks.push_back(Block{*out, bytes, true, 0});
        return e;
    }
    void release(void* p) {
        for (Block& k : blocks)
            if (k.p == p) { k.used = false; k.stamp = ++clock; cached += k.bytes; trim(cap); return; }
        (void)free_fn(p);   // not ours (never happens): plain free
    }
    void trim(size_t keep) {   // drop the least recently released buffers until at most `keep` bytes stay cached
        while (cached > keep) {
            int old = -1;
            for (size_t i = 0; i < blocks.size(); i++)
                if (!blocks[i].used && (old < 0 || blocks[i].stamp < blocks[old].stamp)) old = (int)i;
            if (old < 0) break;
            (void)free_fn(blocks[old].p);
            cached -= blocks[old].bytes;
            blocks.erase(blocks.begin() + old);
        }
    }
    void clear() {
        for (Block& k : blocks) (void)free_fn(k.p);
        blocks.clear(); cached = 0;
    }
};

// ----------------------------------------------------------------------------------------------- tables, schedule
static double hz_to_mel(double v) { return 1127.0 * std::log(1.0 + (v / 700.0)); }  // mel/impl.go:304-308
static double mel_to_hz(double v) { return 700.0 * (std::exp(v / 1127.0) - 1.0); }  // mel/impl.go:298-302

static int classify_row(double vallo, double valhi, bool undomel_rules, int limit, BinRow* r) {
    double inlo, modlo = std::modf(vallo, &inlo);
    double inhi = std::floor(valhi);
    if (inlo < 0) { inlo = 0; modlo = 0; inhi = 0; }  // note: -0.0 < 0 is false, exactly as in Go
    long lo = (long)inlo, hi = (long)inhi;
    r->lo = (int32_t)lo; r->hi = (int32_t)hi; r->modlo = modlo; r->pad_ = 0; r->div = 1;
    if (undomel_rules) {  // mel/impl.go:365-375
        if (lo == hi) { r->mode = 0; if (lo < 0 || lo >= limit) return GOMEL_ERR_TABLE_RANGE; }
        else if (lo + 1 == hi && hi < limit) { r->mode = 1; if (lo < 0) return GOMEL_ERR_TABLE_RANGE; }
        else { r->mode = 2; r->div = inhi - inlo + 1; if (hi > lo && (lo < 0 || hi > limit)) return GOMEL_ERR_TABLE_RANGE; }
    } else {  // domel, mel/impl.go:328-336
        if (lo + 1 == hi) { r->mode = 1; if (lo < 0 || hi >= limit) return GOMEL_ERR_TABLE_RANGE; }
        else { r->mode = 2; r->div = (double)(hi - lo + 1); if (hi > lo && (lo < 0 || hi > limit)) return GOMEL_ERR_TABLE_RANGE; }
    }
    return GOMEL_OK;
}

// Bounds a BinRow must meet before it is used as a device-side index: undomel rows read bands [lo, hi) or {lo, hi} of K
// bands; domel rows read bins [lo, hi) or {lo, hi} of the N/2+1 magnitudes (channel 1 reads one bin higher).
static bool rows_in_range(const std::vector<BinRow>& rows, int limit, bool undomel_rules) {
    for (const BinRow& r : rows) {
        if (r.mode < 0 || r.mode > 2 || r.lo < 0 || !std::isfinite(r.modlo) || !std::isfinite(r.div) || r.div == 0.0) return false;
        if (r.mode == 0 && r.lo >= limit) return false;
        if (r.mode == 1 && (r.hi < 0 || r.hi >= limit || r.lo >= limit)) return false;
        if (r.mode == 2 && r.hi > r.lo && r.hi > limit) return false;
        if (!undomel_rules && r.mode == 0) return false;   // domel has no copy rows
    }
    return true;
}

static int build_mel_tables(const gomel_config& c, std::vector<BinRow>& undo, std::vector<BinRow>& dom, int* dom_status) {
    const int H = c.resolut / 2, K = c.num_mels;
    const double fmin = c.mel_fmin, fmax = c.mel_fmax;
    undo.resize(H);
    const double filterbin = hz_to_mel(fmax) / (double)K;  // mel/impl.go:348
    for (int i = 0; i < H; i++) {
        double vallo = hz_to_mel(((double)i * (fmax + fmin) / (double)H) - fmin) / filterbin;
        double valhi = hz_to_mel(((double)(i + 1) * (fmax + fmin) / (double)H) - fmin) / filterbin;
        int rc = classify_row(vallo, valhi, true, K, &undo[i]);
        if (rc) return rc;
    }
    dom.resize(K);
    *dom_status = GOMEL_OK;
    const double melbin = hz_to_mel(fmax) / (double)K;  // mel/impl.go:311
    for (int i = 0; i < K; i++) {
        double vallo = (double)H * (fmin + mel_to_hz(melbin * (double)i)) / (fmax + fmin);
        double valhi = (double)H * (fmin + mel_to_hz(melbin * (double)(i + 1))) / (fmax + fmin);
        int rc = classify_row(vallo, valhi, false, H, &dom[i]);
        if (rc) { *dom_status = rc; dom[i].mode = 2; dom[i].lo = dom[i].hi = 0; dom[i].div = 1; }
    }
    return GOMEL_OK;
}

// Everything a handle derives from the configuration alone (no device involved): Hann, undomel rows, domel rows.
static int build_host_tables(const gomel_config& c, std::vector<double>& window, std::vector<BinRow>& undo, std::vector<BinRow>& dom,
                             int* dom_status) {
    const int N = c.resolut;
    window.resize(N);
    const double arg = 2.0 * M_PI / (double)(c.hann_periodic ? N : N - 1);  // gossp window.CreateHanning
    for (int i = 0; i < N; i++) window[i] = 0.5 - 0.5 * std::cos(arg * (double)i);
    return build_mel_tables(c, undo, dom, dom_status);
}

static int validate_config(const gomel_config* c) {
    if (!c) return GOMEL_ERR_NULL;
    if (c->num_mels <= 0 || c->num_mels > kMaxMels || c->window <= 0 || c->resolut <= 0 || c->griffin_lim_iterations < 0) return GOMEL_ERR_BAD_CONFIG;
    if (c->tune_mul == 0.0 || !(c->mel_fmax + c->mel_fmin > 0.0) || !(c->mel_fmax > 0.0)) return GOMEL_ERR_BAD_CONFIG;
    if (!gomel_supported_n(c->resolut) && !gomel_generic_n(c->resolut)) return GOMEL_ERR_UNSUPPORTED_N;
    return GOMEL_OK;
}

// The automatic precision schedule: how many of a call's I Griffin-Lim iterations run in fp64 before the fp32 kernels take over
// (mel/mel.go:76-139 is float64 throughout; the schedule keeps the default call within the 1e-4 RMS bar of SURVEY 8(b) at a calibrated
// RATE -- fp64 throughout, gomel_set_precision(h, 64), is the guarantee).  Two calibrated tables (scripts/calibrate_schedule.py; how they
// were made, and every held-out tally, is DESIGN.md section 2 / LABBOOK.md): kAutoSchedule for frames that overlap up to four-fold
// (towav: N 4096 / hop 1280; TTS: 1024 / 256), kAutoScheduleDense from six-fold on (NewMel's defaults, mel/mel.go:30-41: 2048 / 256, eight-fold).
// The rule is MONOTONE in the overlap Resolut / Window: between four- and six-fold k is interpolated linearly between the two tables'
// values, rounded up (N 2048 / hop 400, 5.12-fold: 44 of 60).  Between calibrated iteration counts k is interpolated upwards; beyond the
// last one the number of trailing fp32 iterations is held; up to five iterations run in fp64 throughout (the reference's shipped
// GriffinLimIterations = 2, mel/mel.go:39, among them).
// FROZEN in round 5: no table edit without a new failing full-size test (tests/test_gpu_round5.py holds the default call to < 1e-4 on
// 256 x 400 frames at the NewMel-default geometry and on the interpolated stretch).
struct AutoPoint { int iters, k; };
static const AutoPoint kAutoSchedule[] = {{2, 2}, {5, 5}, {10, 6}, {20, 14}, {30, 25}, {40, 30}, {60, 35}, {100, 80}};        // overlap <= 4
static const AutoPoint kAutoScheduleDense[] = {{2, 2}, {5, 5}, {10, 6}, {20, 16}, {30, 27}, {40, 32}, {60, 50}, {100, 90}};   // overlap >= 6
static int auto_table_k(const AutoPoint* tab, int I) {
    constexpr int n = (int)(sizeof kAutoSchedule / sizeof kAutoSchedule[0]);
    static_assert(sizeof kAutoSchedule == sizeof kAutoScheduleDense, "the two tables have the same calibrated iteration counts");
    if (I <= tab[0].iters) return std::min(I, tab[0].k);
    for (int i = 1; i < n; i++)
        if (I <= tab[i].iters) {
            const AutoPoint a = tab[i - 1], b = tab[i];
            const int k = a.k + ((b.k - a.k) * (I - a.iters) + (b.iters - a.iters) - 1) / (b.iters - a.iters);   // rounded up
            return std::min(I, std::max(k, a.k));
        }
    return I - (tab[n - 1].iters - tab[n - 1].k);
}
static int auto_fp64_iterations(const gomel_config& c, int I) {
    const int k_lo = auto_table_k(kAutoSchedule, I), k_hi = std::max(k_lo, auto_table_k(kAutoScheduleDense, I));
    const int64_t N = c.resolut, hop = c.window;
    if (hop <= 0 || N <= 4 * hop) return k_lo;                  // overlap <= 4 (hops >= Resolut included)
    if (N >= 6 * hop) return k_hi;                              // overlap >= 6
    const int64_t num = (int64_t)(k_hi - k_lo) * (N - 4 * hop), den = 2 * hop;   // (k_hi - k_lo) * (overlap - 4) / 2
    return std::min(I, k_lo + (int)((num + den - 1) / den));
}
static void pack_tables(const gomel_config& c, int dom_status, const std::vector<double>& window, const std::vector<BinRow>& undo,
                        const std::vector<BinRow>& dom, void* blob) {
    TableHeader th{kMagic, kTableFormat, c.resolut, c.window, c.num_mels, c.hann_periodic, dom_status, 0, c.mel_fmin, c.mel_fmax};
    char* p = (char*)blob;
    memcpy(p, &th, sizeof th); p += sizeof th;
    memcpy(p, window.data(), window.size() * 8); p += window.size() * 8;
    memcpy(p, undo.data(), undo.size() * sizeof(BinRow)); p += undo.size() * sizeof(BinRow);
    memcpy(p, dom.data(), dom.size() * sizeof(BinRow));
}

static inline size_t tables_bytes_for(const gomel_config& c) {
    return sizeof(TableHeader) + (size_t)c.resolut * 8 + ((size_t)c.resolut / 2 + c.num_mels) * sizeof(BinRow);
}
// magnitudes per frame the exported ISTFT reads: bins 0 ... N/2, and bin N/2+1 as well when N is odd (both top bins lie outside the conjugate fill)
static inline size_t istft_row(int N) { return (size_t)(N / 2 + 1 + (N & 1)); }

// ----------------------------------------------------------------------------------------------- kernel-path rules, planners
// The fused kernel needs hop = SH * (N/16) with SH in the instantiated set.
static int fused_shift(const gomel_config& c) {
    if (!gomel_supported_n(c.resolut)) return 0;          // generic-length path: gl_any.hpp kernels only
    const int row = c.resolut / 16;
    if (c.window % row) return 0;
    const int sh = c.window / row;
    return (sh == 1 || sh == 2 || sh == 4 || sh == 5 || sh == 8) ? sh : 0;
}
// Every other even hop below N: k_gl_fused_any (the same sliding window, shifted through LDS).  gmin = frames a stream must walk so
// that what it adds past its own samples reaches into the next segment only.
static int fused_gmin(const gomel_config& c) { return c.window > 0 ? (c.resolut - 1) / c.window : 0; }   // ceil((N - hop) / hop)
static bool fused_any_ok(const gomel_config& c) {
    return gomel_supported_n(c.resolut) && c.window > 0 && c.window < c.resolut && fused_gmin(c) <= 31;
}
static int fused_kind(const gomel_config& c) {   // 1 k_gl_fused, 2 k_gl_fused_any
    static const bool prefer_any = getenv("GOMEL_FORCE_ANY") != nullptr;   // A/B knob: the any-hop kernel at an aligned hop
    if (prefer_any && fused_any_ok(c)) return 2;
    return fused_shift(c) ? 1 : fused_any_ok(c) ? 2 : 0;
}
static bool fused_applicable(const gomel_config& c) { return fused_kind(c) != 0; }

// The persistent kernel (k_gl_persist) wants 16-byte quads that never straddle a hop (hop and N multiples of 4), at most
// kPersistMaxNb neighbours on each side, and one of the register FFT cores.
static bool persist_geometry_ok(const gomel_config& c) {
    return gomel_supported_n(c.resolut) && c.window % 4 == 0 && c.window > 0 && (c.resolut - 1) / c.window <= kPersistMaxNb;
}
// Batch size (total frames) from which the fused kernel beats the generic pair.  A fused stream walks at least gmin =
// ceil((16-SH)/SH) frames one after the other, so a launch never takes less than gmin frame latencies, while the generic kernels
// run every frame in parallel: measured crossovers for gmin = 3 (scripts/cross_probe2.py, 60 iterations, one launch per
// iteration): 1.5 k frames at N 4096, 2.5 k at 2048, 3 k at 1024, 6 k at 512, 12 k at 256; scaled with gmin.
// With fp64 iterations in the call (default mode, 64-bit mode) the crossovers lie elsewhere (scripts/size_sweep.py GEOM 400 1|2 ..., profiles/r04/
// size_sweep_paths.txt; default mode, 60 iterations): N 4096 ~2 000 frames, N 2048 / hop 256 ~4 200 (gmin 7: the fp64 fused launch is flat at
// 3.1-3.3 ms up to a full round of 6 400 frames, the generic pair passes that at 4 000), N 1024 ~4 500, N 512 ~7 500, N 256 ~15 000.
static int64_t fused_min_frames(const gomel_config& c, bool fp64_iterations = false) {
    if (!fused_kind(c)) return INT64_MAX;
    const int64_t base = fp64_iterations ? (c.resolut >= 4096 ? 2048 : c.resolut == 2048 ? 1800 : c.resolut == 1024 ? 4500 : c.resolut == 512 ? 7500 : 15000)
                                         : (c.resolut >= 4096 ? 1536 : c.resolut == 2048 ? 2560 : c.resolut == 1024 ? 3072 : c.resolut == 512 ? 6144 : 12288);
    return base * std::max(1, fused_gmin(c)) / 3;
}

// Segment plan of the fused kernel.  What matters (scripts/chunk_probe.py): segments of 25-40 frames amortise the 16-row
// prologue and the workgroup's table fill, and the NUMBER of segments should fill whole rounds of the chip: a workgroup takes
// STREAMS segments and 2 workgroups fit a CU, so one round is 512 workgroups; 533 workgroups cost two rounds (1.75 M frames/s
// where 512 reach 1.95 M on the same 12 800 frames).  So: pick the number of rounds r for ~50 frames per segment, aim at
// r * 512 * STREAMS segments in total, and give each utterance its share in proportion to its frames.
static void plan_segments(const gomel_config& c, const std::vector<UttDesc>& utts, int64_t total_frames, std::vector<SegDesc>& segs, int& g_max) {
    const int gmin = std::max(1, fused_gmin(c));
    const int streams = kFusedThreads / (c.resolut / 32);                      // segments per workgroup (T = N/32 lanes per stream)
    const int64_t per_round = (int64_t)512 * streams;
    // ~50 frames per segment where the batch gives two rounds or more (fewer 16-row prologues and table fills: 2 rounds of 50 beat
    // 4 of 25 by 1 % at config 3, scripts/seg_probe.sh), never above 60 (the kernel's limit is kMaxSegFrames = 64)
    int64_t rounds = std::max<int64_t>(1, (total_frames + 25 * per_round) / (50 * per_round));
    while (total_frames > rounds * per_round * 60) rounds++;
    double target = (double)(rounds * per_round);                              // segments wanted in total
    int forced = 0;
    if (const char* e = getenv("GOMEL_SEG_FRAMES")) { int v = atoi(e); if (v > 0) forced = std::min(std::max(v, gmin), kMaxSegFrames); }  // tuning knob
    segs.clear();
    g_max = 0;
    int64_t frames_done = 0;
    double segs_done = 0;
    for (const UttDesc& u : utts) {
        int nseg;
        if (forced) nseg = std::max(1, (u.frames + forced / 2) / forced);
        else {   // this utterance's share of the target, carried over utterances so that the total lands on it
            const double want = target * (double)(frames_done + u.frames) / (double)total_frames - segs_done;
            nseg = std::max(1, (int)std::llround(want));
        }
        while (nseg > 1 && u.frames / nseg < gmin) nseg--;
        // hard cap kMaxSegFrames frames per segment (k_gl_fused keeps one Nyquist magnitude per lane); it wins over gmin, which
        // only matters for has_next segments of very short utterances and gmin <= 15 < 64 / 2 always leaves room
        nseg = std::max(nseg, (u.frames + kMaxSegFrames - 1) / kMaxSegFrames);
        frames_done += u.frames; segs_done += nseg;
        const int base = u.frames / nseg, rem = u.frames % nseg;
        int f0 = 0;
        for (int s = 0; s < nseg; s++) {
            SegDesc d;
            d.G = base + (s < rem ? 1 : 0);
            d.s0 = u.xoff + (int64_t)f0 * c.window;
            d.frame0 = u.fbase + f0;
            d.has_prev = s > 0; d.has_next = s + 1 < nseg; d.pad_ = 0;
            segs.push_back(d);
            g_max = std::max(g_max, d.G);
            f0 += d.G;
        }
    }
}

static std::vector<size_t> plan_chunks(const gomel_config& cfg, size_t n_utt, const size_t* n_pairs, bool heavy) {
    const int K = cfg.num_mels;
    int64_t total_frames = 0;
    for (size_t u = 0; u < n_utt; u++) total_frames += (int64_t)(n_pairs[u] / (size_t)K);
    // Chunk plan (scripts/chunk_plans.sh, 256 x 400 frames host to host): m full-size chunks' worth of frames, cut as
    // half : full x (m-1) : half -- the half-size first chunk starts the GPU sooner, the half-size last one leaves less download
    // exposed at the end (T geometry: 44.3 ms against 45.6 for eight equal chunks, with m = 8).  A full chunk is ~12k frames at
    // Resolut 4096 (half a round of the chip, 85 us per launch) and ~32k below (shorter frames: the same launch time; at the TTS
    // geometry three full chunks' worth, 13.5 ms, against 16.1 for eight equal chunks).  More, smaller chunks pay launch overhead and
    // per-segment prologues; fewer expose more of the first upload and the last download.
    const int64_t kMinChunkFrames = (cfg.resolut >= 4096 ? 12288 : 32768) * (heavy ? 2 : 1);
    const int m_full = (int)std::min<int64_t>(16, std::max<int64_t>(1, total_frames / kMinChunkFrames));
    int n_chunks = total_frames < kMinChunkFrames ? 1 : m_full + 1;
    std::vector<double> shares;   // relative chunk sizes
    if (n_chunks > 1) { shares.assign((size_t)n_chunks, 2.0); shares.front() = shares.back() = 1.0; }
    if (const char* e = getenv("GOMEL_CHUNKS")) { int v = atoi(e); if (v > 0) { n_chunks = std::min(v, 64); shares.assign((size_t)n_chunks, 1.0); } }   // tuning knobs: equal chunks ...
    if (const char* e = getenv("GOMEL_CHUNK_SHARES")) {                                                                                          // ... or any plan, "1,2,4,2,1"
        std::vector<double> sv;
        for (const char* p = e; *p;) { char* q; const double v = strtod(p, &q); if (q == p) break; if (v > 0) sv.push_back(v); p = *q ? q + 1 : q; }
        if (sv.size() >= 1 && sv.size() <= 64) { shares = sv; n_chunks = (int)sv.size(); }
    }
    n_chunks = (int)std::min<size_t>((size_t)n_chunks, n_utt);
    if ((int)shares.size() != n_chunks) shares.assign((size_t)n_chunks, 1.0);
    double share_sum = 0;
    for (double v : shares) share_sum += v;
    std::vector<size_t> bounds(1, 0);
    {
        // A fused launch lasts as long as its longest segment, ceil(frames / segments of its rounds): a chunk just BELOW a multiple
        // of a round's segment count uses every stream to the end, one just above it leaves them idle for most of a frame (33
        // utterances of 400 frames per launch: 6.26 ns per frame; 32: 6.40; 17: 7.44; 16: 7.86 -- scripts/launch_scaling.py).  Each
        // boundary is therefore moved by up to two utterances to where the chunk it closes fills its rounds best.
        std::vector<int64_t> pre(n_utt + 1, 0);
        for (size_t u = 0; u < n_utt; u++) pre[u + 1] = pre[u] + (int64_t)(n_pairs[u] / (size_t)K);
        const int64_t round0 = fused_applicable(cfg) ? (int64_t)512 * (kFusedThreads / (cfg.resolut / 32)) : 0;
        auto fill = [&](int64_t F) {   // fraction of the launch's stream-time that does work (plan_segments' rounds rule)
            if (round0 == 0 || F <= 0) return 1.0;
            int64_t rnd = std::max<int64_t>(1, (F + 25 * round0) / (50 * round0));
            while (F > rnd * round0 * 60) rnd++;
            const int64_t slots = rnd * round0;
            return (double)F / (double)(slots * ((F + slots - 1) / slots));
        };
        size_t u = 0;
        double cum = 0;
        for (int c = 1; c < n_chunks; c++) {
            cum += shares[(size_t)c - 1];
            const int64_t target = (int64_t)((double)total_frames * cum / share_sum);
            while (u < n_utt && pre[u] < target) u++;
            size_t best = std::max(u, bounds.back() + 1);
            if (!getenv("GOMEL_CHUNK_SHARES") && !getenv("GOMEL_CHUNKS")) {
                const int64_t want = pre[best] - pre[bounds.back()];
                double best_fill = fill(want);
                for (int dlt = -2; dlt <= 2; dlt++) {
                    const int64_t cand = (int64_t)best + dlt;
                    if (dlt == 0 || cand <= (int64_t)bounds.back() || cand >= (int64_t)n_utt) continue;
                    const int64_t F = pre[(size_t)cand] - pre[bounds.back()];
                    if (F * 100 < want * 85 || F * 100 > want * 120) continue;
                    if (fill(F) > best_fill + 0.004) { best_fill = fill(F); u = (size_t)cand; }
                }
                if (best_fill > fill(want)) best = u;
            }
            u = best;
            bounds.push_back(best);
        }
        bounds.push_back(n_utt);
        for (size_t i = 1; i < bounds.size(); i++) if (bounds[i] <= bounds[i - 1] || bounds[i] > n_utt) { bounds.assign({0, n_utt}); break; }
        n_chunks = (int)bounds.size() - 1;
    }
    return bounds;
}

// Host buffers in, host buffers out.  Big batches are cut into chunks of contiguous utterances and pipelined over three
// streams: chunk c+1's log-mel goes up (up_stream) while chunk c computes (stream) and chunk c-1's waveforms come down
// (copy_stream); PCIe is full duplex, so with pinned host buffers the wall time is the compute time plus the first upload and
// the last download.  Caller buffers that are not already pinned (gomel_host_alloc / gomel_host_register) are pinned for the call.
// ----------------------------------------------------------------------------------------------- pinned host ranges
// What the registry asks of the runtime.  The product's backend is HIP; gomel_debug_pin_backend(1) swaps in an in-process fake that
// records registrations and counts contract violations (overlapping registrations, unregistering what is not registered), so that the
// interval logic and its locking can be exercised -- under AddressSanitizer / ThreadSanitizer -- on a machine without a GPU.
struct PinBackend {
    bool (*foreign)(void* p, uintptr_t* base, size_t* size);   // pinned / device memory somebody else owns?  (its allocation's extent)
    bool (*reg)(void* p, size_t bytes);
    void (*unreg)(void* p);
};
static bool no_pin_foreign(void*, uintptr_t* base, size_t* size) { *base = 0; *size = 0; return false; }
static bool no_pin_reg(void*, size_t) { return false; }
static void no_pin_unreg(void*) {}
static const PinBackend kNoPins = {no_pin_foreign, no_pin_reg, no_pin_unreg};   // a build without a runtime behind it: every buffer stays pageable
struct PinRegistry {
    struct Entry { size_t bytes; int users; };
    const PinBackend* be = &kNoPins;          // the HIP TU installs its backend at load time (product_be); gomel_debug_pin_backend(1): the fake one
    const PinBackend* product_be = &kNoPins;
    std::mutex mu;
    std::map<uintptr_t, Entry> live;   // base -> extent, users
    // Ranges found pinned (or device memory) by somebody else -- gomel_host_alloc / hipHostMalloc buffers a bulk caller reuses.
    // Asking the runtime costs ~10-20 us per pointer (hipPointerGetAttributes), 5-10 ms for the 768 buffers of a 256-utterance
    // call; the answer is remembered per allocation.  A stale entry (buffer freed, address reused by pageable memory) only
    // means that buffer is not pinned for the call: slower copy, same result.
    std::map<uintptr_t, size_t> foreign;   // allocation base -> size
    // returns the base of the registration that now covers [p, p+bytes) (a token for release), or 0
    uintptr_t acquire(void* p, size_t bytes) {
        const uintptr_t a = (uintptr_t)p;
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.upper_bound(a);
        if (it != live.begin()) {
            auto pr = std::prev(it);
            if (a + bytes <= pr->first + pr->second.bytes) { pr->second.users++; return pr->first; }   // inside a live registration
            if (a < pr->first + pr->second.bytes) return 0;                                          // straddles its end
        }
        if (it != live.end() && it->first < a + bytes) return 0;                                      // runs into the next one
        {
            auto f = foreign.upper_bound(a);
            if (f != foreign.begin()) { auto pf = std::prev(f); if (a + bytes <= pf->first + pf->second) return 0; }
        }
        uintptr_t fbase = 0; size_t fsize = 0;
        if (be->foreign(p, &fbase, &fsize)) {
            if (fbase && fsize) {
                if (foreign.size() > 65536) foreign.clear();
                foreign[fbase] = fsize;
            }
            return 0;
        }
        if (!be->reg(p, bytes)) return 0;
        live.emplace(a, Entry{bytes, 1});
        return a;
    }
    void release(uintptr_t base) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(base);
        if (it == live.end()) return;
        if (--it->second.users == 0) {
            be->unreg((void*)base);
            live.erase(it);
        }
    }
};
inline PinRegistry& pin_registry() { static PinRegistry r; return r; }   // one per process (C++17 inline: shared by both translation units)
#define g_pins (pin_registry())

// The buffers of ONE call, without the runtime: want() only records (base, longest extent asked for) -- prefix views buf[:a], buf[:b] of
// one base become one registration of max(a, b) bytes -- and commit() registers.  (gomel_hip.hip's PinnedScope adds the streams a
// call's copies run on and drains them before release_all.)
struct PinWants {
    std::vector<std::pair<void*, size_t>> wanted;
    std::vector<uintptr_t> held;
    void want(const void* p, size_t bytes) {
        if (!p || bytes < (1u << 20)) return;
        void* q = const_cast<void*>(p);
        for (auto& w : wanted) if (w.first == q) { w.second = std::max(w.second, bytes); return; }
        wanted.emplace_back(q, bytes);
    }
    void commit() {
        for (auto& w : wanted) if (uintptr_t t = g_pins.acquire(w.first, w.second)) held.push_back(t);
        wanted.clear();
    }
    void release_all() {
        for (uintptr_t t : held) g_pins.release(t);
        held.clear();
    }
};

// ----------------------------------------------------------------------------------------------- device gate
// EvOps: { typedef Event; static int query(Event) -> 0 done, 1 not ready, < 0 error (forgotten) } -- HIP events in the product, a fake in tests/host_stress.cpp
// The whole admission protocol lives here, so that the sanitizer builds (no HIP) exercise the code the product runs:
//   ordinary kernel work:  enter_ordinary(wait_for) ... enqueue ... leave_ordinary(&ev, owner)     (or leave_ordinary(nullptr, ..) after a sync)
//   a persistent launch:   if (try_take_persistent(units, owner)) { ... enqueue ... record_persistent(&ev, owner); }   -- mu is HELD in between
//                          release_persistent(units, owner) once the host has seen the launch finish
// Invariant: a persistent launch is admitted only while no ordinary run of ANOTHER owner is enqueueing or in flight, and the shares of
// the persistent launches in flight never exceed the device (1024 units).
template <class EvOps> struct DeviceGateT {
    typedef typename EvOps::Event Event;
    struct Ent { Event ev; const void* owner; };
    std::mutex mu;
    int persist_units = 0;      // 1/1024ths of the device taken by queued persistent launches
    int enqueueing = 0;         // ordinary runs that have entered the gate and not yet recorded their event
    std::vector<Ent> np;        // events behind ordinary kernel work in flight
    std::vector<Ent> ps;        // events behind persistent launches in flight
    static void prune(std::vector<Ent>& v) {   // caller holds mu; drops what has completed
        size_t k = 0;
        for (size_t i = 0; i < v.size(); i++) {
            if (EvOps::query(v[i].ev) == 1) v[k++] = v[i];
        }
        v.resize(k);
    }
    static void put(std::vector<Ent>& v, Event ev, const void* owner) {
        for (const Ent& e : v) if (e.ev == ev) return;   // re-recorded: the entry now stands for the latest record
        v.push_back(Ent{ev, owner});
    }
    // Ordinary work is about to be enqueued: counts it, and hands every persistent launch still in flight to `wait_for` (the product
    // makes its stream wait for that event, device-side).
    template <class F> void enter_ordinary(F&& wait_for) {
        std::lock_guard<std::mutex> lk(mu);
        enqueueing++;
        prune(ps);
        for (const Ent& e : ps) wait_for(e.ev);
    }
    // ... has been enqueued.  ev: the event just recorded behind it (nullptr: it has been synchronised with, nothing stays in flight).
    void leave_ordinary(const Event* ev, const void* owner) {
        std::lock_guard<std::mutex> lk(mu);
        enqueueing--;
        if (ev) put(np, *ev, owner);
    }
    // A persistent launch of `units` (0: an earlier run of the same owner is still accounted -- same stream, same share).  true: admitted,
    // and mu stays LOCKED until record_persistent, so that "nothing else in flight" still holds when the kernel reaches the queue.
    bool try_take_persistent(int units, const void* owner) {
        mu.lock();
        prune(np);
        bool others = enqueueing != 0;
        for (const Ent& e : np) others |= e.owner != owner;   // (the owner's own earlier runs sit on the same stream: it runs behind them)
        if (others || persist_units + units > 1024) { mu.unlock(); return false; }
        persist_units += units;
        return true;
    }
    // ev recorded behind the launch (nullptr: nothing was launched after all).  Unlocks mu.
    void record_persistent(const Event* ev, const void* owner) {
        if (ev) put(ps, *ev, owner);
        mu.unlock();
    }
    // the owner's launches are known to have finished (or never ran)
    void release_persistent(int units, const void* owner) {
        std::lock_guard<std::mutex> lk(mu);
        persist_units -= units;
        for (size_t i = 0; i < ps.size();) { if (ps[i].owner == owner) ps.erase(ps.begin() + i); else i++; }
    }
    // the owner (and its events) is going away
    void forget(const void* owner) {
        std::lock_guard<std::mutex> lk(mu);
        for (std::vector<Ent>* v : {&np, &ps})
            for (size_t i = 0; i < v->size();) { if ((*v)[i].owner == owner) v->erase(v->begin() + i); else i++; }
    }
};

// ----------------------------------------------------------------------------------------------- bounded host waits
// gomel_job_wait_timeout: poll an event until it is done or `timeout_ms` have passed -- a server goroutine must not block forever
// behind a device that stopped answering.  Polls with short sleeps that grow to 1 ms (a job is tens of milliseconds long).
// Returns 0 done, 1 still running at the deadline, < 0 the query failed.  Clock and sleep are injectable (tests/host_stress.cpp).
struct SteadyWaitClock {
    static int64_t now_us() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static void sleep_us(int64_t us) { std::this_thread::sleep_for(std::chrono::microseconds(us)); }
};
template <class EvOps, class Clock = SteadyWaitClock> static int wait_event_bounded(typename EvOps::Event ev, int64_t timeout_ms) {
    const int64_t deadline = Clock::now_us() + std::min<int64_t>(timeout_ms < 0 ? 0 : timeout_ms, (int64_t)1 << 40) * 1000;   // (clamped: ~35 years, no overflow)
    int64_t nap = 50;
    for (;;) {
        const int q = EvOps::query(ev);
        if (q != 1) return q;
        const int64_t left = deadline - Clock::now_us();
        if (left <= 0) return 1;
        Clock::sleep_us(std::min(nap, left));
        nap = std::min<int64_t>(nap * 2, 1000);
    }
}

// ----------------------------------------------------------------------------------------------- NUMA locality of host threads
// The CPUs of the NUMA node a PCI device hangs off, from sysfs (root injectable: the CPU tests run on a fake tree).  Used by the
// one-process multi-GPU path (gomel_from_mel_batch_multi: one host thread per device) the way gomel_amd.dist.bind_to_gpu_numa_node
// binds the one-process-per-GPU ranks.
static bool read_small_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    out.assign(buf, n);
    return true;
}
static bool parse_cpulist(const std::string& s, std::vector<int>& cpus) {   // "0-3,16-19\n" -> 0 1 2 3 16 17 18 19
    cpus.clear();
    size_t i = 0;
    auto num = [&](long& v) {
        if (i >= s.size() || s[i] < '0' || s[i] > '9') return false;
        v = 0;
        while (i < s.size() && s[i] >= '0' && s[i] <= '9') { v = v * 10 + (s[i] - '0'); if (v > (1 << 20)) return false; i++; }
        return true;
    };
    while (i < s.size() && s[i] != '\n' && s[i] != '\0') {
        long a, b;
        if (!num(a)) return false;
        b = a;
        if (i < s.size() && s[i] == '-') { i++; if (!num(b) || b < a) return false; }
        for (long c = a; c <= b; c++) cpus.push_back((int)c);
        if (i < s.size() && s[i] == ',') i++;
    }
    return !cpus.empty();
}
static int numa_cpus_of(const char* sysfs_root, const char* bdf, int* node_out, std::vector<int>& cpus) {
    if (!bdf || !*bdf) return GOMEL_ERR_NULL;
    std::string root = sysfs_root && *sysfs_root ? sysfs_root : "/sys", id = bdf, txt;
    for (char& c : id) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');   // sysfs spells bus ids in lower case
    if (!read_small_file(root + "/bus/pci/devices/" + id + "/numa_node", txt)) return GOMEL_ERR_EMPTY;
    const long node = strtol(txt.c_str(), nullptr, 10);
    if (node_out) *node_out = (int)node;
    if (node < 0) return GOMEL_ERR_EMPTY;                                       // single-node machine, or not reported
    if (!read_small_file(root + "/devices/system/node/node" + std::to_string(node) + "/cpulist", txt) || !parse_cpulist(txt, cpus)) return GOMEL_ERR_EMPTY;
    return GOMEL_OK;
}

