// gl_types.hpp -- plain data types and constants shared by the HIP kernels (gl_kernels.hpp) and the device-free host logic
// (gomel_host.hpp): nothing here needs the HIP headers, so the host logic compiles as plain C++ (g++ / clang++, sanitizers).
#pragma once
#include <stdint.h>

struct BinRow {  // one linear bin of undomel (or one mel band of domel): data-independent index math, built in fp64 on host
    int32_t lo, hi;
    int32_t mode;  // 0 copy band lo; 1 two-tap lo,hi with modlo; 2 sum lo..hi-1 divided by div
    int32_t pad_;
    double modlo, div;
};

constexpr int kBigMaxN = 65536;          // longest frame (gl_big.hpp: 2N-1 <= 131072 = 16 x 8192)
static inline bool gomel_supported_n(int n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096; }
// what gl_any.hpp covers besides: every other length (odd ones too) up to 8192 (radix-2; Bluestein, above 4096 in its split form);
// gl_big.hpp: 8193 ... 65536 (Bluestein through HBM)
static inline bool gomel_generic_n(int n) {
    if (n < 4 || gomel_supported_n(n)) return false;
    return n <= kBigMaxN;
}
static inline bool gomel_big_n(int n) { return n > 8192 && n <= kBigMaxN; }

struct UttDesc {  // one utterance of a batch
    int64_t xoff;     // offset of its signal in the x / out buffers (samples)
    int64_t fbase;    // index of its first frame in the frame-major buffers
    int32_t frames;
    int32_t has_x0;
    int64_t len;      // L = N + (F-1) hop
    uint64_t seed;
};

// One segment of the fused kernels (k_gl_fused and its forms; the plan is made on the host, gomel_host.hpp plan_segments):
// a segment OWNS the samples [s0, s0 + G*hop) (the last one of an utterance: to the utterance's end).
struct SegDesc {
    int64_t s0;       // offset of the segment's first sample in the x buffers
    int64_t frame0;   // global index of its first frame (row of mag / nyq)
    int32_t G;        // frames in the segment
    int32_t has_prev; // not the first segment of its utterance -> add tail[seg-1] over the first 16-SH rows
    int32_t has_next; // not the last -> rows >= G*SH belong to the next segment: read x + tail[seg], write tail[seg]
    int32_t pad_;
};

constexpr int kFusedThreads = 256;
constexpr int kMaxSegFrames = 64;   // a wave fetches the Nyquist magnitudes of all its segment's frames with ONE load (lane = frame)
constexpr int kPersistMaxNb = 15;   // neighbours on each side the persistent kernel handles: (N-1)/hop <= 15, i.e. hop >= N/16
constexpr int kMaxMels = 16384;              // NumMels beyond this cannot keep one frame's exp() row in the 160 KiB LDS
