// Microbenchmark (developer tool, round 5): what v_mfma_f64_4x4x4_4b_f64 costs ONE wave per SIMD in issue slots beside fp64 vector
// instructions, and whether "A = identity" turns it into a per-lane add (D = B + C) that can read / write accumulation registers
// (AGPRs) directly -- the question behind "overlap-add accumulators that never leave the AGPRs" for k_gl_fused_d.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64_probe scripts/mfma_f64_probe.hip ; run: ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// mode 0: 64 independent v_fma_f64                                  (the vector floor: 4 cycles each for a lone wave)
// mode 1: 64 v_fma_f64 with one MFMA (VGPR C/D) after every 4th     (16 MFMAs: their marginal issue cost)
// mode 2: 64 MFMAs back to back, four independent accumulators      (the matrix pipe's own rate)
// mode 3: 64 x (2 v_accvgpr_read + v_add_f64 + 2 v_accvgpr_write)   (an AGPR-resident accumulator updated by vector instructions)
// mode 4: 64 MFMAs with AGPR C/D, four accumulators                 (the same update on the matrix pipe)
// mode 5: 64 v_fma_f64 with one AGPR-C/D MFMA after every 2nd       (32 MFMAs hidden beside vector work?)
// mode 6: 128 v_accvgpr_read                                         (the parking read itself)
template <int MODE> __global__ __launch_bounds__(256, 1) void k_probe(double* out, unsigned long long* cycles, int iters) {
    __shared__ char pad[120 * 1024];   // one workgroup per CU: one wave per SIMD
    if (threadIdx.x == 9999) pad[0] = 1;
    const int lane = threadIdx.x & 63;
    double a = ((lane & 3) == ((lane >> 2) & 3)) ? 1.0 : 0.0;
    double b = 1.0 + lane * 0.001, c0 = 0.5, c1 = 0.25, c2 = 0.125, c3 = 2.0;
    double x0 = 1.0, x1 = 1.1, x2 = 1.2, x3 = 1.3, x4 = 1.4, x5 = 1.5, x6 = 1.6, x7 = 1.7, m = 1.0000001, ad = 1e-9;
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %0\n"
                 "v_accvgpr_write_b32 a4, %0\n v_accvgpr_write_b32 a5, %0\n v_accvgpr_write_b32 a6, %0\n v_accvgpr_write_b32 a7, %0" :: "v"(0) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define FMA8 "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n" \
             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
#define FMAOPS "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(ad)
        if constexpr (MODE == 0) {
            asm volatile(REP4(FMA8 FMA8) : FMAOPS);
        } else if constexpr (MODE == 1) {
#define F4A "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
#define F4B "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
            asm volatile(REP4(F4A "v_mfma_f64_4x4x4_4b_f64 %10, %14, %15, %10\n" F4B "v_mfma_f64_4x4x4_4b_f64 %11, %14, %15, %11\n"
                              F4A "v_mfma_f64_4x4x4_4b_f64 %12, %14, %15, %12\n" F4B "v_mfma_f64_4x4x4_4b_f64 %13, %14, %15, %13\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(ad), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a), "v"(b));
        } else if constexpr (MODE == 2) {
            asm volatile(REP16("v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %4, %5, %1\n"
                               "v_mfma_f64_4x4x4_4b_f64 %2, %4, %5, %2\n v_mfma_f64_4x4x4_4b_f64 %3, %4, %5, %3\n")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        } else if constexpr (MODE == 3) {
#define UPD(lo, hi) "v_accvgpr_read_b32 %0, a" #lo "\n v_accvgpr_read_b32 %1, a" #hi "\n v_add_f64 %2, %2, %3\n v_accvgpr_write_b32 a" #lo ", %0\n v_accvgpr_write_b32 a" #hi ", %1\n"
            int r0, r1;
            asm volatile(REP16(UPD(0, 1) UPD(2, 3) UPD(4, 5) UPD(6, 7)) : "=&v"(r0), "=&v"(r1), "+v"(x0) : "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        } else if constexpr (MODE == 4) {
            asm volatile(REP16("v_mfma_f64_4x4x4_4b_f64 a[0:1], %0, %1, a[0:1]\n v_mfma_f64_4x4x4_4b_f64 a[2:3], %0, %1, a[2:3]\n"
                               "v_mfma_f64_4x4x4_4b_f64 a[4:5], %0, %1, a[4:5]\n v_mfma_f64_4x4x4_4b_f64 a[6:7], %0, %1, a[6:7]\n")
                         :: "v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        } else if constexpr (MODE == 5) {
#define F2A "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n"
#define F2B "v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
#define F2C "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n"
#define F2D "v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
            asm volatile(REP4(REP4(F2A "v_mfma_f64_4x4x4_4b_f64 a[0:1], %10, %11, a[0:1]\n") REP4(F2B "v_mfma_f64_4x4x4_4b_f64 a[2:3], %10, %11, a[2:3]\n")
                              REP4(F2C "v_mfma_f64_4x4x4_4b_f64 a[4:5], %10, %11, a[4:5]\n") REP4(F2D "v_mfma_f64_4x4x4_4b_f64 a[6:7], %10, %11, a[6:7]\n"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(ad), "v"(a), "v"(b)
                         : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        } else if constexpr (MODE >= 10) {   // 64 independent instances of ONE instruction (eight destinations in turn)
#define ONE8(ins) ins " %0, %8\n " ins " %1, %8\n " ins " %2, %8\n " ins " %3, %8\n " ins " %4, %8\n " ins " %5, %8\n " ins " %6, %8\n " ins " %7, %8\n"
#define TWO8(ins) ins " %0, %8, %9\n " ins " %1, %8, %9\n " ins " %2, %8, %9\n " ins " %3, %8, %9\n " ins " %4, %8, %9\n " ins " %5, %8, %9\n " ins " %6, %8, %9\n " ins " %7, %8, %9\n"
            if constexpr (MODE == 10) asm volatile(REP4(ONE8("v_rsq_f64") ONE8("v_rsq_f64")) : FMAOPS);
            if constexpr (MODE == 11) asm volatile(REP4(TWO8("v_mul_f64") TWO8("v_mul_f64")) : FMAOPS);
            if constexpr (MODE == 12) asm volatile(REP4(TWO8("v_add_f64") TWO8("v_add_f64")) : FMAOPS);
            if constexpr (MODE == 13) asm volatile(REP4(TWO8("v_min_f64") TWO8("v_min_f64")) : FMAOPS);
            if constexpr (MODE == 14) asm volatile(REP4(ONE8("v_mov_b64") ONE8("v_mov_b64")) : FMAOPS);
            if constexpr (MODE == 15) asm volatile(REP16("v_cmp_class_f64 vcc, %8, %10\n v_cmp_class_f64 vcc, %9, %10\n v_cmp_gt_f64 vcc, %8, %9\n v_cmp_lt_f64 vcc, %8, %9\n") : FMAOPS, "v"(0x100) : "vcc");
            if constexpr (MODE == 16) asm volatile(REP4(ONE8("v_rcp_f64") ONE8("v_rcp_f64")) : FMAOPS);
            if constexpr (MODE == 17) asm volatile(REP4(ONE8("v_sqrt_f64") ONE8("v_sqrt_f64")) : FMAOPS);
        } else if constexpr (MODE == 6) {
            int r0, r1, r2, r3;
            asm volatile(REP16("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n"
                               "v_accvgpr_read_b32 %0, a4\n v_accvgpr_read_b32 %1, a5\n v_accvgpr_read_b32 %2, a6\n v_accvgpr_read_b32 %3, a7\n")
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) :: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
            x0 += r0 + r1 + r2 + r3;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0 + c1 + c2 + c3;
}

// D = A x B + C with A = "identity" (1 where lane % 4 == (lane / 4) % 4): is it D[lane] = B[lane] + C[lane]?
__global__ void k_layout(double* d_out) {
    const int lane = threadIdx.x & 63;
    const double a = ((lane & 3) == ((lane >> 2) & 3)) ? 1.0 : 0.0;
    const double b = 100.0 + lane, c = 0.001 * (lane + 1);
    d_out[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);        // A = identity pattern
    d_out[64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, c, 0, 0, 0);   // B = identity pattern
}

template <int MODE> void run(const char* what, int per_iter) {
    const int blocks = 256, iters = 200;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * sizeof(double)); hipMalloc(&cyc, blocks * 4 * sizeof(unsigned long long));
    for (int rep = 0; rep < 3; rep++) k_probe<MODE><<<blocks, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("mode %d  %-72s median %.2f cycles per block of %d (= %.2f per instruction group)\n", MODE, what, (double)h[h.size() / 2] / iters, per_iter, (double)h[h.size() / 2] / iters / per_iter);
    hipFree(out); hipFree(cyc);
}

int main() {
    double* d; hipMalloc(&d, 128 * sizeof(double));
    k_layout<<<1, 64>>>(d);
    double h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; l++) { okA &= h[l] == (100.0 + l) + 0.001 * (l + 1); okB &= h[64 + l] == (100.0 + l) + 0.001 * (l + 1); }
    printf("identity in A: D == B + C per lane: %s;  identity in B: D == A + C per lane: %s\n", okA ? "YES" : "no", okB ? "YES" : "no");
    if (!okA && !okB) for (int l = 0; l < 8; l++) printf("  lane %d: A-id %.6f  B-id %.6f  want %.6f\n", l, h[l], h[64 + l], (100.0 + l) + 0.001 * (l + 1));
    run<0>("64 v_fma_f64", 64);
    run<1>("64 v_fma_f64 + 16 MFMA (VGPR C/D), one per 4", 64);
    run<2>("64 MFMA back to back, 4 accumulators (VGPR)", 64);
    run<3>("64 x (2 accvgpr_read, v_add_f64, 2 accvgpr_write)", 64);
    run<4>("64 MFMA back to back, 4 accumulators (AGPR C/D)", 64);
    run<5>("64 v_fma_f64 + 32 MFMA (AGPR C/D), one per 2", 64);
    run<6>("128 v_accvgpr_read", 128);
    run<10>("64 v_rsq_f64", 64);
    run<11>("64 v_mul_f64", 64);
    run<12>("64 v_add_f64", 64);
    run<13>("64 v_min_f64", 64);
    run<14>("64 v_mov_b64", 64);
    run<15>("64 v_cmp (32 class, 32 gt/lt) f64", 64);
    run<16>("64 v_rcp_f64", 64);
    run<17>("64 v_sqrt_f64", 64);
    return 0;
}
