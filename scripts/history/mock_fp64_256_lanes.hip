// Mock of a 256-lane, 8-points-per-lane fp64 fused Griffin-Lim frame loop (N 4096): instruction mix and LDS / HBM traffic only, no correct math.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));
#define DEV __device__ __forceinline__
DEV v2d cmul(v2d a, v2d b) { return v2d{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <bool TWREG> DEV void dft8(v2d* v) {   // ~ radix-8 butterfly op count: 24 complex adds + 2 constant twiddles + 1 rotation
    v2d u0[4], u1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { u0[j] = v[j] + v[j + 4]; u1[j] = v[j] - v[j + 4]; }
    u1[1] = v2d{(u1[1].x + u1[1].y) * 0.70710678, (u1[1].y - u1[1].x) * 0.70710678};
    u1[3] = v2d{(u1[3].y - u1[3].x) * 0.70710678, -(u1[3].x + u1[3].y) * 0.70710678};
    u1[2] = v2d{u1[2].y, -u1[2].x};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        v2d* a = h ? u1 : u0;
        v2d s02 = a[0] + a[2], d02 = a[0] - a[2], s13 = a[1] + a[3], d13 = a[1] - a[3];
        d13 = v2d{d13.y, -d13.x};
        v[0 + h] = s02 + s13; v[2 + h] = d02 + d13; v[4 + h] = s02 - s13; v[6 + h] = d02 - d13;
    }
}
constexpr int T = 256, FR = 2;          // lanes per frame, frames per workgroup
constexpr int SLOTS = 2048 + 64;       // exchange area per frame (padded)
template <bool TWREG>
__global__ __launch_bounds__(T * FR, 1) void k_mock(const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ mag, const v2d* __restrict__ tab, int frames_per_stream) {
    extern __shared__ v2d lds[];
    v2d* my = lds + (threadIdx.x / T) * SLOTS;
    v2d* win = lds + FR * SLOTS;                  // [2048]
    v2d* tw = win + 2048;                         // [3][7][256] pass twiddles when not in registers
    const int t = threadIdx.x % T;
    for (int i = threadIdx.x; i < 2048; i += T * FR) win[i] = tab[i];
    if (!TWREG) for (int i = threadIdx.x; i < 3 * 7 * 256; i += T * FR) tw[i] = tab[2048 + i];
    __syncthreads();
    v2d twr[3][7];
    if (TWREG) {
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 7; q++) twr[p][q] = tab[2048 + (p * 7 + q) * 256 + t];
    }
    v2d wn[4];
#pragma unroll
    for (int s = 0; s < 4; s++) wn[s] = tab[s * 256 + t];
    const long stream = (long)blockIdx.x * FR + threadIdx.x / T;
    const v2d* xin_p = reinterpret_cast<const v2d*>(x) + stream * frames_per_stream * 640;
    v2d* xo = reinterpret_cast<v2d*>(y) + stream * frames_per_stream * 640;
    const double* mg = mag + stream * frames_per_stream * 2048;
    v2d xin[8], acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { xin[j] = xin_p[t + 256 * j]; acc[j] = v2d{0, 0}; }
    for (int g = 0; g < frames_per_stream; g++) {
        double m[9];
#pragma unroll
        for (int e = 0; e < 8; e += 2) { v2d q = *reinterpret_cast<const v2d*>(mg + (long)g * 2048 + (e / 2 * 256 + t) * 2); m[e] = q.x; m[e + 1] = q.y; }
        m[8] = mg[(long)g * 2048 + (t & 3)];
        v2d xl[3];
#pragma unroll
        for (int jj = 0; jj < 3; jj++) xl[jj] = xin_p[(long)(g + 1) * 640 + t + 256 * jj + 1408];
        v2d v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = xin[j] * win[t + 256 * j];
        // forward: 3 x (radix-8, twiddle, exchange) + final radix-4 pair
#pragma unroll
        for (int p = 0; p < 3; p++) {
            dft8<TWREG>(v);
#pragma unroll
            for (int q = 1; q < 8; q++) v[q] = cmul(v[q], TWREG ? twr[p][q - 1] : tw[(p * 7 + q - 1) * 256 + t]);
#pragma unroll
            for (int q = 0; q < 8; q++) my[q * 260 + t] = v[q];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = my[q * 260 + ((t + 64 * (p + 1)) & 255)];
            __syncthreads();
        }
        dft8<TWREG>(v);   // stands for the two mirrored radix-4 butterflies
        // spectral step: 4 bin pairs
#pragma unroll
        for (int s = 0; s < 4; s++) {
            v2d a = v[s], b = v[7 - s];
            v2d bc = v2d{b.x, -b.y}, pp = a + bc, d = a - bc, q = cmul(wn[s], v2d{d.y, -d.x}), xk = pp + q, xc = pp - q;
            double pk = xk.x * xk.x + xk.y * xk.y, pc = xc.x * xc.x + xc.y * xc.y;
            double rk = __builtin_amdgcn_rsq(pk), rc = __builtin_amdgcn_rsq(pc);
            double gk = pk * rk, hk = 0.5 * rk, r1 = __builtin_fma(-hk, gk, 0.5); gk = __builtin_fma(gk, r1, gk); hk = __builtin_fma(hk, r1, hk); r1 = __builtin_fma(-hk, gk, 0.5); hk = __builtin_fma(hk, r1, hk);
            double gc = pc * rc, hc = 0.5 * rc, r2 = __builtin_fma(-hc, gc, 0.5); gc = __builtin_fma(gc, r2, gc); hc = __builtin_fma(hc, r2, hc); r2 = __builtin_fma(-hc, gc, 0.5); hc = __builtin_fma(hc, r2, hc);
            v2d u = xk * (hk * m[s] * 2.0), w = xc * (hc * m[4 + s] * 2.0);
            v2d e = u + w, o = cmul(u - w, v2d{wn[s].x, -wn[s].y}), io = v2d{-o.y, o.x};
            v[s] = e + io; v2d tt = e - io; v[7 - s] = v2d{tt.x, -tt.y};
        }
        v[0].x += m[8];
        // inverse
        dft8<TWREG>(v);
#pragma unroll
        for (int p = 0; p < 3; p++) {
#pragma unroll
            for (int q = 0; q < 8; q++) my[q * 260 + ((t + 64 * (p + 1)) & 255)] = v[q];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = my[q * 260 + t];
            __syncthreads();
#pragma unroll
            for (int q = 1; q < 8; q++) v[q] = cmul(v[q], TWREG ? twr[2 - p][q - 1] : tw[((2 - p) * 7 + q - 1) * 256 + t]);
            dft8<TWREG>(v);
        }
        // window, overlap-add with a lane-dependent row shift (2 or 3 rows), store 2.5 rows
        const bool hi = t >= 128;
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += v[j] * win[t + 256 * j];
        xo[(long)g * 640 + t] = acc[0]; xo[(long)g * 640 + 256 + t] = acc[1];
        if (hi) xo[(long)g * 640 + 384 + t] = acc[2];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            acc[j] = hi ? acc[j + 3] : acc[j + 2];
            xin[j] = hi ? xin[j + 3] : xin[j + 2];
        }
        acc[5] = hi ? v2d{0, 0} : acc[7]; acc[6] = v2d{0, 0}; acc[7] = v2d{0, 0};
        xin[5] = hi ? xl[0] : xin[7]; xin[6] = hi ? xl[1] : xl[0]; xin[7] = hi ? xl[2] : xl[1];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) xo[(long)frames_per_stream * 640 + t + 256 * j] = acc[j] + xin[j];
}
int main(int argc, char** argv) {
    const int streams = 2048, G = 50;                   // 102 400 frames
    const size_t nx = (size_t)streams * G * 1280 + 8192, nm = (size_t)streams * G * 2048 + 64;
    double *x, *y, *mag; v2d* tab;
    hipMalloc(&x, nx * 8); hipMalloc(&y, nx * 8); hipMalloc(&mag, nm * 8); hipMalloc(&tab, (2048 + 21 * 256 + 64) * 16);
    hipMemset(x, 0, nx * 8); hipMemset(mag, 0, nm * 8); hipMemset(tab, 0, (2048 + 21 * 256 + 64) * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int variant = 0; variant < 2; variant++) {
        const size_t lds = (size_t)(FR * SLOTS + 2048 + (variant ? 0 : 21 * 256)) * 16;
        auto kern = variant ? k_mock<true> : k_mock<false>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, dim3(streams / FR), dim3(T * FR), lds, 0, x, y, mag, tab, G);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("variant %s (LDS %zu KB): %.4f ms per launch of 102400 frames (err %d)\n", variant ? "twiddles in registers" : "twiddles in LDS", lds / 1024, ms / 10, (int)hipGetLastError());
        }
    }
    return 0;
}
