// Microbenchmark: cycles per v_pk_add_f32 / v_pk_fma_f32 / v_add_f32 at dependency distance D (D independent chains, round robin).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int D, int KIND>
__global__ void k(v2f* out, long long* cyc, int iters) {
    v2f a[D];
    for (int i = 0; i < D; i++) a[i] = v2f{(float)threadIdx.x + i, 1.0f};
    v2f b = v2f{1.0001f, 0.9999f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                else if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                else asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    v2f s = a[0];
    for (int i = 1; i < D; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int D, int KIND> void run(const char* name, int threads, int blocks) {
    v2f* out; long long* cyc;
    hipMalloc(&out, sizeof(v2f) * threads * blocks); hipMalloc(&cyc, 8 * blocks);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<D, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<D, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double c = (double)h[0] / ((double)iters * 16 * D);
    printf("%-10s D=%d waves/SIMD=%d: %.2f ticks per instruction per wave -> SIMD issues one every %.2f ticks\n", name, D, threads / 256, c, c / (threads / 256)); printf("      kernel %.1f us -> %.2f ns per instruction per wave, tick = %.3f ns\n", ms * 1e3, ms * 1e6 / ((double)iters * 16 * D), ms * 1e6 / (double)h[0]);
    hipFree(out); hipFree(cyc);
}

int main() {
    // one block per CU (256 blocks): 256 threads = 1 wave/SIMD, 512 = 2 waves/SIMD
    for (int threads : {256, 512, 1024}) {
        run<1, 0>("pk_add", threads, 256); run<2, 0>("pk_add", threads, 256); run<3, 0>("pk_add", threads, 256); run<4, 0>("pk_add", threads, 256);
        run<1, 1>("pk_fma", threads, 256); run<2, 1>("pk_fma", threads, 256); run<4, 1>("pk_fma", threads, 256);
        run<1, 2>("add_f32", threads, 256); run<2, 2>("add_f32", threads, 256); run<4, 2>("add_f32", threads, 256);
        run<1, 3>("pk_opsel", threads, 256); run<2, 3>("pk_opsel", threads, 256); run<4, 3>("pk_opsel", threads, 256);
    }
    return 0;
}
