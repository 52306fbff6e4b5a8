// Micro-benchmark: issue rate of plain vs packed fp32 VALU, f64, sqrt/rcp on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void k(float *out, long long *cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const float m = 1.0001f, c = 0.5f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 8 independent plain fma
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        } else if (MODE == 1) {  // 4 packed fma (8 lane-ops... 8 results)
            f2 mm = {m, m}, cc = {c, c};
            p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc);
            p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc);
        } else if (MODE == 2) {  // 4 f64 fma
            d0 = __builtin_fma(d0, 1.0001, 0.5); d1 = __builtin_fma(d1, 1.0001, 0.5); d2 = __builtin_fma(d2, 1.0001, 0.5); d3 = __builtin_fma(d3, 1.0001, 0.5);
        } else if (MODE == 3) {  // 4 v_sqrt_f32
            a0 = __builtin_amdgcn_sqrtf(a0 + c); a1 = __builtin_amdgcn_sqrtf(a1 + c); a2 = __builtin_amdgcn_sqrtf(a2 + c); a3 = __builtin_amdgcn_sqrtf(a3 + c);
        } else if (MODE == 4) {  // 1 dependent chain of plain fma (8 deep)
            a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c);
            a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE> void run(const char *name, int instr_per_iter, int threads) {
    float *out; long long *cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s threads/block=%4d (waves/SIMD=%d): %.2f cycles per instr per wave-slot, %.2f cycles per instr per SIMD\n", name, threads, threads / 256,
           (double)c / iters / instr_per_iter, (double)c / iters / instr_per_iter / (threads / 256.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        run<0>("plain v_fma_f32 x8", 8, threads);
        run<1>("v_pk_fma_f32 x4", 4, threads);
        run<2>("v_fma_f64 x4", 4, threads);
        run<3>("v_sqrt_f32+add x4", 8, threads);
        run<4>("dependent v_fma x8", 8, threads);
    }
    return 0;
}
