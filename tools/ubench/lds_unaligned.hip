// ds_read_b128 from 4-byte-aligned (not 16-byte-aligned) LDS addresses on gfx950: is it correct, and what does it cost?
// Each lane reads 4 consecutive floats starting at index base(lane) + 4 * t; bases are per-lane (groups of 8 lanes
// share one, as the lanes of one cell do in the density pass) and either multiples of 4 or not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float4 lds_read4(const float *p) {
    float4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return v;
}

template <int MODE> __global__ __launch_bounds__(256) void k(const int *bases, float *out, int iters, long long *cyc) {
    __shared__ float buf[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 256) buf[i] = (float)i;
    __syncthreads();
    const int base = bases[threadIdx.x];
    float acc = 0.f;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        float4 v[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const float *p = buf + ((base + 4 * t + 37 * it) & 2047);
            if (MODE == 0) v[t] = *reinterpret_cast<const float4 *>(buf + (((base + 4 * t + 37 * it) & 2047) & ~3));
            else if (MODE == 1) v[t] = lds_read4(p);
            else { v[t].x = p[0]; v[t].y = p[1]; v[t].z = p[2]; v[t].w = p[3]; }
        }
        if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (MODE == 1) asm volatile("" : "+v"(v[t].x), "+v"(v[t].y), "+v"(v[t].z), "+v"(v[t].w));
            acc += v[t].x + 2.f * v[t].y + 3.f * v[t].z + 4.f * v[t].w;
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256 * 8, iters = 200;
    std::vector<int> hb(256);
    int *db; float *dout; long long *dc;
    CK(hipMalloc(&db, 256 * 4)); CK(hipMalloc(&dout, blocks * 256 * 4)); CK(hipMalloc(&dc, blocks * 8));
    for (int align = 0; align < 2; align++) {
        for (int l = 0; l < 256; l++) hb[l] = (l / 8) * 8 + (align ? 0 : (l / 8) % 4 ? ((l / 8) % 4) : 1);  // shared by 8 lanes
        if (align) for (int l = 0; l < 256; l++) hb[l] = (l / 8) * 8;
        CK(hipMemcpy(db, hb.data(), 256 * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 3; mode++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, db, dout, iters, dc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, db, dout, iters, dc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, db, dout, iters, dc);
                hipEventRecord(b);
                CK(hipDeviceSynchronize());
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            std::vector<float> ho(256);
            CK(hipMemcpy(ho.data(), dout, 256 * 4, hipMemcpyDeviceToHost));
            // expected value for lane 0..255 computed on the host
            int bad = 0;
            for (int l = 0; l < 256; l++) {
                float acc = 0.f;
                for (int it = 0; it < iters; it++) for (int t = 0; t < 8; t++) {
                    int i = (hb[l] + 4 * t + 37 * it) & 2047;
                    if (mode == 0) i &= ~3;
                    acc += (float)i + 2.f * (i + 1) + 3.f * (i + 2) + 4.f * (i + 3);
                }
                if (acc != ho[l]) bad++;
            }
            printf("bases %s, mode %d (%s): %.3f ms, wrong lanes %d\n", align ? "aligned" : "unaligned", mode,
                   mode == 0 ? "float4 read of the aligned group" : mode == 1 ? "ds_read_b128 at the address" : "compiler's choice for 4 floats", ms, bad);
        }
    }
    return 0;
}
