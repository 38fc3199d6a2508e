// Microbenchmark: do transcendental (v_exp_f32 / v_rcp_f32) and plain VALU (v_fma_f32) instructions of
// ONE wave overlap on a CDNA4 SIMD, and does their order matter?  (design input: the gate block)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}

// MODE 0: 32 exp2 only; 1: 96 fma only; 2: clustered (32 exp then 96 fma); 3: interleaved 1 exp : 3 fma
template <int MODE> __global__ __launch_bounds__(64, 1) void k(float* out, unsigned long long* tout, int iters) {
    float x[32], y[32];
    for (int i = 0; i < 32; i++) { x[i] = threadIdx.x * 0.001f + i * 0.01f; y[i] = 1.0f + i; }
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int i = 0; i < 32; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(1.0001f));
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(1.0001f));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[(i + 11) & 31]) : "v"(1.0001f));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[(i + 22) & 31]) : "v"(1.0001f));
            }
        }
    }
    const unsigned long long t1 = now();
    float s = 0;
    for (int i = 0; i < 32; i++) s += x[i] + y[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) tout[0] = t1 - t0;
}

template <int MODE> void run(const char* name, float* out, unsigned long long* tout) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, tout, 10);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, tout, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long t;
    CHECK(hipMemcpy(&t, tout, 8, hipMemcpyDeviceToHost));
    printf("%-28s %8.1f clk per iteration\n", name, (double)t / iters);
}

int main() {
    float* out; unsigned long long* tout;
    CHECK(hipMalloc(&out, 256)); CHECK(hipMalloc(&tout, 8));
    run<0>("32 v_exp_f32", out, tout);
    run<1>("96 v_fma_f32", out, tout);
    run<2>("clustered 32 exp + 96 fma", out, tout);
    run<3>("interleaved 1 exp : 3 fma", out, tout);
    return 0;
}
