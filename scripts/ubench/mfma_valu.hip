// Microbenchmark: does ONE wave of a CDNA4 SIMD overlap its MFMAs with its own VALU / transcendental instructions?
// (design input: wavenet_wg issues the skip GEMM between the stages of the gate arithmetic)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}

// MODE 0: 16 independent MFMAs; 1: 48 fma; 2: 16 exp; 3: MFMA then 3 fma, x16; 4: MFMA then 1 exp, x16;
//      5: 16 MFMA clustered then 48 fma; 6: MFMA + exp + 2 fma x16
template <int MODE> __global__ __launch_bounds__(64, 1) void k(float* out, unsigned long long* tout, int iters) {
    float x[16], y[16];
    floatx4 acc[8];
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    for (int i = 0; i < 16; i++) { x[i] = threadIdx.x * 0.001f + i * 0.01f; y[i] = 1.0f + i; }
    for (int i = 0; i < 8; i++) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0 || MODE == 3 || MODE == 4 || MODE == 6)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a), "v"(b));
            if (MODE == 1 || MODE == 3) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(1.0001f));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[(i + 5) & 15]) : "v"(1.0001f));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[(i + 10) & 15]) : "v"(1.0001f));
            }
            if (MODE == 2 || MODE == 4 || MODE == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 6) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(1.0001f));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[(i + 5) & 15]) : "v"(1.0001f));
            }
        }
        if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < 48; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[i & 15]) : "v"(1.0001f));
        }
    }
    const unsigned long long t1 = now();
    float s = 0;
    for (int i = 0; i < 16; i++) s += x[i] + y[i];
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][3];
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
    printf("%-44s %8.1f clk per iteration\n", name, (double)t / iters);
}

int main() {
    float* out; unsigned long long* tout;
    CHECK(hipMalloc(&out, 256)); CHECK(hipMalloc(&tout, 8));
    run<0>("16 mfma_16x16x32_f16 (independent)", out, tout);
    run<1>("48 v_fma_f32", out, tout);
    run<2>("16 v_exp_f32", out, tout);
    run<3>("16 x (mfma, 3 fma)", out, tout);
    run<4>("16 x (mfma, exp)", out, tout);
    run<5>("16 mfma then 48 fma", out, tout);
    run<6>("16 x (mfma, exp, 2 fma)", out, tout);
    return 0;
}
