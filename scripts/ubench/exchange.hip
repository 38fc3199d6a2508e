// Microbenchmark: latency of the primitives on wavenet_wg's per-layer critical path, 4 waves of one workgroup:
//  (a) LDS exchange round trip: ds_write_b64, wait, s_barrier, 2 x ds_read_b128, wait
//  (b) dependent MFMA chain (same accumulator), (c) dependent exp2 -> add -> rcp -> fma -> mul chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}

template <int MODE> __global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* tout, int iters) {
    __shared__ __attribute__((aligned(16))) char buf[8192];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    floatx4 v = {threadIdx.x * 0.001f, 1.f, 2.f, 3.f};
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * i); b[i] = (_Float16)(0.5f); }
    float x = threadIdx.x * 0.001f;
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 3) {
            half4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *(half4*)(buf + (((w >> 1) * 64 + lane) << 4) + ((w & 1) << 3)) = h;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            half8 f0 = *(const half8*)(buf + ((0 * 64 + lane) << 4));
            half8 f1 = *(const half8*)(buf + ((1 * 64 + lane) << 4));
            if (MODE == 0) {
                v[0] += (float)f0[0] + (float)f1[7];
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, f0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, f1, acc, 0, 0, 0);
                v = acc;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (second barrier: buffer reuse, as in a real exchange pair)
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float e = __builtin_amdgcn_exp2f(x);
                float r = __builtin_amdgcn_rcpf(e + 1.0f);
                x = (1.0f - 2.0f * r) * r;
            }
        }
        if (MODE == 4) {   // mfma result -> VALU use -> next mfma operand (the D -> cvt -> B turn-around)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
                b[0] = (_Float16)acc[0];
            }
        }
    }
    const unsigned long long t1 = now();
    out[threadIdx.x] = v[0] + acc[0] + x + (float)b[0];
    if (threadIdx.x == 0) tout[0] = t1 - t0;
}

template <int MODE> void run(const char* name, float* out, unsigned long long* tout, double per) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, out, tout, 10);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, out, tout, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long t;
    CHECK(hipMemcpy(&t, tout, 8, hipMemcpyDeviceToHost));
    printf("%-64s %8.1f clk\n", name, (double)t / iters / per);
}

int main() {
    float* out; unsigned long long* tout;
    CHECK(hipMalloc(&out, 1024)); CHECK(hipMalloc(&tout, 8));
    run<0>("exchange: cvt, ds_write, barrier, 2 ds_read_b128, use, barrier", out, tout, 1);
    run<3>("same + 2 dependent MFMAs on the fragments", out, tout, 1);
    run<1>("dependent MFMA (same accumulator), per MFMA", out, tout, 8);
    run<2>("exp2 -> add -> rcp -> fma -> mul chain, per gate value", out, tout, 4);
    run<4>("mfma -> cvt of its result -> next mfma's B operand, per turn", out, tout, 4);
    return 0;
}
