// Microbenchmark: how fast can ONE wavefront (or a few) stream a 1.7 MB L2-resident weight blob
// into VGPRs with a software prefetch ring of depth PF?  (design input for wn_kernels.hpp)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int PF, bool MFMA>
__global__ __launch_bounds__(64, 1) void stream_kernel(const floatx4* __restrict__ w, int nfrag, int iters, float* out) {
    const int lane = threadIdx.x;
    const floatx4* base = w + lane;
    floatx4 buf[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) buf[i] = base[(size_t)i * 64];
    floatx4 acc = {0, 0, 0, 0};
    half8 b; for (int e = 0; e < 8; e++) b[e] = (_Float16)0.01f;
    for (int it = 0; it < iters; it++) {
        for (int f = 0; f < nfrag; f += PF) {
#pragma unroll
            for (int i = 0; i < PF; i++) {
                floatx4 a = buf[i];
                int nf = f + i + PF; if (nf >= nfrag) nf -= nfrag;
                buf[i] = base[(size_t)nf * 64];
                if (MFMA) {
                    half8 ah = __builtin_bit_cast(half8, a);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b, acc, 0, 0, 0);
                } else {
                    acc += a;
                }
            }
        }
    }
    out[blockIdx.x * 64 + lane] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int PF, bool MFMA> void run(const floatx4* w, int nfrag, int iters, int nblocks, float* out) {
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    hipLaunchKernelGGL((stream_kernel<PF, MFMA>), dim3(nblocks), dim3(64), 0, 0, w, nfrag, 2, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(t0));
    hipLaunchKernelGGL((stream_kernel<PF, MFMA>), dim3(nblocks), dim3(64), 0, 0, w, nfrag, iters, out);
    CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1));
    float ms; CHECK(hipEventElapsedTime(&ms, t0, t1));
    double bytes = (double)nfrag * 1024 * iters;
    printf("PF=%2d mfma=%d blocks=%4d: %8.3f ms  %.2f us/pass  %.1f GB/s per wave, %.1f GB/s total\n", PF, (int)MFMA,
           nblocks, ms, 1e3 * ms / iters, bytes / ms / 1e6, bytes * nblocks / ms / 1e6);
}

int main() {
    const int nfrag = 1728;  // 1.73 MB, a multiple of 8,12,16,24,32,36,48,54,64? -> use divisors below
    floatx4* w; float* out;
    CHECK(hipMalloc(&w, (size_t)nfrag * 1024)); CHECK(hipMemset(w, 0, (size_t)nfrag * 1024));
    CHECK(hipMalloc(&out, 4096 * 64 * 4));
    const int iters = 200;
    for (int nb : {1, 256, 512, 1024}) {
        run<8, false>(w, nfrag, iters, nb, out);
        run<16, false>(w, nfrag, iters, nb, out);
        run<32, false>(w, nfrag, iters, nb, out);
        run<48, false>(w, nfrag, iters, nb, out);
        run<32, true>(w, nfrag, iters, nb, out);
        run<48, true>(w, nfrag, iters, nb, out);
    }
    return 0;
}
