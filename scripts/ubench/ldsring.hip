// Microbenchmark for the loader/consumer design: 4 loader waves stream weight fragments
// global -> LDS with LDS-DMA (global_load_lds_dwordx4) into a ring of NS slots x CH KiB; 4 consumer
// waves (one per SIMD) each read EVERY fragment (ds_read_b128) and feed an MFMA; one s_barrier per
// chunk. Reports KiB-fragments per microsecond per CU and the implied time for 1.76 MB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CH, int NS, int MFMA_PER_FRAG>
__global__ __launch_bounds__(512, 2) void ring_kernel(const char* __restrict__ w, int nchunks, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wv >= 4;
    const int total = nchunks * iters;
    floatx4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    half8 b; for (int e = 0; e < 8; e++) b[e] = (_Float16)0.01f;
    constexpr int Q = CH / 4;   // fragments per loader wave per chunk
    if (loader) {
        const int lw = wv - 4;
        // prologue: chunks 0..NS-2
        for (int c = 0; c < NS - 1; c++) {
            const char* src = w + ((size_t)(c % nchunks) * CH + lw * Q) * 1024 + lane * 16;
            char* dst = lds + ((c % NS) * CH + lw * Q) * 1024;
#pragma unroll
            for (int q = 0; q < Q; q++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();   // chunks 0..NS-2 ready
        for (int c = 0; c < total; c++) {
            // slot of chunk c-1 is free (consumers passed the barrier that ended step c-1)
            const int cn = c + NS - 1;
            const char* src = w + ((size_t)(cn % nchunks) * CH + lw * Q) * 1024 + lane * 16;
            char* dst = lds + ((cn % NS) * CH + lw * Q) * 1024;
#pragma unroll
            for (int q = 0; q < Q; q++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
            // everything issued before this step has landed; this step's loads stay in flight
            if (Q == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (Q == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (Q == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
        }
    } else {
        bar();
        for (int c = 0; c < total; c++) {
            const char* slot = lds + (c % NS) * CH * 1024 + lane * 16;
#pragma unroll
            for (int f = 0; f < CH; f++) {
                half8 a = *(const half8*)(slot + f * 1024);
#pragma unroll
                for (int m = 0; m < MFMA_PER_FRAG; m++) acc[(f + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[(f + m) & 3], 0, 0, 0);
            }
            bar();
        }
    }
    if (!loader) out[blockIdx.x * 256 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int CH, int NS, int M> void run(const char* w, int nfrag, int nblocks, float* out) {
    const int nchunks = nfrag / CH, iters = 50;
    size_t ldsb = (size_t)NS * CH * 1024;
    CHECK(hipFuncSetAttribute((const void*)ring_kernel<CH, NS, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    hipLaunchKernelGGL((ring_kernel<CH, NS, M>), dim3(nblocks), dim3(512), ldsb, 0, w, nchunks, 2, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(t0));
    hipLaunchKernelGGL((ring_kernel<CH, NS, M>), dim3(nblocks), dim3(512), ldsb, 0, w, nchunks, iters, out);
    CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1));
    float ms; CHECK(hipEventElapsedTime(&ms, t0, t1));
    double frags = (double)nchunks * CH * iters;
    printf("CH=%2d NS=%d mfma/frag=%d blocks=%4d: %7.2f us per 1728-frag pass, %.1f GB/s per CU (each of 4 waves consumes all)\n",
           CH, NS, M, nblocks, 1e3 * ms / iters, frags * 1024 / ms / 1e6);
}

int main() {
    const int nfrag = 1728;
    char* w; float* out;
    CHECK(hipMalloc(&w, (size_t)nfrag * 1024)); CHECK(hipMemset(w, 0, (size_t)nfrag * 1024));
    CHECK(hipMalloc(&out, 4096 * 256 * 4));
    for (int nb : {1, 256}) {
        run<24, 4, 1>(w, nfrag, nb, out);
        run<24, 5, 1>(w, nfrag, nb, out);
        run<16, 6, 1>(w, nfrag, nb, out);
        run<32, 4, 1>(w, nfrag, nb, out);
        run<24, 4, 0>(w, nfrag, nb, out);
        run<24, 4, 2>(w, nfrag, nb, out);
    }
    return 0;
}
