// Microbenchmark: what does ISSUING a 1-KiB global_load_dwordx4 cost a wave, as a function of how many
// are issued back to back (K), how long the wave then computes before the next burst (GAP), and how
// many waves of the CU stream concurrently?  (design input: placement of the weight refills)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}

template <int K>
__global__ __launch_bounds__(256, 1) void issue_kernel(const char* __restrict__ blob, size_t waveBytes, int iters, int gap,
                                                       unsigned long long* out, float* sink) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = blob + (size_t)w * waveBytes;
    floatx4 buf[K];
    floatx4 acc = {0, 0, 0, 0};
    float dummy = (float)lane;
    unsigned long long tIssue = 0, tWait = 0, tGap = 0;
    unsigned off = 0;
    const unsigned long long tStart = now();
    for (int it = 0; it < iters; it++) {
        const unsigned long long t0 = now();
#pragma unroll
        for (int i = 0; i < K; i++) {
            const unsigned vo = off + (unsigned)lane * 16u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(buf[i]) : "v"(vo), "s"(base) : "memory");
            off += 1024;
            if (off >= waveBytes) off = 0;
        }
        const unsigned long long t1 = now();
        // compute phase: a dependent chain of `gap` FMAs (4 clk each on a wave64)
        for (int g = 0; g < gap; g++) dummy = __builtin_fmaf(dummy, 1.0000001f, 0.5f);
        const unsigned long long t2 = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = now();
#pragma unroll
        for (int i = 0; i < K; i++) acc += buf[i];
        tIssue += t1 - t0;
        tGap += t2 - t1;
        tWait += t3 - t2;
    }
    const unsigned long long tEnd = now();
    if (lane == 0) {
        unsigned long long* o = out + (blockIdx.x * 4 + w) * 4;
        o[0] = tIssue; o[1] = tGap; o[2] = tWait; o[3] = tEnd - tStart;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + dummy;
}

template <int K> void run(const char* blob, size_t waveBytes, int gap, int nblocks, unsigned long long* out, float* sink) {
    const int iters = 400;
    hipLaunchKernelGGL((issue_kernel<K>), dim3(nblocks), dim3(256), 0, 0, blob, waveBytes, 20, gap, out, sink);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((issue_kernel<K>), dim3(nblocks), dim3(256), 0, 0, blob, waveBytes, iters, gap, out, sink);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[16];
    CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz on this part: report ticks and the derived per-iteration numbers of wave 0
    printf("K=%2d gap=%5d blocks=%4d | per iter (memtime ticks): issue %.2f  gap %.2f  wait %.2f  total %.2f | per load: issue %.3f total %.3f\n",
           K, gap, nblocks, (double)h[0] / iters, (double)h[1] / iters, (double)h[2] / iters, (double)h[3] / iters,
           (double)h[0] / iters / K, (double)h[3] / iters / K);
}

int main() {
    const size_t waveBytes = 432 * 1024;     // 4 waves x 432 KiB = the 1.73 MB weight blob of the C3 model
    char* blob; unsigned long long* out; float* sink;
    CHECK(hipMalloc(&blob, 4 * waveBytes)); CHECK(hipMemset(blob, 0, 4 * waveBytes));
    CHECK(hipMalloc(&out, 1024 * 16 * 8)); CHECK(hipMalloc(&sink, 1024 * 256 * 4));
    for (int nb : {1, 256}) {
        for (int gap : {0, 100, 400}) {
            run<1>(blob, waveBytes, gap, nb, out, sink);
            run<2>(blob, waveBytes, gap, nb, out, sink);
            run<4>(blob, waveBytes, gap, nb, out, sink);
            run<9>(blob, waveBytes, gap, nb, out, sink);
            run<18>(blob, waveBytes, gap, nb, out, sink);
        }
    }
    return 0;
}
