// Microbenchmark (round 3): what does a trickle of HBM loads cost a wave whose vector-memory queue also carries an
// L2-resident weight stream?  Models wavenet_wg's queue: 4 waves per CU, each streams 18 KiB of weights per "layer" through
// a 9-deep register ring (one 1-KiB fragment per "take", some arithmetic per take), and requests NH 1-KiB pieces of a
// never-reused stream two layers ahead.  Variants of where those pieces come from / go to:
//   0 none   1 HBM -> VGPRs (what wavenet_wg does)   2 the same from two L2-resident addresses ("hot")
//   3 HBM -> LDS by LDS-DMA (buffer_load ... lds), read back with ds_read two layers later
//   4 the same DMA from L2-resident addresses
//   hipcc --offload-arch=gfx950 -O3 hbmq.hip -o hbmq && ./hbmq
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int PF = 9, FL = 18, NH = 4;

template <int MODE, int WORK>
__global__ __launch_bounds__(256, 1) void k(const char* w, const char* big, size_t bigStride, int layers, float* out) {
    __shared__ __attribute__((aligned(16))) char dma[2][4][NH][1024];      // [parity][wave][piece]
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned laneOff = lane * 16u;
    const rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(w + (size_t)wv * 512 * 1024), 0, -1, 0x00020000);
    const char* mine = big + ((size_t)blockIdx.x * 4 + wv) * bigStride;
    uintx4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) ring[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, laneOff, i * 1024, 0);
    uintx4 hA[NH], hB[NH];
#pragma unroll
    for (int i = 0; i < NH; i++) hA[i] = hB[i] = uintx4{0, 0, 0, 0};
    floatx4 acc = {0, 0, 0, 0};
    float x = (float)lane;
    int pos = 0;                                    // fragment position in the 432-fragment weight stream of this wave
    auto layer = [&](int l, uintx4 (&hUse)[NH], int par) {
        // consume what was requested two layers ago
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < NH; i++) acc[i & 3] += __builtin_bit_cast(floatx4, hUse[i])[0];
        } else if (MODE >= 3) {
#pragma unroll
            for (int i = 0; i < NH; i++) acc[i & 3] += (*(const floatx4*)(&dma[par][wv][i][lane * 16]))[0];
        }
#pragma unroll
        for (int i = 0; i < FL; i++) {
            const uintx4 a = ring[i % PF];
            int np = pos + i + PF;
            if (np >= 432) np -= 432;
            ring[i % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, laneOff, np * 1024, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[i & 3] += __builtin_bit_cast(floatx4, a)[0];
#pragma unroll
            for (int k2 = 0; k2 < WORK; k2++) x = __builtin_fmaf(x, 1.0001f, 0.5f);     // ~WORK * 4..5 clk of dependent VALU
            if (i == 12 && MODE != 0) {             // the request for two layers ahead, three quarters into the layer
                const bool hot = MODE == 2 || MODE == 4;
                const rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)(mine + (hot ? (size_t)(l & 1) : (size_t)l) * (NH * 1024)), 0, -1, 0x00020000);
#pragma unroll
                for (int q = 0; q < NH; q++) {
                    if (MODE <= 2) hUse[q] = __builtin_amdgcn_raw_buffer_load_b128(rh, laneOff, q * 1024, 2);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (__attribute__((address_space(3))) void*)&dma[par][wv][q][0], 16, laneOff, q * 1024, 0, 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        pos += FL;
        if (pos >= 432) pos -= 432;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (the layer's exchange barrier)
    };
    for (int l = 0; l < layers; l += 2) {
        layer(l, hA, 0);
        layer(l + 1, hB, 1);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + x;
}

template <int MODE, int WORK> void run(const char* w, const char* big, size_t stride, int layers, int blocks, float* out, const char* name) {
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    hipLaunchKernelGGL((k<MODE, WORK>), dim3(blocks), dim3(256), 0, 0, w, big, stride, 64, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(t0));
    hipLaunchKernelGGL((k<MODE, WORK>), dim3(blocks), dim3(256), 0, 0, w, big, stride, layers, out);
    CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1));
    float ms; CHECK(hipEventElapsedTime(&ms, t0, t1));
    printf("work=%2d blocks=%3d %-28s %8.3f us per layer\n", WORK, blocks, name, 1e3 * ms / layers);
}

int main() {
    const int layers = 4000;
    char *w, *big; float* out;
    CHECK(hipMalloc(&w, (size_t)4 * 512 * 1024)); CHECK(hipMemset(w, 0, (size_t)4 * 512 * 1024));
    const size_t stride = (size_t)(layers + 64) * NH * 1024;              // per wave: a fresh 4 KiB per layer
    CHECK(hipMalloc(&big, stride * 4 * 256)); CHECK(hipMemset(big, 0, stride * 4 * 256));
    CHECK(hipMalloc(&out, 256 * 256 * 4));
    for (int blocks : {1, 256}) {
        run<0, 30>(w, big, stride, layers, blocks, out, "no extra loads");
        run<1, 30>(w, big, stride, layers, blocks, out, "HBM -> VGPR");
        run<2, 30>(w, big, stride, layers, blocks, out, "L2-resident -> VGPR");
        run<3, 30>(w, big, stride, layers, blocks, out, "HBM -> LDS (DMA)");
        run<4, 30>(w, big, stride, layers, blocks, out, "L2-resident -> LDS (DMA)");
    }
    return 0;
}
