// Where does LDS-DMA put its data?  One wave copies a 1-KiB piece (lane l: 16 bytes holding the words 4l..4l+3 of a ramp) to
// LDS address DST with (a) buffer_load_dwordx4 ... lds, (b) global_load_lds_dwordx4, M0 = DST, then scans the whole 160 KiB
// for the ramp and prints where it landed.  hipcc --offload-arch=gfx950 -O3 ldsdma_addr.hip -o ldsdma_addr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ __launch_bounds__(64, 1) void k(const unsigned* ramp, unsigned dst, int mode, int imm, int* found) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned lane16 = threadIdx.x * 16u;
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64) ((unsigned*)lds)[i] = 0xdeadbeefu;
    __syncthreads();
    rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ramp, 0, -1, 0x00020000);
    if (mode == 0) {
        if (imm) asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds" ::"s"(dst), "v"(lane16), "s"(rs) : "memory");
        else asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(lane16), "s"(rs) : "memory");
    } else {
        if (imm) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" ::"s"(dst), "v"(lane16), "s"(ramp) : "memory");
        else asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lane16), "s"(ramp) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // find word value W0 (first word of what lane 0 loaded) and the address of lane 1's first word
    const unsigned w0 = imm ? 256u : 0u;        // an immediate offset of 1024 bytes moves the SOURCE by 256 words
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64) {
        unsigned v = ((unsigned*)lds)[i];
        if (v == w0) found[0] = i * 4;
        if (v == w0 + 4) found[1] = i * 4;
        if (v == w0 + 255) found[2] = i * 4;
    }
}
int main() {
    unsigned* ramp; int* found;
    CHECK(hipMalloc(&ramp, 1 << 16)); CHECK(hipMalloc(&found, 16));
    unsigned h[1 << 14]; for (int i = 0; i < (1 << 14); i++) h[i] = i;
    CHECK(hipMemcpy(ramp, h, sizeof(h), hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int mode = 0; mode < 2; mode++)
        for (int imm = 0; imm < 2; imm++)
            for (unsigned dst : {0u, 4096u, 61440u, 65536u, 69632u, 102400u, 131072u, 162816u}) {
                int f[4] = {-1, -1, -1, -1};
                CHECK(hipMemcpy(found, f, 16, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, ramp, dst, mode, imm, found);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(f, found, 16, hipMemcpyDeviceToHost));
                printf("%s imm=%d  M0=%6u: lane0 word at %6d, lane1 word at %6d, last word at %6d\n", mode ? "global_load_lds" : "buffer_load lds ", imm ? 1024 : 0, dst, f[0], f[1], f[2]);
            }
    return 0;
}
