// cond_producer.hip -- the conditioning of a WaveNet straight into the engine's fragment order.
//
// A WaveNet's conditioning is the output of its 1x1 `cond_layers` convolution over the upsampled features
// (pytorch/wavenet.py:190-202): Lh[n][l][b][c] = sum_k W[l*2R + c][k] x[b][k][n] + bias.  The generation kernels read it as
// [sample][layer][tile of 16 utterances][fragment][lane (g, j)][8 halves] with lane (g, j) holding positions 8g..8g+7 of the
// fragment's 32 channels for utterance j (nv_wavenet.py: cond_fragment_order gives position -> channel and the gate's
// pre-scale).  That is two 16x16 MFMA result tiles per fragment -- lane (g, j) of a result tile holds rows 4g..4g+3 of column j
// -- so the convolution can be computed in place: one wave takes a tile of 16 utterances and NB samples, keeps their feature
// columns as B operands in registers, streams the pre-arranged weight fragments (A operands, rows in position order) from L2 and
// writes every fragment with ONE 1-KiB store per wave: no [B][2R*L][N] intermediate, no permuting copy (torch ops: 48 GB of
// traffic per 256-sample chunk of 12 288 utterances; here the 16 GB of the result).
//   x      [tiles*16][num_samples][32*KF] fp16, channels last, zero-padded to whole k-fragments
//   wfrag  [L][NWF][2][KF][64 lanes][8] fp16: A fragments of row tile tt of fragment wf (row m = 4g+r <-> position (wf*4+g)*8 + tt*4 + r),
//          gate pre-scale and channel permutation folded in (nv_wavenet.py: cond_producer_weights)
//   bias   [L][NWF*32] fp32 in position order, pre-scaled
//   out    [num_samples][L][tiles][NWF][64 lanes][8] fp16  (a slice of the buffer handed to nvw_set_conditioning_packed_n)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nv_wavenet_c.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int NB = 8;        // samples per wave task
constexpr int KF_MAX = 4;    // up to 128 conditioning features

template <int KF>
__global__ __launch_bounds__(256) void cond_producer_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ wfrag,
                                                            const float* __restrict__ bias, _Float16* __restrict__ out, int tiles,
                                                            int num_samples, int num_layers, int nwf) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nblocks = (num_samples + NB - 1) / NB;
    if (task >= (long)tiles * nblocks) return;
    const int tile = (int)(task % tiles);
    const int n0 = (int)(task / tiles) * NB;
    const int KP = 32 * KF;

    // B operands: feature columns of this tile's 16 utterances for NB samples (lane (g, j): k = 32 kf + 8g .. +7 of utterance j)
    half8 bfr[NB][KF];
#pragma unroll
    for (int s = 0; s < NB; s++) {
        const int n = n0 + s < num_samples ? n0 + s : num_samples - 1;
        const _Float16* px = x + ((size_t)(tile * 16 + j) * num_samples + n) * KP + g * 8;
#pragma unroll
        for (int kf = 0; kf < KF; kf++) bfr[s][kf] = *(const half8*)(px + kf * 32);
    }
    const size_t sampleStride = (size_t)num_layers * tiles * nwf * 512;      // halves per sample of the packed buffer
    // weight fragments of (layer, fragment) it = l * nwf + wf: loaded one iteration ahead (L2 latency under the MFMAs and stores)
    const int iters = num_layers * nwf;
    half8 a[2][2][KF];
    auto load_a = [&](int buf, int it) {
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int kf = 0; kf < KF; kf++) a[buf][tt][kf] = *(const half8*)(wfrag + ((((size_t)it * 2 + tt) * KF + kf) * 64 + lane) * 8);
    };
    load_a(0, 0);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; half++) {           // (two iterations per trip: the buffer index stays a compile-time constant)
            const int cur = it + half;
            if (cur >= iters) break;
            if (cur + 1 < iters) load_a(half ^ 1, cur + 1);
            const float* pb = bias + (size_t)cur * 32 + g * 8;                 // positions (wf * 4 + g) * 8 .. + 7 of layer l
            const floatx4 b0 = *(const floatx4*)pb, b1 = *(const floatx4*)(pb + 4);
            const int l = cur / nwf, wf = cur - l * nwf;
            _Float16* po = out + ((size_t)l * tiles + tile) * nwf * 512 + (size_t)wf * 512 + lane * 8;
#pragma unroll
            for (int s = 0; s < NB; s++) {
                floatx4 c0 = b0, c1 = b1;
#pragma unroll
                for (int kf = 0; kf < KF; kf++) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[half][0][kf], bfr[s][kf], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[half][1][kf], bfr[s][kf], c1, 0, 0, 0);
                }
                half8 o;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    o[r] = (_Float16)c0[r];
                    o[4 + r] = (_Float16)c1[r];
                }
                if (n0 + s < num_samples) *(half8*)(po + (size_t)(n0 + s) * sampleStride) = o;
            }
        }
    }
}

}  // namespace

extern "C" int nvw_produce_conditioning_f16(const void* x, const void* wfrag, const float* bias, void* out, int tiles, int num_samples,
                                            int num_layers, int kfrags, int nwf, void* stream) {
    if (tiles <= 0 || num_samples <= 0 || num_layers <= 0 || nwf <= 0 || kfrags < 1 || kfrags > KF_MAX) return 0;
    const long tasks = (long)tiles * ((num_samples + NB - 1) / NB);
    const dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* px = (const _Float16*)x;
    const _Float16* pw = (const _Float16*)wfrag;
    _Float16* po = (_Float16*)out;
    switch (kfrags) {
        case 1: hipLaunchKernelGGL(cond_producer_kernel<1>, grid, block, 0, st, px, pw, bias, po, tiles, num_samples, num_layers, nwf); break;
        case 2: hipLaunchKernelGGL(cond_producer_kernel<2>, grid, block, 0, st, px, pw, bias, po, tiles, num_samples, num_layers, nwf); break;
        case 3: hipLaunchKernelGGL(cond_producer_kernel<3>, grid, block, 0, st, px, pw, bias, po, tiles, num_samples, num_layers, nwf); break;
        default: hipLaunchKernelGGL(cond_producer_kernel<4>, grid, block, 0, st, px, pw, bias, po, tiles, num_samples, num_layers, nwf); break;
    }
    return hipGetLastError() == hipSuccess ? 1 : 0;
}
