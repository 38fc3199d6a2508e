// wn_primitives.hip -- TEST-ONLY kernel entries (libwn_primitives.so, not part of the product ABI) that run the
// engine's device primitives in isolation, in the spirit of /root/reference/math_test.cu:262-410:
//   wnp_gemm          pack_weight_kernel -> per-wave fragment streams -> take()/gemm() over one tile of 16 columns
//                     (the MFMA fragment packing, the K permutation, the gated tile pairing, the prefetch ring)
//   wnp_softmax_pick  wn::softmax_pick (max / sum / scan / inverse-CDF pick over LPU lanes per utterance)
//   wnp_handoff       wn::send_tiles / recv_tiles_fast between two workgroups (tagged granules, both store scopes)
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../nv_wavenet_amd/csrc/wn_chain.hpp"
#include "../../nv_wavenet_amd/csrc/wn_kernels.hpp"

using namespace wn;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "wn_primitives: %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                 \
        }                                                                              \
    } while (0)

// out[M][16] = W (M x K) * X (K x 16): NW waves, wave w owns the tiles w, w+NW, ... (or the gated pairs)
template <bool F16, int M, int K, int NW, bool GATED>
__global__ __launch_bounds__(NW * 64) void gemm_kernel(const typename Prec<F16>::elem* wblob, const float* X, float* out) {
    using P = Prec<F16>;
    using frag = typename P::frag;
    constexpr int MT = M / 16 / NW;                 // tile slots per wave
    constexpr int KT = K / 16, KF = KT / P::TPF;
    constexpr int FW = MT * KF;                     // fragments per wave
    constexpr int PF = FW >= 6 ? 3 : 1;             // prefetch ring depth (divides nothing in particular: idx % PF)
    __shared__ __attribute__((aligned(16))) char xbuf[KF * 1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, j = lane & 15;
    // X -> B fragments through the same LDS exchange the engine uses: wave w puts the tiles k = w, w+NW, ...
    for (int t = w; t < KT; t += NW) {
        floatx4 v;
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = X[(t * 16 + g * 4 + r) * 16 + j];
        lds_put_tile<F16>(xbuf, t, lane, v);
    }
    __syncthreads();
    frag b[1][KF];
    lds_get_frags<F16, KF>(xbuf, lane, b[0]);
    const char* base = (const char*)wblob + (size_t)w * FW * 1024;
    const unsigned laneOff = lane * 16u;
    WStream<F16, PF> ws;
#pragma unroll
    for (int i = 0; i < PF; i++) ws.buf[i] = *(const frag*)(base + (size_t)i * 1024 + laneOff);
    floatx4 acc[1][MT];
#pragma unroll
    for (int i = 0; i < MT; i++) acc[0][i] = floatx4{0.f, 0.f, 0.f, 0.f};
    // the stream wraps onto itself at its end (the refills past the last fragment are never consumed)
    gemm<F16, PF, FW, 1, MT, KF>(ws, 0, base, base, laneOff, acc, b);
#pragma unroll
    for (int i = 0; i < MT; i++) {
        const int tile = GATED ? w + NW * (i >> 1) + (i & 1) * (M / 32) : w + NW * i;
#pragma unroll
        for (int r = 0; r < 4; r++) out[(tile * 16 + g * 4 + r) * 16 + j] = acc[0][i][r];
    }
}

template <bool F16, int M, int K, int NW, bool GATED>
static int run_gemm(const float* W, const float* X, float* out) {
    using elem = typename Prec<F16>::elem;
    float *dW, *dX, *dOut;
    elem* blob;
    CK(hipMalloc(&dW, sizeof(float) * M * K));
    CK(hipMalloc(&dX, sizeof(float) * K * 16));
    CK(hipMalloc(&dOut, sizeof(float) * M * 16));
    CK(hipMalloc(&blob, sizeof(elem) * M * K));
    CK(hipMemcpy(dW, W, sizeof(float) * M * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, X, sizeof(float) * K * 16, hipMemcpyHostToDevice));
    // (the gate pre-scaling of the fp16 engine is part of the packing: the test divides it out on the host)
    hipLaunchKernelGGL((pack_weight_kernel<F16>), dim3(64), dim3(256), 0, 0, blob, dW, M, K, NW, (size_t)(M / NW) * K,
                       GATED ? M / 32 : 0);
    hipLaunchKernelGGL((gemm_kernel<F16, M, K, NW, GATED>), dim3(1), dim3(NW * 64), 0, 0, blob, dX, dOut);
    CK(hipGetLastError());
    CK(hipMemcpy(out, dOut, sizeof(float) * M * 16, hipMemcpyDeviceToHost));
    CK(hipFree(dW));
    CK(hipFree(dX));
    CK(hipFree(dOut));
    CK(hipFree(blob));
    return 0;
}

template <int A, int NW>
__global__ __launch_bounds__(NW * 64) void softmax_kernel(const float* logits, const float* sel, int* picks, float* probs) {
    constexpr int LPU = 4 * NW, RPL = A / LPU, LROW = A + 4;
    __shared__ __attribute__((aligned(16))) float lg[16 * LROW];
    for (int i = threadIdx.x; i < 16 * A; i += NW * 64) lg[(i / A) * LROW + i % A] = logits[i];
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    const int su = tid / LPU, sq = tid % LPU;
    float e[RPL];
    float total;
    const int pick = softmax_pick<A, LPU, RPL>(lg + su * LROW + sq * RPL, sq, lane, sel[su], e, total);
    if (sq == 0) picks[su] = pick;
    for (int i = 0; i < RPL; i++) probs[su * A + sq * RPL + i] = e[i] / total;
}

template <int A, int NW> static int run_softmax(const float* logits, const float* sel, int* picks, float* probs) {
    float *dl, *ds, *dp;
    int* dk;
    CK(hipMalloc(&dl, sizeof(float) * 16 * A));
    CK(hipMalloc(&ds, sizeof(float) * 16));
    CK(hipMalloc(&dp, sizeof(float) * 16 * A));
    CK(hipMalloc(&dk, sizeof(int) * 16));
    CK(hipMemcpy(dl, logits, sizeof(float) * 16 * A, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, sel, sizeof(float) * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((softmax_kernel<A, NW>), dim3(1), dim3(NW * 64), 0, 0, dl, ds, dk, dp);
    CK(hipGetLastError());
    CK(hipMemcpy(picks, dk, sizeof(int) * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(probs, dp, sizeof(float) * 16 * A, hipMemcpyDeviceToHost));
    CK(hipFree(dl));
    CK(hipFree(ds));
    CK(hipFree(dp));
    CK(hipFree(dk));
    return 0;
}

// ---- hand-off: `pairs` producer workgroups each send `rounds` messages of NT*4 waves' tiles to their consumer
// workgroup, which echoes a checksum of every word back to the host; producers are delayed unevenly and the
// consumers pre-read the mailbox lines (L1-warm), as the guide asks of a hand-off test
template <int NT>
__global__ __launch_bounds__(256) void handoff_kernel(unsigned long long* mail, unsigned* status, int rounds, int forceAgent,
                                                      unsigned long long* sums, int* sameOut) {
    constexpr int NW = 4;
    const int pair = blockIdx.x >> 1, role = blockIdx.x & 1;       // role 0: producer, 1: consumer
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int MSG = NT * NW * 256, ACK = NW * 256;             // granules of a message / of its acknowledgement
    unsigned long long* place = mail;                               // [2 * pairs] placement words (padded to 64)
    unsigned long long* box = mail + 64 * ((gridDim.x + 63) / 64) + (size_t)pair * (MSG + ACK);
    gu32* st = (gu32*)status;
    bool same = false;
    if (!chain_place(place, blockIdx.x, blockIdx.x ^ 1, st, same)) return;
    if (role == 0 && threadIdx.x == 0) sameOut[pair] = same ? 1 : 0;
    if (forceAgent) same = false;
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; r++) {
        const unsigned tag = r + 1u;
        if (role == 0) {
            // uneven load: pair p's producer idles p*37 % 11 sleeps more in odd rounds
            for (int k = 0; k < ((pair * 37 + r * 5) % 11) * (r & 1 ? 8 : 1); k++) __builtin_amdgcn_s_sleep(8);
            floatx4 v[NT];
#pragma unroll
            for (int i = 0; i < NT; i++)
#pragma unroll
                for (int q = 0; q < 4; q++) v[i][q] = __uint_as_float(0x3f800000u ^ (unsigned)(((r * 131 + pair) * 64 + lane) * 16 + (w + NW * i) * 4 + q));
            send_tiles<NT, NW>(box, w, lane, tag, v, same);
            // wait for the consumer's acknowledgement of this round before overwriting the single slot
            floatx4 ack[1];
            if (!recv_tiles<1, NW>(box + MSG, w, lane, tag, ack, st, 0x700u)) return;
        } else {
            floatx4 v[NT];
            if (!recv_tiles_fast<NT, NW>(box, w, lane, tag, v, st, 0x600u)) return;
#pragma unroll
            for (int i = 0; i < NT; i++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned want = 0x3f800000u ^ (unsigned)(((r * 131 + pair) * 64 + lane) * 16 + (w + NW * i) * 4 + q);
                    acc += (__float_as_uint(v[i][q]) == want) ? 1ull : 0x100000000ull;
                }
            floatx4 ack[1] = {floatx4{1.f, 2.f, 3.f, 4.f}};
            send_tiles<1, NW>(box + MSG, w, lane, tag, ack, same);
        }
    }
    if (role == 1) atomicAdd(&sums[pair], acc);
}

extern "C" {

// precision 32|16; W col-major M x K (fp32, small integers are exact in both precisions); X [K][16]; out [M][16]
int wnp_gemm(int precision, int M, int K, int nw, int gated, const float* W, const float* X, float* out) {
#define CASE(F16, m, k, n, gt) \
    if ((precision == 16) == F16 && M == m && K == k && nw == n && (gated != 0) == gt) return run_gemm<F16, m, k, n, gt>(W, X, out);
#define BOTH(m, k, n, gt) CASE(false, m, k, n, gt) CASE(true, m, k, n, gt)
    BOTH(128, 64, 4, true)     // Wprev / Wcur at R = 64 (gated pairs)
    BOTH(64, 64, 4, false)     // Wres at R = 64
    BOTH(256, 64, 4, false)    // Wskip, S = 256, R = 64
    BOTH(128, 64, 4, false)    // Wskip, S = 128
    BOTH(256, 256, 4, false)   // Wzs (A x S) / Wza (A x A)
    BOTH(256, 128, 4, false)   // Wzs, S = 128;  Wskip at R = 128
    BOTH(256, 128, 4, true)    // Wprev / Wcur at R = 128
    BOTH(128, 128, 4, false)   // Wres at R = 128
    BOTH(64, 32, 2, true)      // R = 32: two waves
    BOTH(32, 32, 2, false)
    BOTH(128, 32, 2, false)
    BOTH(512, 256, 4, false)   // A = 512
    return -2;
}

int wnp_softmax_pick(int A, int nw, const float* logits, const float* sel, int* picks, float* probs) {
    if (A == 256 && nw == 4) return run_softmax<256, 4>(logits, sel, picks, probs);
    if (A == 256 && nw == 2) return run_softmax<256, 2>(logits, sel, picks, probs);
    if (A == 512 && nw == 4) return run_softmax<512, 4>(logits, sel, picks, probs);
    if (A == 1024 && nw == 4) return run_softmax<1024, 4>(logits, sel, picks, probs);
    return -2;
}

// returns 0 and fills good[pair] = words that arrived right, bad[pair] = words that did not, same[pair] = the pair
// found itself on one XCD; status != 0 = a hand-off timed out
int wnp_handoff(int pairs, int rounds, int force_agent_scope, long long* good, long long* bad, int* same, unsigned* status_out) {
    constexpr int NT = 2;
    unsigned long long *mail, *sums;
    unsigned* status;
    int* dsame;
    const size_t words = 64 * ((2 * pairs + 63) / 64) + (size_t)pairs * (NT * 4 * 256 + 4 * 256);
    CK(hipMalloc(&mail, words * 8));
    CK(hipMemset(mail, 0, words * 8));
    CK(hipMalloc(&sums, pairs * 8));
    CK(hipMemset(sums, 0, pairs * 8));
    CK(hipMalloc(&status, 16));
    CK(hipMemset(status, 0, 16));
    CK(hipMalloc(&dsame, pairs * 4));
    CK(hipMemset(dsame, 0, pairs * 4));
    hipLaunchKernelGGL((handoff_kernel<NT>), dim3(2 * pairs), dim3(256), 0, 0, mail, status, rounds, force_agent_scope, sums, dsame);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    for (int p = 0; p < pairs; p++) {
        unsigned long long s;
        CK(hipMemcpy(&s, sums + p, 8, hipMemcpyDeviceToHost));
        good[p] = (long long)(s & 0xffffffffull);
        bad[p] = (long long)(s >> 32);
    }
    CK(hipMemcpy(same, dsame, pairs * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(status_out, status, 4, hipMemcpyDeviceToHost));
    CK(hipFree(mail));
    CK(hipFree(sums));
    CK(hipFree(status));
    CK(hipFree(dsame));
    return 0;
}
}
