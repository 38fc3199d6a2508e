// wn_primitives.hip -- TEST-ONLY kernel entries (libwn_primitives.so, not part of the product ABI) that run the
// engine's device primitives in isolation, in the spirit of /root/reference/math_test.cu:262-410:
//   wnp_gemm          pack_weight_kernel -> per-wave fragment streams -> take()/gemm() over one tile of 16 columns
//                     (the MFMA fragment packing, the K permutation, the gated tile pairing, the prefetch ring)
//   wnp_softmax_pick  wn::softmax_pick (max / sum / scan / inverse-CDF pick over LPU lanes per utterance)
//   wnp_handoff       wn::send_tiles / recv_tiles_fast between two workgroups (tagged granules, both store scopes)
//   wnp_stream_walk   pack_layer_kernel + Cfg::streamPos -> the per-wave weight stream walked exactly like wavenet_wg walks
//                     it (buffer-resource ring: gemm_b / take_group / refill_group / skip_frags, the wrap behind the head)
//   wnp_gate          the fp16 engine's gate on pre-scaled pre-activations (gate1<true> and the staged gate_stage)
//   wnp_hog_*         a kernel that holds a number of CUs for a while (the chain's fault-tolerance test)
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
    if (!chain_place(place, blockIdx.x, blockIdx.x ^ 1, st, same, kChainTimeoutTicks)) return;
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
            if (!recv_tiles<1, NW>(box + MSG, w, lane, tag, ack, st, 0x700u, kChainTimeoutTicks)) return;
        } else {
            floatx4 v[NT];
            if (!recv_tiles_fast<NT, NW>(box, w, lane, tag, v, st, 0x600u, kChainTimeoutTicks)) return;
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

// ---- the weight stream of a whole model walked the way wavenet_wg walks it ---------------------------------------------
// One workgroup, one tile of 16 columns X.  Every wave goes through its stream in consumption order
//   cur(0) res(0) prev(1) | cur(1) skip(0) res(1) prev(2) | ... | cur(L-1) skip(L-2) res(L-1) prev(0) | skip(L-1) | zs pad za pad
// with the buffer-resource prefetch ring (take_group / refill_group inside gemm_b, skip_frags over the padding), `passes`
// times (the ring wraps behind the head onto layer 0), and writes W X of every matrix: out[pass][l] = prev 2R | cur 2R |
// res R | skip S rows of 16, then zs A | za A.  (prev(0) of a pass is the one taken at the end of that pass.)
template <bool F16, int R, int S, int A>
__global__ __launch_bounds__((Cfg<F16, R, S, A, 3>::THREADS)) void stream_walk_kernel(const void* wblob, const float* X, float* out, int L, int passes) {
    using C = Cfg<F16, R, S, A, 3>;            // (three tiles per workgroup: the configuration that streams the WHOLE head)
    using P = Prec<F16>;
    using frag = typename P::frag;
    constexpr int PF = C::PF, FLW = C::FLW, NW = C::NW, HTW = C::HTW, STW = C::STW, ATW = C::ATW, RT = C::RT;
    constexpr int KF_R = C::KF_R, KF_S = C::KF_S, KF_A = C::KF_A;
    static_assert(C::HS == C::FHW, "this walk expects the whole head in the stream");
    __shared__ __attribute__((aligned(16))) char xr[KF_R * 1024], xs[KF_S * 1024], xa[KF_A * 1024];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    auto put = [&](char* buf, int tiles) {
        for (int t = w; t < tiles; t += NW) {
            floatx4 v;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = X[(t * 16 + g * 4 + r) * 16 + j];
            lds_put_tile<F16>(buf, t, lane, v);
        }
    };
    put(xr, R / 16), put(xs, S / 16), put(xa, A / 16);
    __syncthreads();
    frag br[1][KF_R], bs[1][KF_S], ba[1][KF_A];
    lds_get_frags<F16, KF_R>(xr, lane, br[0]);
    lds_get_frags<F16, KF_S>(xs, lane, bs[0]);
    lds_get_frags<F16, KF_A>(xa, lane, ba[0]);
    const unsigned laneOff = lane * 16u;
    const char* wbase = (const char*)wblob + (size_t)w * C::waveStreamFrags(L) * 1024;
    const rsrc_t rs = make_rsrc(wbase);
    WStream<F16, PF, F16> ws;
#pragma unroll
    for (int i = 0; i < PF; i++) ws.buf[i] = buf_load<frag>(rs, laneOff + (unsigned)(i & 3) * 1024u, (unsigned)(i & ~3) * 1024u);
    const size_t perLayer = (size_t)(5 * R + S) * 16, perPass = (size_t)L * perLayer + (size_t)2 * A * 16;
    auto zero = [](auto& acc) {
        for (auto& t : acc[0]) t = floatx4{0.f, 0.f, 0.f, 0.f};
    };
    auto storeGate = [&](float* dst, const floatx4 (&acc)[1][2 * HTW]) {     // gated pairs: slot 2i = tile w + NW i, slot 2i+1 = + RT
#pragma unroll
        for (int i = 0; i < 2 * HTW; i++) {
            const int tile = w + NW * (i >> 1) + (i & 1) * RT;
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(tile * 16 + g * 4 + r) * 16 + j] = acc[0][i][r];
        }
    };
    auto storePlain = [&](float* dst, const auto& acc, int n) {
        for (int i = 0; i < n; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) dst[((w + NW * i) * 16 + g * 4 + r) * 16 + j] = acc[0][i][r];
    };
    for (int pass = 0; pass < passes; pass++) {
        float* po = out + (size_t)pass * perPass;
        for (int l = 0; l < L; l++) {
            // (exactly wavenet_wg's position arithmetic: layer 0 counts from the start of the stream)
            const int wl = l ? (l - 1) * FLW : 0;
            floatx4 cur[1][2 * HTW], res[1][HTW], prv[1][2 * HTW];
            zero(cur), zero(res), zero(prv);
            if (l) gemm_b<F16, PF, 0, 1, 2 * HTW, KF_R>(ws, rs, C::P_CUR, wl, 0, laneOff, cur, br);
            else gemm_b<F16, PF, 0, 1, 2 * HTW, KF_R>(ws, rs, C::P_CUR0 - FLW, wl, 0, laneOff, cur, br);
            storeGate(po + l * perLayer + 2 * R * 16, cur);
            if (l) {
                floatx4 sk[1][STW];
                zero(sk);
                gemm_b<F16, PF, 0, 1, STW, KF_R>(ws, rs, C::P_SKIP, wl, 0, laneOff, sk, br);
                storePlain(po + (l - 1) * perLayer + 5 * R * 16, sk, STW);
            }
            if (l) {
                gemm_b<F16, PF, 0, 1, HTW, KF_R>(ws, rs, C::P_RES, wl, 0, laneOff, res, br);
                gemm_b<F16, PF, 0, 1, 2 * HTW, KF_R>(ws, rs, C::P_PREV, wl, 0, laneOff, prv, br);
            } else {
                gemm_b<F16, PF, 0, 1, HTW, KF_R>(ws, rs, C::P_RES - FLW, wl, 0, laneOff, res, br);
                gemm_b<F16, PF, 0, 1, 2 * HTW, KF_R>(ws, rs, C::P_PREV - FLW, wl, 0, laneOff, prv, br);
            }
            storePlain(po + l * perLayer + 4 * R * 16, res, HTW);
            storeGate(po + (l + 1 < L ? l + 1 : 0) * perLayer, prv);
        }
        {
            floatx4 sk[1][STW];
            zero(sk);
            gemm_b<F16, PF, 0, 1, STW, KF_R>(ws, rs, C::P_CUR, (L - 1) * FLW, 0, laneOff, sk, br);
            storePlain(po + (L - 1) * perLayer + 5 * R * 16, sk, STW);
        }
        floatx4 zs[1][ATW], za[1][ATW];
        zero(zs), zero(za);
        gemm_b<F16, PF, C::HSP, 1, ATW, KF_S>(ws, rs, C::O_ZS, L * FLW, 0, laneOff, zs, bs);
        skip_frags<F16, PF, C::HSP, F16, C::PAD1>(ws, rs, C::FW_ZS, L * FLW, 0, laneOff);
        gemm_b<F16, PF, C::HSP, 1, ATW, KF_A>(ws, rs, C::O_ZA, L * FLW, 0, laneOff, za, ba);
        skip_frags<F16, PF, C::HSP, F16, C::PAD2>(ws, rs, C::O_ZA + C::FW_ZA, L * FLW, 0, laneOff);
        storePlain(po + L * perLayer, zs, ATW);
        storePlain(po + L * perLayer + A * 16, za, ATW);
    }
}

// W: per layer [Wprev 2RxR | Wcur 2RxR | Wres RxR | Wskip SxR] col-major, then Wzs AxS, Wza AxA; biases are not under test
template <bool F16, int R, int S, int A>
static int run_stream_walk(int L, int passes, const float* W, const float* X, float* out) {
    using C = Cfg<F16, R, S, A, 3>;
    using elem = typename Prec<F16>::elem;
    const size_t perLayerW = (size_t)5 * R * R + (size_t)S * R, nW = L * perLayerW + (size_t)A * S + (size_t)A * A;
    const size_t perPass = (size_t)L * (5 * R + S) * 16 + (size_t)2 * A * 16;
    const int maxK = R > S ? (R > A ? R : A) : (S > A ? S : A);
    float *dW, *dX, *dOut, *dBias;
    elem* blob;
    const size_t blobElems = (size_t)C::NW * C::waveStreamFrags(L) * C::FRAG_ELEMS;
    CK(hipMalloc(&dW, sizeof(float) * nW));
    CK(hipMalloc(&dX, sizeof(float) * maxK * 16));
    CK(hipMalloc(&dOut, sizeof(float) * perPass * passes));
    CK(hipMalloc(&dBias, sizeof(float) * (size_t)L * C::BIAS_L));
    CK(hipMalloc(&blob, sizeof(elem) * blobElems));
    CK(hipMemset(blob, 0, sizeof(elem) * blobElems));
    CK(hipMemset(dBias, 0, sizeof(float) * (size_t)L * C::BIAS_L));
    CK(hipMemset(dOut, 0xff, sizeof(float) * perPass * passes));
    CK(hipMemcpy(dW, W, sizeof(float) * nW, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, X, sizeof(float) * maxK * 16, hipMemcpyHostToDevice));
    for (int l = 0; l < L; l++) {           // exactly nvWavenetInfer::setLayerWeights
        LayerSrc src;
        const float* wl = dW + l * perLayerW;
        src.Wprev = wl, src.Wcur = wl + 2 * R * R, src.Wres = wl + 4 * R * R, src.Wskip = wl + 5 * R * R;
        src.Bh = dBias, src.Bres = dBias, src.Bskip = dBias;
        hipLaunchKernelGGL((pack_layer_kernel<F16>), dim3(64), dim3(256), 0, 0, blob, dBias + (size_t)l * C::BIAS_L, src, R, S, C::NW,
                           C::waveStreamFrags(L) * C::FRAG_ELEMS, (int)C::streamPos(l, C::O_PREV, L), (int)C::streamPos(l, C::O_CUR, L),
                           (int)C::streamPos(l, C::O_RES, L), (int)C::streamPos(l, C::O_SKIP, L));
    }
    const size_t hf = C::headOffsetFrags(L);   // ... and setOutWeights
    hipLaunchKernelGGL((pack_weight_kernel<F16>), dim3(64), dim3(256), 0, 0, blob + (hf + C::O_ZS) * C::FRAG_ELEMS, dW + L * perLayerW, A, S, C::NW,
                       C::waveStreamFrags(L) * C::FRAG_ELEMS, 0);
    hipLaunchKernelGGL((pack_weight_kernel<F16>), dim3(64), dim3(256), 0, 0, blob + (hf + C::O_ZA) * C::FRAG_ELEMS,
                       dW + L * perLayerW + (size_t)A * S, A, A, C::NW, C::waveStreamFrags(L) * C::FRAG_ELEMS, 0);
    hipLaunchKernelGGL((stream_walk_kernel<F16, R, S, A>), dim3(1), dim3(C::THREADS), 0, 0, (const void*)blob, dX, dOut, L, passes);
    CK(hipGetLastError());
    CK(hipMemcpy(out, dOut, sizeof(float) * perPass * passes, hipMemcpyDeviceToHost));
    CK(hipFree(dW));
    CK(hipFree(dX));
    CK(hipFree(dOut));
    CK(hipFree(dBias));
    CK(hipFree(blob));
    return 0;
}

// h = tanh(a) sigmoid(b) of the fp16 engine: pre-activations arrive pre-scaled (2 log2 e, -log2 e); way 0 = gate1<true>, way 1 =
// the five-stage form wavenet_wg interleaves with MFMAs
__global__ void gate_kernel(const float* a, const float* b, float* h, int n, int way) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a0 = a[2 * i] * gate_prescale<true>(false), a1 = a[2 * i + 1] * gate_prescale<true>(false);
    const float b0 = b[2 * i] * gate_prescale<true>(true), b1 = b[2 * i + 1] * gate_prescale<true>(true);
    if (way == 0) {
        h[2 * i] = gate1<true>(a0, b0);
        h[2 * i + 1] = gate1<true>(a1, b1);
    } else {
        floatx2 ea, eb, ra, rb, hp;
        gate_stage<true, 0>(a0, a1, b0, b1, ea, eb, ra, rb, hp);
        gate_stage<true, 1>(a0, a1, b0, b1, ea, eb, ra, rb, hp);
        gate_stage<true, 2>(a0, a1, b0, b1, ea, eb, ra, rb, hp);
        gate_stage<true, 3>(a0, a1, b0, b1, ea, eb, ra, rb, hp);
        gate_stage<true, 4>(a0, a1, b0, b1, ea, eb, ra, rb, hp);
        h[2 * i] = hp[0];
        h[2 * i + 1] = hp[1];
    }
}

// holds `blockIdx` CUs (100 KiB of LDS per workgroup: one per CU, and nothing with a large LDS footprint fits beside it)
// until the wall clock passes `until`
__global__ __launch_bounds__(64) void hog_kernel(long long ticks, unsigned* sink) {
    extern __shared__ char hogLds[];
    const long long t0 = (long long)wall_clock64();
    unsigned n = 0;
    while ((long long)wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        n++;
    }
    if (threadIdx.x == 0) {
        hogLds[0] = (char)n;
        sink[blockIdx.x] = n + (unsigned)hogLds[0];
    }
}
static hipStream_t g_hogStream = nullptr;
static unsigned* g_hogSink = nullptr;

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

// the whole weight stream of an L-layer model; see stream_walk_kernel for the layouts
int wnp_stream_walk(int precision, int R, int S, int A, int L, int passes, const float* W, const float* X, float* out) {
#define WALK(F16, r, s, a) \
    if ((precision == 16) == F16 && R == r && S == s && A == a) return run_stream_walk<F16, r, s, a>(L, passes, W, X, out);
    WALK(true, 64, 256, 256)
    WALK(false, 64, 256, 256)
    WALK(true, 64, 128, 256)
    WALK(true, 128, 256, 256)
    WALK(false, 32, 128, 256)
    WALK(true, 32, 256, 256)       // FLW = 13 fragments per layer and wave: only a ring of depth 1 divides it
    return -2;
}

int wnp_gate(int n, int way, const float* a, const float* b, float* h) {
    float *da, *db, *dh;
    CK(hipMalloc(&da, sizeof(float) * n));
    CK(hipMalloc(&db, sizeof(float) * n));
    CK(hipMalloc(&dh, sizeof(float) * n));
    CK(hipMemcpy(da, a, sizeof(float) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b, sizeof(float) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gate_kernel, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, da, db, dh, n, way);
    CK(hipGetLastError());
    CK(hipMemcpy(h, dh, sizeof(float) * n, hipMemcpyDeviceToHost));
    CK(hipFree(da));
    CK(hipFree(db));
    CK(hipFree(dh));
    return 0;
}

// asynchronously: `cus` workgroups of one wave and 100 KiB of LDS each spin for `ms` milliseconds on a stream of their own
int wnp_hog_start(int cus, double ms) {
    if (!g_hogStream) {
        CK(hipStreamCreateWithFlags(&g_hogStream, hipStreamNonBlocking));
        CK(hipMalloc(&g_hogSink, 4096 * sizeof(unsigned)));
        CK(hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    }
    hipLaunchKernelGGL(hog_kernel, dim3(cus), dim3(64), 100 * 1024, g_hogStream, (long long)(ms * 1e5), g_hogSink);
    CK(hipGetLastError());
    return 0;
}
int wnp_hog_wait(void) {
    if (g_hogStream) CK(hipStreamSynchronize(g_hogStream));
    return 0;
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
