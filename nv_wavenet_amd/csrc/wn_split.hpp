// wn_split.hpp -- wn::wavenet_split: the single-workgroup organisation with its work split by ROLE over two
// waves per SIMD (round 3).
//
// wn::wavenet_wg (wn_kernels.hpp) runs one wave per SIMD, and that lone wave is what bounds it: it issues one
// instruction every ~5.5 clk whatever its kind (its MFMAs do not overlap its own VALU work unless interleaved by hand), and
// its vector-memory queue returns in order, so the weight refills (L2 hits) wait behind the HBM loads of taps and conditioning.
// MI355X_MICROARCH.md ("Two waves per SIMD"): the matrix pipe and the VALU of a SIMD are separate, and two co-resident
// waves use both at the same time -- if their work is complementary.  Splitting the utterance tiles over two such waves
// would stream the weights twice (the CU's 58 B/clk L1 path does not allow it); splitting the WORK does not:
//
//   role A (waves 0-3, one per SIMD, raised priority): the dependent chain of a layer --
//       Wcur x (on top of the pre-activation role B prepared) -> gate -> h exchange -> Wres h -> x exchange --
//       and, in its spare registers, the staging of the launch's HBM reads (dilated taps and conditioning: requested four
//       layers ahead, written to LDS two layers ahead of their use).  VALU-heavy (the gate), few MFMAs, a short stream.
//   role B (waves 4-7, the other wave of each SIMD): everything that does not depend on the current x --
//       bias + conditioning + dilated-tap GEMM of the NEXT layer (handed to role A through LDS as fp32 accumulator
//       tiles, which role A reads where wavenet_wg read the bias), the skip GEMM of the PREVIOUS layer, the ring
//       stores.  MFMA-heavy, almost no VALU; its load queue carries nothing but the weight stream.
//   head (skip ReLU -> Zs -> Za -> softmax -> pick): all eight waves, output rows split eight ways.
//
// The weight streams are per wave and per role (role A: cur | res per layer; role B: prev | skip), so the model is still
// streamed exactly once per sample and workgroup.  Arithmetic, operand rounding and summation order are those of
// wavenet_wg (bias, conditioning, dilated tap, current tap; skip sums in layer order), so the samples are bit-identical
// to it -- which is how it is tested.  fp16 engine, R = 64, even layer counts, production launches (no activation dump).
//
// Barrier discipline: every s_barrier is executed by all eight waves, in the same order:
//   prologue: P1 P2;   per sample: E, (H(l) X(l)) x L, S, Z, Z2, G, Y.
// LDS hand-offs (who writes between which barriers / who reads); h, taps and conditioning double-buffered by layer parity:
//   xbuf            A writes x_{l+1} in (H(l),X(l)) [and x_0 before E]; A and B read in (X(l),H(l+1))
//   hbuf[l&1]       A writes h_l in (X(l-1),H(l));    A reads in (H(l),X(l)), B (skip GEMM of layer l) in (X(l),X(l+1))
//   accbuf          B writes pre-act(l+1) in (H(l),X(l));              A reads in (X(l),H(l+1))
//   xpbuf[l&1], condbuf[l&1]   A writes tap / conditioning of layer l in (X(l-3),H(l-2)); B reads in (X(l-2),H(l-1))
#pragma once

#include "wn_split_cfg.hpp"

namespace wn {

// acc[bt][mt] += W(tile mt) * b[bt], fragment order of gemm_b, with the B fragments read from their LDS image KC k steps at a
// time (between gemm_b: all of them in registers, and gemm_ldsb_b: one k step at a time, an LDS round trip per step)
template <int PF, int WRAP, int BT, int MT, int KF, int KC, bool PIN>
WN_DEV void gemm_ldsc_b(WStream<true, PF, PIN>& ws, rsrc_t rs, int pos0, int basePos, int wrapPos, unsigned laneOff,
                        floatx4 (&acc)[BT][MT], const char* bimg, int lane) {
    using frag = typename Prec<true>::frag;
    constexpr int G = MT >= 4 ? 4 : MT;
    static_assert(KF % KC == 0, "k steps per chunk");
#pragma unroll
    for (int mg = 0; mg < MT / G; mg++) {
#pragma unroll
        for (int kc = 0; kc < KF / KC; kc++) {
            frag b[BT][KC];
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int kk = 0; kk < KC; kk++) b[bt][kk] = *(const frag*)(bimg + (((bt * KF + kc * KC + kk) * 64 + lane) << 4));
#pragma unroll
            for (int kk = 0; kk < KC; kk++) {
#pragma unroll
                for (int mi = 0; mi < G; mi++) {
                    const int idx = pos0 + (mg * KF + kc * KC + kk) * G + mi;
                    frag a[1];
                    take_group<true, PF, PIN, 1>(ws, idx, a);
#pragma unroll
                    for (int bt = 0; bt < BT; bt++) acc[bt][mg * G + mi] = mma(a[0], b[bt][kk], acc[bt][mg * G + mi]);
                    refill_group<true, PF, WRAP, PIN, 1>(ws, rs, idx, basePos, wrapPos, laneOff);
                }
            }
        }
    }
}

WN_DEV const char* uniform_ptr(const char* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// EMBLDS: the current tap's embedding table lives in LDS.  RAW: see wavenet_wg.
template <int R, int S, int A, int BT, bool EMBLDS, int RAW>
__global__ __launch_bounds__(512) void wavenet_split(const Params p) {
    using SC = SCfg<R, S, A, BT>;
    using C = typename SC::C;
    using P = Prec<true>;
    using frag = typename P::frag;
    using quad = typename P::quad;
    using elem = typename P::elem;
    constexpr int RT = SC::RT, HTW = SC::HTW, STW = SC::STW, ATW8 = SC::ATW8;
    constexpr int KF_R = SC::KF_R, KF_S = SC::KF_S, KF_A = SC::KF_A;
    constexpr int PFA = SC::PFA, PFB = SC::PFB, FLA = SC::FLA, FLB = SC::FLB, PHB = SC::PHB;
    constexpr int HEADP = SC::HEADP, NPASS = SC::NPASS;
    constexpr int RPL = A / 16;
    static_assert(SC::SUPPORTED, "wavenet_split: fp16, R = 64");
    // The weight rings of this kernel are small (one layer of a role) and live in architectural VGPRs: a kernel that touches
    // the accumulator file at all gets its 256 registers split 128 + 128 by the compiler (two waves per SIMD), one that does
    // not gets all 256 as VGPRs.
    constexpr bool SPIN = false;
#ifdef WN_SPLIT_TIMING
    // experiment build only: per-phase shader-clock sums of wave 0 (role A) and wave 4 (role B) of workgroup 0 -> p.p[0..31]
    unsigned long long tacc[16] = {0};
    unsigned long long tmark = __builtin_amdgcn_s_memtime();
#define WN_STM(i)                                                          \
    {                                                                      \
        unsigned long long _n = __builtin_amdgcn_s_memtime();              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 \
        tacc[i] += _n - tmark;                                             \
        tmark = _n;                                                        \
    }
#else
#define WN_STM(i)
#endif
    static_assert(RAW == 0 || RAW == 1 || RAW == 2, "RAW");

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const xbuf = lds + SC::OFF_X;
    char* const hbuf = lds + SC::OFF_H;
    char* const xpbuf = lds + SC::OFF_XP;
    char* const accbuf = lds + SC::OFF_ACC;
    char* const condbuf = lds + SC::OFF_COND;
    char* const skbuf = lds + SC::OFF_SK;
    char* const zsbuf = lds + SC::OFF_ZS;
    float* const lgbuf = (float*)(lds + SC::OFF_LG);
    int* const ybuf = (int*)(lds + SC::OFF_Y);
    float* const biasLds = (float*)(lds + SC::LDS_FIXED);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool roleB = w8 >= 4;
    const int w = w8 & 3;
    const int g = lane >> 4, j = lane & 15;
    const int L = p.numLayers;
    const int tile0 = p.tileBase + blockIdx.x * BT;

    int ub[BT];
#pragma unroll
    for (int bt = 0; bt < BT; bt++) {
        const int b = (tile0 + bt) * 16 + j;
        ub[bt] = b < p.batch ? b : p.batch - 1;
    }
    const int su = tid >> 4, sq = tid & 15;        // softmax role: utterance su (+32 per pass), lane sq of its 16

    // ---- bias table -> LDS: gate and residual biases per layer, the skip biases summed in layer order (added once, at the
    //      head, like wavenet_wg does), head biases ----
    float* const skipSum = biasLds + L * SC::BIAS_L;
    float* const headBias = skipSum + S;
    for (int i = tid; i < L * SC::BIAS_L; i += 512) {
        const int l = i / SC::BIAS_L, c = i % SC::BIAS_L;
        biasLds[i] = p.bias[(size_t)l * C::BIAS_L + c];
    }
    for (int s0 = tid; s0 < S; s0 += 512) {
        float run = p.bias[3 * R + s0];
        for (int l = 1; l < L; l++) run += p.bias[(size_t)l * C::BIAS_L + 3 * R + s0];
        skipSum[s0] = run;
    }
    for (int i = tid; i < 2 * A; i += 512) headBias[i] = p.bias[(size_t)L * C::BIAS_L + i];
    const elem* embPrev = (const elem*)p.embPrev;
    const elem* embCur = (const elem*)p.embCur;
    if constexpr (EMBLDS) {
        elem* const embLds = (elem*)(biasLds + SC::biasFloats(L));
        const floatx4* s0 = (const floatx4*)p.embCur;
        constexpr int CH = (int)(A * R * sizeof(elem) / 16);
        for (int i = tid; i < CH; i += 512) ((floatx4*)embLds)[i] = s0[i];
        embCur = embLds;
    }
    __syncthreads();

    const unsigned laneOff = (unsigned)lane * 16u;
    // (uniform_ptr: wave-uniform addresses pinned to SGPRs -- where the compiler shares a sub-expression with code under a
    //  lane mask it computes them with the VALU, and every buffer instruction through such a resource becomes a waterfall loop)
    const char* const wbase = uniform_ptr((const char*)p.wsplit + (size_t)w8 * SC::waveStrideFrags(L) * 1024);
    const rsrc_t rsW = make_rsrc(wbase);

    // selectors of the utterances this lane serves in the softmax passes
    auto load_selectors = [&](int t, float (&selv)[NPASS]) {
        if (p.useRng) return;
#pragma unroll
        for (int pp = 0; pp < NPASS; pp++) {
            int sb = tile0 * 16 + pp * 32 + su;
            sb = sb < p.batch ? sb : p.batch - 1;
            selv[pp] = p.sel[(size_t)t * p.maxBatch + sb];
        }
    };
    auto draw_selectors = [&](int t, float (&selv)[NPASS]) {
        if (!p.useRng) return;
        // lane q < NPASS of every 16-lane row draws the selector of pass q, the row takes it over with a row broadcast
        const int q = sq < NPASS ? sq : 0;
        int sb = tile0 * 16 + q * 32 + su;
        sb = sb < p.batch ? sb : p.batch - 1;
        const float mine = philox_selector(p.rngKey0, p.rngKey1, (unsigned)t, (unsigned)sb);
        selv[0] = dpp_f<0x150>(mine);
        if constexpr (NPASS > 1) selv[1] = dpp_f<0x151>(mine);
        if constexpr (NPASS > 2) selv[2] = dpp_f<0x152>(mine);
    };

    // ---- the head from the zs GEMM on: same code for both roles, each wave with its own ring and stream ----
    auto head = [&](auto& ws, auto PFc, const int headBase, const float (&selv)[NPASS], const int t) {
        constexpr int PF = decltype(PFc)::value;
        floatx4 zs[BT][ATW8];
        constexpr int KCH = BT >= 3 ? 4 : 8;       // k steps of B fragments in registers at a time (head GEMMs)
        {
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < ATW8; i++) zs[bt][i] = *(const floatx4*)(headBias + (w8 + 8 * i) * 16 + g * 4);
            gemm_ldsc_b<PF, HEADP, BT, ATW8, KF_S, (KF_S % KCH == 0 ? KCH : KF_S), SPIN>(ws, rsW, SC::O_ZS, headBase, 0, laneOff, zs, skbuf, lane);
            skip_frags<true, PF, HEADP, SPIN, SC::ZSP - SC::FW_ZS>(ws, rsW, SC::O_ZS + SC::FW_ZS, headBase, 0, laneOff);
        }
#pragma unroll
        for (int bt = 0; bt < BT; bt++)
#pragma unroll
            for (int i = 0; i < ATW8; i++) {
#pragma unroll
                for (int r = 0; r < 4; r++) zs[bt][i][r] = __builtin_fmaxf(zs[bt][i][r], 0.f);
                lds_put_tile<true>(zsbuf + bt * KF_A * 1024, w8 + 8 * i, lane, zs[bt][i]);
            }
        wg_barrier();   // Z
        floatx4 za[BT][ATW8];
        {
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < ATW8; i++) za[bt][i] = *(const floatx4*)(headBias + A + (w8 + 8 * i) * 16 + g * 4);
            gemm_ldsc_b<PF, HEADP, BT, ATW8, KF_A, (KF_A % KCH == 0 ? KCH : KF_A), SPIN>(ws, rsW, SC::O_ZA, headBase, 0, laneOff, za, zsbuf, lane);
            skip_frags<true, PF, HEADP, SPIN, SC::ZAP - SC::FW_ZA>(ws, rsW, SC::O_ZA + SC::FW_ZA, headBase, 0, laneOff);
        }
        wg_barrier();   // Z2: every wave is done with the zs image (the logits take its place)
#pragma unroll
        for (int bt = 0; bt < BT; bt++)
#pragma unroll
            for (int i = 0; i < ATW8; i++)
                *(floatx4*)(lgbuf + (bt * 16 + j) * SC::LROW + (w8 + 8 * i) * 16 + g * 4) = za[bt][i];
        wg_barrier();   // G
#pragma unroll
        for (int pp = 0; pp < NPASS; pp++) {
            const int u = pp * 32 + su;
            const bool uval = u < BT * 16;
            const int uc = uval ? u : BT * 16 - 1;
            float e[RPL];
            float total;
            const float* lrow = lgbuf + uc * SC::LROW + sq * RPL;
            const int pick = softmax_pick<A, 16, RPL>(lrow, sq, lane, selv[pp], e, total);
            const int sb = tile0 * 16 + u;
            if (sq == 0 && uval) {
                ybuf[u] = pick;
                if (sb < p.batch) p.yOut[(size_t)sb * p.numSamples + t] = pick;
            }
        }
        wg_barrier();   // Y
    };

    const int tEnd = p.initSample + p.count;
    constexpr int CR = RAW == 1 ? 2 * C::COND_FR : C::COND_FR;     // conditioning registers per tile (raw fp32: two quads per fragment)
    static_assert(RAW != 1, "wavenet_split reads packed or fp16 conditioning");
    constexpr int TAPN = (BT * KF_R + 3) / 4;                       // dilated-tap fragments staged per role A wave

    if (!roleB) {
        // =====================================================================================================
        // role A: the dependent chain + the HBM traffic of the launch (taps and conditioning staged through its spare registers)
        // =====================================================================================================
        __builtin_amdgcn_s_setprio(2);
        WStream<true, PFA, SPIN> ws;
#pragma unroll
        for (int i = 0; i < PFA; i++) ws.buf[i] = buf_load<frag, WN_W_AUX>(rsW, laneOff + (unsigned)(i & 3) * 1024u, (unsigned)(i & ~3) * 1024u);

        // ---- staging of dilated taps and conditioning: HBM -> registers (requested four layers ahead) -> LDS (two layers
        //      ahead of role B's use).  Two register sets by layer parity: the set of layer n is written to LDS during layer
        //      n-2 and re-requested for layer n+2 right behind.  Loads return in order per wave: the refills of this role's
        //      weight ring wait behind them, which is why that ring is two layers deep (PFA = 2 FLA).
        frag tpE[TAPN], tpO[TAPN];
        frag cdE[BT][CR], cdO[BT][CR];
        const size_t condStride = (size_t)__builtin_amdgcn_readfirstlane(p.tiles * 4 * C::COND_FR * 1024);
        const char* condNext = uniform_ptr((const char*)p.cond + ((size_t)tile0 * 4 + w) * C::COND_FR * 1024 +
                                           (size_t)p.initSample * L * condStride);
        const size_t ringTile = (size_t)__builtin_amdgcn_readfirstlane(p.ringSlots * KF_R * 1024);
        const rsrc_t rsRing = make_rsrc(uniform_ptr((const char*)p.ring + (size_t)tile0 * ringTile));
        constexpr unsigned RAWE = 2u;
        const size_t rawRow = (size_t)__builtin_amdgcn_readfirstlane((int)(p.maxBatch * (2 * R) * RAWE));
        unsigned rawOff[BT];
#pragma unroll
        for (int bt = 0; bt < BT; bt++) rawOff[bt] = ((unsigned)ub[bt] * (unsigned)(2 * R) + (unsigned)g * 4u) * RAWE;
        // tap fragments (tile bt, k) of a layer, numbered q = bt * KF_R + k, are spread over the four waves: wave w stages
        // q = w, w + 4, ... (modulo their number: the last waves stage a fragment twice -- same data to the same place --
        // so that every wave issues the same instructions and the compiler's load counting stays exact)
        unsigned tapSrc[TAPN], tapDst[TAPN];
#pragma unroll
        for (int i = 0; i < TAPN; i++) {
            const int q = (w + 4 * i) % (BT * KF_R);
            const int bt = q / KF_R, k = q % KF_R;
            tapSrc[i] = (unsigned)bt * (unsigned)ringTile + (unsigned)k * 1024u;
            tapDst[i] = (unsigned)((bt * KF_R + k) * 1024);
        }
        auto request = [&](int tn, int ln, const Dil dl, frag (&tp)[TAPN], frag (&cdd)[BT][CR]) {
            if (ln >= L) { ln -= L; tn += 1; }
            const unsigned rp0 = (unsigned)(dl.off + (tn & (dl.d - 1))) * (unsigned)(KF_R * 1024);
#pragma unroll
            for (int i = 0; i < TAPN; i++) tp[i] = buf_load<frag, 2>(rsRing, laneOff, rp0 + tapSrc[i]);
            if constexpr (RAW == 2) {
                const int tc = tn < p.condSamples ? tn : p.condSamples - 1;
                const rsrc_t rsRaw = make_rsrc(uniform_ptr((const char*)p.condRaw + ((size_t)tc * L + ln) * rawRow));
                auto slotOff = [&](int it) { return (unsigned)((w + 4 * (it >> 1) + (it & 1) * RT) * 16) * RAWE; };
#pragma unroll
                for (int bt = 0; bt < BT; bt++)
#pragma unroll
                    for (int k = 0; k < HTW; k++) {
                        const uintx2 qa = __builtin_amdgcn_raw_buffer_load_b64(rsRaw, rawOff[bt], slotOff(2 * k), WN_RAW_AUX);
                        const uintx2 qb = __builtin_amdgcn_raw_buffer_load_b64(rsRaw, rawOff[bt], slotOff(2 * k + 1), WN_RAW_AUX);
                        cdd[bt][k] = __builtin_bit_cast(frag, uintx4{qa[0], qa[1], qb[0], qb[1]});
                    }
            } else {
                const rsrc_t rsCond = make_rsrc(condNext);
                condNext += condStride;
#pragma unroll
                for (int bt = 0; bt < BT; bt++)
#pragma unroll
                    for (int k = 0; k < C::COND_FR; k++)
                        cdd[bt][k] = buf_load<frag, 2>(rsCond, laneOff + (unsigned)(k & 3) * 1024u,
                                                       (unsigned)((bt * 4 * C::COND_FR + (k & ~3)) * 1024));
            }
        };
        // registers -> the LDS slot of the layer's parity (zero taps before the start, t < d: nv_wavenet_reference.cpp:287)
        auto stage = [&](const frag (&tp)[TAPN], const frag (&cdd)[BT][CR], const bool have, const int par) {
            char* const tdst = xpbuf + par * SC::XPBUF1 + lane * 16;
            frag z;
#pragma unroll
            for (int e = 0; e < P::EPL; e++) z[e] = (elem)0.f;
#pragma unroll
            for (int i = 0; i < TAPN; i++) *(frag*)(tdst + tapDst[i]) = have ? tp[i] : z;
            char* const cdst = condbuf + par * SC::CONDBUF1 + lane * 16;
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int k = 0; k < CR; k++) *(frag*)(cdst + ((bt * 4 + w) * CR + k) * 1024) = cdd[bt][k];
        };
        auto have_tap = [&](int t, int l) { return (l < L ? t : t + 1) >= p.dil[l].d; };

        int yPrev[BT], yCur[BT];
        floatx4 ep[BT][HTW];
#pragma unroll
        for (int bt = 0; bt < BT; bt++) {
            yPrev[bt] = p.yInPrev[ub[bt]];
            yCur[bt] = p.yInCur[ub[bt]];
#pragma unroll
            for (int i = 0; i < HTW; i++)
                ep[bt][i] = quad_to_f32(*(const quad*)(embPrev + (size_t)yPrev[bt] * R + (w + 4 * i) * 16 + g * 4));
        }
        // prologue: layers 0 and 1 of the first sample staged, layers 2 and 3 requested
        request(p.initSample, 0, p.dil[0], tpE, cdE);
        request(p.initSample, 1, p.dil[1], tpO, cdO);
        stage(tpE, cdE, have_tap(p.initSample, 0), 0);
        stage(tpO, cdO, have_tap(p.initSample, 1), 1);
        request(p.initSample, 2, p.dil[2], tpE, cdE);
        request(p.initSample, 3, p.dil[3], tpO, cdO);
        wg_barrier();   // P1
        wg_barrier();   // P2

        for (int t = p.initSample; t < tEnd; t++) {
            float selv[NPASS];
            load_selectors(t, selv);
            // embedding (nv_wavenet_reference.cpp:42-56)
            floatx4 x[BT][HTW];
#pragma unroll
            for (int bt = 0; bt < BT; bt++) {
#pragma unroll
                for (int i = 0; i < HTW; i++) {
                    const int tile = w + 4 * i;
                    floatx4 ec = quad_to_f32(*(const quad*)(embCur + (size_t)yCur[bt] * R + tile * 16 + g * 4));
                    floatx4 v = ep[bt][i] + ec;
                    if (p.tanhEmbed) {
#pragma unroll
                        for (int r = 0; r < 4; r++) v[r] = tanh_t<true>(v[r]);
                    }
                    x[bt][i] = v;
                    lds_put_tile<true>(xbuf + bt * KF_R * 1024, tile, lane, v);
                }
#pragma unroll
                for (int i = 0; i < HTW; i++)
                    ep[bt][i] = quad_to_f32(*(const quad*)(embPrev + (size_t)yCur[bt] * R + (w + 4 * i) * 16 + g * 4));
            }
            WN_STM(8)
            wg_barrier();   // E
            WN_STM(9)

            // one layer of role A; (tp, cdd): the register set of this layer's parity -- it holds layer l+2, which goes to LDS
            // now, and is re-requested for layer l+4
            auto layer = [&](auto PARc, const int l, frag (&tp)[TAPN], frag (&cdd)[BT][CR]) {
                constexpr int par = decltype(PARc)::value;          // l & 1
                constexpr int PHA = (par * FLA) % PFA;               // ring phase of this layer's fragments (the ring may span two layers)
                const int wl = l * FLA - PHA;
                const float* bl = biasLds + l * SC::BIAS_L;
                // P1: pre-activation prepared by role B + current tap -> gate -> h
                floatx4 acc[BT][2 * HTW];
                frag xb[BT][KF_R];
#pragma unroll
                for (int bt = 0; bt < BT; bt++) {
#pragma unroll
                    for (int it = 0; it < 2 * HTW; it++)
                        acc[bt][it] = *(const floatx4*)(accbuf + ((((bt * 4 + w) * 2 * HTW + it) * 64 + lane) << 4));
                    lds_get_frags<true, KF_R>(xbuf + bt * KF_R * 1024, lane, xb[bt]);
                }
                // (while those come back from LDS) layer l+2 -> LDS, layer l+4 requested into the same registers
                stage(tp, cdd, have_tap(t, l + 2), par);
                request(t, l + 4, p.dil[l + 4], tp, cdd);
                __builtin_amdgcn_sched_barrier(0);
                WN_STM(0)
                gemm_b<true, PFA, 0, BT, 2 * HTW, KF_R, SPIN>(ws, rsW, PHA, wl, 0, laneOff, acc, xb);
                WN_STM(1)
#pragma unroll
                for (int bt = 0; bt < BT; bt++)
#pragma unroll
                    for (int i = 0; i < HTW; i++) {
                        const floatx4 hv = gate4<true>(acc[bt][2 * i], acc[bt][2 * i + 1]);
                        lds_put_tile<true>(hbuf + par * SC::HBUF1 + bt * KF_R * 1024, w + 4 * i, lane, hv);
                    }
                WN_STM(2)
                wg_barrier();   // H
                WN_STM(3)
                // P2: residual
                frag hb[BT][KF_R];
                floatx4 xa[BT][HTW];
#pragma unroll
                for (int bt = 0; bt < BT; bt++) {
                    lds_get_frags<true, KF_R>(hbuf + par * SC::HBUF1 + bt * KF_R * 1024, lane, hb[bt]);
#pragma unroll
                    for (int i = 0; i < HTW; i++) xa[bt][i] = *(const floatx4*)(bl + 2 * R + (w + 4 * i) * 16 + g * 4) + x[bt][i];
                }
                WN_STM(4)
                gemm_b<true, PFA, 0, BT, HTW, KF_R, SPIN>(ws, rsW, PHA + SC::FW_GATE, wl, 0, laneOff, xa, hb);
#pragma unroll
                for (int bt = 0; bt < BT; bt++)
#pragma unroll
                    for (int i = 0; i < HTW; i++) {
                        x[bt][i] = xa[bt][i];
                        lds_put_tile<true>(xbuf + bt * KF_R * 1024, w + 4 * i, lane, xa[bt][i]);
                    }
                WN_STM(5)
                wg_barrier();   // X
                WN_STM(6)
            };
            for (int l = 0; l < L; l += 2) {
                layer(std::integral_constant<int, 0>{}, l, tpE, cdE);
                layer(std::integral_constant<int, 1>{}, l + 1, tpO, cdO);
            }
            draw_selectors(t, selv);
            wg_barrier();   // S
            WN_STM(10)
            head(ws, std::integral_constant<int, PFA>{}, L * FLA, selv, t);
            WN_STM(11)
#pragma unroll
            for (int bt = 0; bt < BT; bt++) {
                yPrev[bt] = yCur[bt];
                yCur[bt] = ybuf[bt * 16 + j];
            }
        }
#ifdef WN_SPLIT_TIMING
        if (tid == 0 && blockIdx.x == 0)
            for (int i = 0; i < 16; i++) p.p[i] = (float)tacc[i];
#endif
        if (w == 0 && g == 0) {
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
                if ((tile0 + bt) * 16 + j < p.batch) {
                    p.yInPrev[ub[bt]] = yPrev[bt];
                    p.yInCur[ub[bt]] = yCur[bt];
                }
        }
    } else {
        // =====================================================================================================
        // role B: everything that does not depend on the current x; its vector-memory queue carries the weight stream
        // (L2 hits) and the ring stores, nothing that waits for HBM
        // =====================================================================================================
        WStream<true, PFB, SPIN> ws;
#pragma unroll
        for (int i = 0; i < PFB; i++) ws.buf[i] = buf_load<frag, WN_W_AUX>(rsW, laneOff + (unsigned)(i & 3) * 1024u, (unsigned)(i & ~3) * 1024u);

        constexpr int XPW = C::XPW;
        const size_t ringTile = (size_t)__builtin_amdgcn_readfirstlane(p.ringSlots * KF_R * 1024);
        // ring stores: the wave that owns fragment k (k % 4 == w) stores it; the other waves issue the same instructions
        // through an empty buffer (out-of-range stores are dropped), so that the instruction stream is the same for all
        const rsrc_t rsRing = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr((const char*)p.ring + (size_t)tile0 * ringTile), 0,
                                                                w < KF_R ? -1 : 0, 0x00020000);
        const unsigned ringTileB = (unsigned)ringTile;
        const int kOwn = w < KF_R ? w : 0;
        // selA[tt]: A operand that copies the rows of tile tt of a B-layout fragment into a result tile (see wavenet_wg)
        frag selA[P::TPF];
#pragma unroll
        for (int tt = 0; tt < P::TPF; tt++)
#pragma unroll
            for (int e = 0; e < P::EPL; e++) selA[tt][e] = (elem)(((e >> 2) == tt && g * 4 + (e & 3) == j) ? 1.0f : 0.0f);
        // pre-activation of layer lN: bias + conditioning (through the 0/1 selection MFMAs, like wavenet_wg), ready for the
        // tap GEMM; the conditioning comes from the LDS slot role A staged it in (packed fragments, or raw fp16 quads)
        auto preact_init = [&](const int lN, const int par, floatx4 (&accN)[BT][2 * HTW]) {
            const float* blN = biasLds + lN * SC::BIAS_L;
            frag cd[BT][CR];
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int k = 0; k < CR; k++)
                    cd[bt][k] = *(const frag*)(condbuf + par * SC::CONDBUF1 + (((bt * 4 + w) * CR + k) * 64 + lane) * 16);
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < HTW; i++) {
                    accN[bt][2 * i] = *(const floatx4*)(blN + (w + 4 * i) * 16 + g * 4);
                    accN[bt][2 * i + 1] = *(const floatx4*)(blN + (w + 4 * i + RT) * 16 + g * 4);
                }
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int k = 0; k < C::COND_FR; k++)
#pragma unroll
                    for (int tt = 0; tt < P::TPF; tt++)
                        accN[bt][k * P::TPF + tt] = mma(selA[tt], cond_frag<true, RAW>(cd[bt], k), accN[bt][k * P::TPF + tt]);
        };
        auto put_acc = [&](const floatx4 (&accN)[BT][2 * HTW]) {
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int it = 0; it < 2 * HTW; it++)
                    *(floatx4*)(accbuf + ((((bt * 4 + w) * 2 * HTW + it) * 64 + lane) << 4)) = accN[bt][it];
        };

        // ---- prologue: pre-activation of layer 0 of the first sample ----
        wg_barrier();   // P1: role A has staged layers 0 and 1
        {
            frag xp[BT][KF_R];
#pragma unroll
            for (int bt = 0; bt < BT; bt++) lds_get_frags<true, KF_R>(xpbuf + bt * KF_R * 1024, lane, xp[bt]);
            floatx4 accN[BT][2 * HTW];
            preact_init(0, 0, accN);
            gemm_direct<true, BT, 2 * HTW, KF_R>(wbase + SC::posPrev(0, L) * 1024, laneOff, accN, xp);
            put_acc(accN);
        }
        wg_barrier();   // P2

        floatx4 skip[BT][STW];

        for (int t = p.initSample; t < tEnd; t++) {
            float selv[NPASS];
            load_selectors(t, selv);
            draw_selectors(t, selv);
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < STW; i++) skip[bt][i] = floatx4{0.f, 0.f, 0.f, 0.f};
            WN_STM(8)
            wg_barrier();   // E
            WN_STM(9)

            // layer l of role B.  Before H(l): the ring store of x_l, the pre-activation of layer l+1 (kept in registers:
            // role A reads that of layer l until H(l)), the first part of the skip GEMM of layer l-1 (h_{l-1} stays in its
            // buffer until layer l+1 starts).  Behind H(l): the pre-activation -> LDS, the rest of the skip GEMM.
            auto layer = [&](auto withSkip, const int l) {
                constexpr bool SKIP = decltype(withSkip)::value;
                const int lN = l + 1 < L ? l + 1 : 0;
                const int basePos = SKIP ? FLB * (l - 1) + SC::BASEB : 0;
                constexpr int PH = SKIP ? PHB : 0;
                const int par = l & 1;
                WN_STM(6)
                // x_l[t] replaces x_l[t-d] in the ring (same slot): this wave's fragment, read before x_l is overwritten
                {
                    const Dil dl = p.dil[l];
                    const unsigned rp = (unsigned)(dl.off + (t & (dl.d - 1))) * (unsigned)(KF_R * 1024);
#pragma unroll
                    for (int i = 0; i < XPW; i++) {
#pragma unroll
                        for (int bt = 0; bt < BT; bt++) {
                            const frag xs = *(const frag*)(xbuf + (((bt * KF_R + kOwn + 4 * i) * 64 + lane) << 4));
                            buf_store<frag, 2>(rsRing, laneOff, rp + (unsigned)bt * ringTileB + (unsigned)(kOwn + 4 * i) * 1024u, xs);
                        }
                    }
                }
                WN_STM(0)
                floatx4 accN[BT][2 * HTW];
                {
                    frag xp[BT][KF_R];
#pragma unroll
                    for (int bt = 0; bt < BT; bt++) lds_get_frags<true, KF_R>(xpbuf + (par ^ 1) * SC::XPBUF1 + bt * KF_R * 1024, lane, xp[bt]);
                    preact_init(lN, par ^ 1, accN);
                    gemm_b<true, PFB, 0, BT, 2 * HTW, KF_R, SPIN>(ws, rsW, PH, basePos, 0, laneOff, accN, xp);
                }
                WN_STM(1)
                if constexpr (SKIP) {
                    frag hb[BT][KF_R];
#pragma unroll
                    for (int bt = 0; bt < BT; bt++) lds_get_frags<true, KF_R>(hbuf + (par ^ 1) * SC::HBUF1 + bt * KF_R * 1024, lane, hb[bt]);
                    constexpr int G0 = STW >= 4 ? 4 : STW;
                    constexpr int NT = STW * KF_R;                  // takes of the skip GEMM, order of gemm_b
                    constexpr int CUT = NT * WN_SPLIT_SKIPCUT / 8;     // takes before H(l)
                    auto part = [&](auto lo, auto hi) {
                        static_for_range<decltype(lo)::value, decltype(hi)::value>([&](auto QI) {
                            constexpr int q = decltype(QI)::value;
                            constexpr int kf = (q / G0) % KF_R, mt = (q / (G0 * KF_R)) * G0 + q % G0;
                            frag a[1];
                            take_group<true, PFB, SPIN, 1>(ws, PH + SC::FW_GATE + q, a);
#pragma unroll
                            for (int bt = 0; bt < BT; bt++) skip[bt][mt] = mma(a[0], hb[bt][kf], skip[bt][mt]);
                            refill_group<true, PFB, 0, SPIN, 1>(ws, rsW, PH + SC::FW_GATE + q, basePos, 0, laneOff);
                        });
                    };
                    part(std::integral_constant<int, 0>{}, std::integral_constant<int, CUT>{});
                    WN_STM(2)
                    WN_STM(3)
                    wg_barrier();   // H
                    WN_STM(4)
                    put_acc(accN);
                    part(std::integral_constant<int, CUT>{}, std::integral_constant<int, NT>{});
                } else {
                    WN_STM(3)
                    wg_barrier();   // H
                    WN_STM(4)
                    put_acc(accN);
                }
                WN_STM(5)
                wg_barrier();   // X
            };
            layer(std::false_type{}, 0);
            for (int l = 1; l < L; l++) layer(std::true_type{}, l);
            // skip GEMM of the last layer, skip biases, ReLU -> LDS
            WN_STM(6)
            {
                frag hb[BT][KF_R];
#pragma unroll
                for (int bt = 0; bt < BT; bt++) lds_get_frags<true, KF_R>(hbuf + ((L - 1) & 1) * SC::HBUF1 + bt * KF_R * 1024, lane, hb[bt]);
                gemm_b<true, PFB, 0, BT, STW, KF_R, SPIN>(ws, rsW, PHB, FLB * (L - 1) + SC::BASEB, 0, laneOff, skip, hb);
            }
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < STW; i++) {
                    floatx4 v = skip[bt][i] + *(const floatx4*)(skipSum + (w + 4 * i) * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = __builtin_fmaxf(v[r], 0.f);
                    lds_put_tile<true>(skbuf + bt * KF_S * 1024, w + 4 * i, lane, v);
                }
            WN_STM(7)
            wg_barrier();   // S
            WN_STM(10)
            head(ws, std::integral_constant<int, PFB>{}, L * FLB, selv, t);
            WN_STM(11)
        }
#ifdef WN_SPLIT_TIMING
        if (tid == 256 && blockIdx.x == 0)
            for (int i = 0; i < 16; i++) p.p[16 + i] = (float)tacc[i];
#endif
    }
}

}  // namespace wn
