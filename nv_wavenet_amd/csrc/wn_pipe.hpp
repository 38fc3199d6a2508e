// wn_pipe.hpp -- the THROUGHPUT form of the multi-CU organisation: the chain of wn_chain.hpp kept full.
//
// wavenet_wg / wavenet_stream re-stream the model's weights through every CU for every sample: at C3 that is
// 1.76 MB per sample and workgroup over a 58 B/clk L1 path, a floor of ~15 us per pass however many tiles share
// it, and the reason the full-chip point sits at 18 % of the MFMA rate.  wavenet_chain removes the stream (every
// stage keeps its layers' weights on chip) but leaves each CU idle four fifths of the time: one tile, one sample
// in flight per chain.  Here the same chain of stages is kept FULL:
//
//   * a chain serves NG groups of G tiles (G = 4: 64 utterances per group).  Every stage works through the groups
//     round-robin, sample after sample; while group g is in stage s, group g+1 is in stage s-1: with NG >= stages
//     (+ hop latency) no stage ever waits, and each utterance still advances one sample per NG stage-steps.
//   * inside a step the G tiles are interleaved: every weight fragment (AGPR-pinned or in LDS, never re-read from
//     memory) feeds G MFMAs, one barrier serves the h (or x) exchange of all G tiles, and the gate VALU work of
//     one tile runs under the MFMAs of the others.
//   * what a step needs from HBM -- the conditioning and the dilated taps of a layer, G tiles each -- is requested
//     one layer ahead, into the registers the previous layer's operands have just left.
//   * arithmetic, operand rounding and summation order are those of wavenet_wg / wavenet_chain: bit-identical
//     samples (fp16 and fp32).
//
// Hand-offs are the tagged granules of wn_chain.hpp, one single-slot mailbox per (stage, group): the
// autoregressive loop of a group is its flow control exactly as for the one tile of wavenet_chain.  Per
// utterance-sample the organisation moves 5 KB of conditioning and 5 KB of ring traffic through HBM (like every
// organisation) and ~10 KB of granules through L2 -- and no weights at all.
#pragma once

#include "wn_chain.hpp"

namespace wn {

struct PipeParams {
    unsigned long long* mail;   // placement words, then [chain][stage][group] mailboxes; zeroed before every launch
    unsigned* status;
    int stages;                 // layer stages + 1 (head)
    int lpc;                    // layers per layer stage
    int chains;
    int groups;                 // groups of G tiles per chain (NG)
    int tiles;                  // tiles of the batch (ceil(batch / 16)); tile = (chain * groups + g) * G + bt; Params::tiles = allocated tiles
};

template <bool F16, int R, int S, int A>
struct PCfg {
    using C = Cfg<F16, R, S, A, 1>;
    static constexpr int NW = C::NW, FLW = C::FLW, FHW = C::FHW;
    static constexpr int LDS_MAX = 160 * 1024;
#ifndef WN_PIPE_G
#define WN_PIPE_G 2
#endif
    static constexpr int G = WN_PIPE_G;                    // tiles interleaved per step
    static constexpr int AGPR_FRAGS = F16 ? 64 : 40;       // resident weight fragments per wave in the accumulator file
    // layer stage with n layers: G x images | n*G h images | n x (Bh, Bres) | weights beyond the AGPRs
    static constexpr int fixedLds(int n) { return G * C::XBUF + n * G * C::HBUF + n * 3 * R * 4; }
    static constexpr int ldsFrags(int n) { return n * FLW > AGPR_FRAGS ? n * FLW - AGPR_FRAGS : 0; }   // per wave, whole stage
    static constexpr bool fits(int n) { return fixedLds(n) + ldsFrags(n) * NW * 1024 <= LDS_MAX; }
    static constexpr int pickLpc() {
        int best = 0;
        for (int n = 1; n <= 8; n++)
            if (fits(n)) best = n;
        return best;
    }
    static constexpr int LPC = pickLpc();
    static constexpr bool SUPPORTED = LPC > 0 && FHW <= AGPR_FRAGS;   // (the head keeps its weights in AGPRs)
    static constexpr int LP = LPC > 0 ? LPC : 1;
    // the stage's resident stream: fragment (li, idx) has global position li*FLW + idx; positions < NAG in AGPRs, else LDS
    static constexpr int NTOT = LP * FLW;
    static constexpr int NAG = NTOT < AGPR_FRAGS ? NTOT : AGPR_FRAGS;
    static constexpr int NLD = NTOT - NAG;
    static constexpr int OFF_LX = 0, OFF_LH = G * C::XBUF, OFF_LB = OFF_LH + LP * G * C::HBUF;
    static constexpr int OFF_LW = (OFF_LB + LP * 3 * R * 4 + 15) & ~15;
    static constexpr int LAYER_LDS = OFF_LW + NW * NLD * 1024;
    // head stage: G skip images | G zs images | one logits tile | picks | biases | embedding tables
    static constexpr int OFF_HSK = 0, OFF_HZS = G * C::SKBUF, OFF_HLG = OFF_HZS + G * C::ZSBUF;
    static constexpr int OFF_HY = OFF_HLG + 16 * C::LROW * 4;
    static constexpr int MAX_GROUPS = 16;                  // groups per chain (history of MAX_GROUPS * G tiles in the head's LDS)
    static constexpr int OFF_HB = OFF_HY + MAX_GROUPS * G * 16 * 2 * 4;
    static constexpr int OFF_HE = (OFF_HB + (S + 2 * A) * 4 + 15) & ~15;
    static size_t headLds(int embTables) { return (size_t)OFF_HE + (size_t)embTables * A * R * sizeof(typename C::P::elem); }
    static int embTables() { return headLds(2) <= (size_t)LDS_MAX ? 2 : headLds(1) <= (size_t)LDS_MAX ? 1 : 0; }
    static size_t ldsBytes() {
        const size_t h = headLds(embTables());
        return h > (size_t)LAYER_LDS ? h : (size_t)LAYER_LDS;
    }
    static constexpr int XG = R * 16, SG = S * 16;         // granules of one tile's x / skip message
    static constexpr size_t boxGranules() { return (size_t)G * (XG + SG); }                // one (stage, group) mailbox
    static constexpr size_t placeWords(int chains, int stages) { return ((size_t)chains * stages + 63) & ~(size_t)63; }
    static constexpr size_t mailGranules(int chains, int stages, int groups) {
        return placeWords(chains, stages) + (size_t)chains * stages * groups * boxGranules();
    }
};

// One sweep pass over this wave's tiles of ALL the GT tiles of a group (tile bt's mailbox at mbox + bt*stride):
// the loads of every tile are in flight together, so a group costs one memory round trip, not GT of them.
template <int NT, int NW, int GT>
WN_DEV bool recv_group(const unsigned long long* mbox, size_t stride, int w, int lane, unsigned tag, floatx4 (&v)[GT][NT],
                       gu32* status, unsigned code) {
    Spin s{status, (long long)wall_clock64(), 0u};
    for (;;) {
        unsigned long long q[GT][NT * 4];
#pragma unroll
        for (int bt = 0; bt < GT; bt++) sweep_issue<NT, NW>(mbox + bt * stride, w, lane, q[bt]);
        bool ok = true;
#pragma unroll
        for (int bt = 0; bt < GT; bt++) ok = sweep_check<NT>(q[bt], tag, v[bt]) && ok;
        if (ok) return true;
        if (!spin_more(s, code)) return false;
    }
}

// acc[bt][mt] += W(tile slot mt) * b[bt] for the G tiles of a group; the weight fragment at stage position
// pos0 + ... comes from the AGPR-pinned array or from this wave's LDS slice (fragment order of gemm())
template <bool F16, typename PC, int BT, int MT, int KF>
WN_DEV void gemm_p(const floatx4 (&wag)[PC::NAG ? PC::NAG : 1], const char* wlds, unsigned laneOff, int pos0, floatx4 (&acc)[BT][MT],
                   const typename Prec<F16>::frag (&b)[BT][KF]) {
    using frag = typename Prec<F16>::frag;
    constexpr int Gs = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / Gs; mg++)
#pragma unroll
        for (int kf = 0; kf < KF; kf++)
#pragma unroll
            for (int mi = 0; mi < Gs; mi++) {
                const int pos = pos0 + (mg * KF + kf) * Gs + mi;
                frag a;
                if (pos < PC::NAG) a = __builtin_bit_cast(frag, wag[pos < PC::NAG ? pos : 0]);
                else a = *(const frag*)(wlds + (size_t)(pos - PC::NAG) * 1024 + laneOff);
#pragma unroll
                for (int bt = 0; bt < BT; bt++) acc[bt][mg * Gs + mi] = mma(a, b[bt][kf], acc[bt][mg * Gs + mi]);
            }
}

// one tile
template <bool F16, typename PC, int MT, int KF>
WN_DEV void gemm_p1(const floatx4 (&wag)[PC::NAG ? PC::NAG : 1], const char* wlds, unsigned laneOff, int pos0, floatx4 (&acc)[MT],
                    const typename Prec<F16>::frag (&b)[KF]) {
    using frag = typename Prec<F16>::frag;
    constexpr int Gs = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / Gs; mg++)
#pragma unroll
        for (int kf = 0; kf < KF; kf++)
#pragma unroll
            for (int mi = 0; mi < Gs; mi++) {
                const int pos = pos0 + (mg * KF + kf) * Gs + mi;
                frag a;
                if (pos < PC::NAG) a = __builtin_bit_cast(frag, wag[pos < PC::NAG ? pos : 0]);
                else a = *(const frag*)(wlds + (size_t)(pos - PC::NAG) * 1024 + laneOff);
                acc[mg * Gs + mi] = mma(a, b[kf], acc[mg * Gs + mi]);
            }
}

// the same with AGPR-pinned fragments wres[pos0 ...] only (the head)
template <bool F16, int BT, int MT, int KF, int NFR>
WN_DEV void gemm_pinned_bt(const floatx4 (&wres)[NFR], int pos0, floatx4 (&acc)[BT][MT], const typename Prec<F16>::frag (&b)[BT][KF]) {
    using frag = typename Prec<F16>::frag;
    constexpr int Gs = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / Gs; mg++)
#pragma unroll
        for (int kf = 0; kf < KF; kf++)
#pragma unroll
            for (int mi = 0; mi < Gs; mi++) {
                const frag a = __builtin_bit_cast(frag, wres[pos0 + (mg * KF + kf) * Gs + mi]);
#pragma unroll
                for (int bt = 0; bt < BT; bt++) acc[bt][mg * Gs + mi] = mma(a, b[bt][kf], acc[bt][mg * Gs + mi]);
            }
}

// ------------------------------------------------------------------------------------------------
// layer stage
// ------------------------------------------------------------------------------------------------
template <bool F16, int R, int S, int A, bool DUMP>
WN_DEV void pipe_layers(const Params& p, const PipeParams& pp, char* lds, int chain, int stage) {
    using PC = PCfg<F16, R, S, A>;
    using C = typename PC::C;
    using P = Prec<F16>;
    using frag = typename P::frag;
    using elem = typename P::elem;
    constexpr int G = PC::G, LP = PC::LP, NW = C::NW, FLW = C::FLW;
    constexpr int RT = C::RT, HTW = C::HTW, STW = C::STW, KF_R = C::KF_R;

    char* const xbuf = lds + PC::OFF_LX;                   // [G] x images
    char* const hbuf = lds + PC::OFF_LH;                   // [LP][G] h images
    float* const biasLds = (float*)(lds + PC::OFF_LB);     // [LP][Bh 2R | Bres R]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, j = lane & 15;
    const int L = p.numLayers;
    const int l0 = stage * pp.lpc;
    const int nl = cmin(pp.lpc, L - l0);
    const bool lastLayerStage = stage == pp.stages - 2;
    const unsigned laneOff = (unsigned)lane * 16u;
    char* const wlds = lds + PC::OFF_LW + (size_t)w * PC::NLD * 1024;

    unsigned long long* const boxes = pp.mail + PC::placeWords(pp.chains, pp.stages);
    auto boxOf = [&](int st, int grp) { return boxes + (((size_t)chain * pp.stages + st) * pp.groups + grp) * PC::boxGranules(); };
    gu32* const status = (gu32*)pp.status;
    bool sameXcd = false;
    if (!chain_place(pp.mail, chain * pp.stages + stage, chain * pp.stages + stage + 1, status, sameXcd)) return;

    for (int i = tid; i < nl * 3 * R; i += C::THREADS) biasLds[i] = p.bias[(size_t)(l0 + i / (3 * R)) * C::BIAS_L + i % (3 * R)];

    // resident weights: the stage's layers back to back; positions < NAG pinned in AGPRs, the rest in LDS
    const char* const wbase = (const char*)p.wblob + (size_t)w * C::waveStreamFrags(L) * 1024;
    floatx4 wag[PC::NAG ? PC::NAG : 1];
#pragma unroll
    for (int li = 0; li < LP; li++) {
        const int lw = l0 + (li < nl ? li : 0);            // (the stream is in wavenet_wg's consumption order: Cfg::streamPos)
#pragma unroll
        for (int i = 0; i < FLW; i++) {
            const int pos = li * FLW + i;
            const frag f = *(const frag*)(wbase + C::streamPos(lw, i, L) * 1024 + laneOff);
            if (pos < PC::NAG) wag[pos < PC::NAG ? pos : 0] = agpr_pin(__builtin_bit_cast(floatx4, f));
            else if (li < nl) *(frag*)(wlds + (size_t)(pos - PC::NAG) * 1024 + laneOff) = f;
        }
    }

    Dil dl[LP];
    {
        Dil s = dil_first();
        for (int l = 0; l < l0; l++) s = dil_next(s, p.maxDilation, false);
#pragma unroll
        for (int li = 0; li < LP; li++) {
            dl[li] = s;
            s = dil_next(s, p.maxDilation, false);
        }
    }
    const size_t condStride = (size_t)p.tiles * NW * C::COND_FR * 1024;
    const size_t ringTile = (size_t)p.ringSlots * KF_R * 1024;

    frag selA[P::TPF];
#pragma unroll
    for (int tt = 0; tt < P::TPF; tt++)
#pragma unroll
        for (int e = 0; e < P::EPL; e++) selA[tt][e] = (elem)(((e >> 2) == tt && g4 * 4 + (e & 3) == j) ? 1.0f : 0.0f);

    // conditioning + dilated tap of (sample t, own layer li) for the G tiles of a group, one layer ahead of their use
    frag cd[G][C::COND_FR], xp[G][KF_R];
    auto prefetch = [&](int t, int grp, int li) {
        const int tile0 = (chain * pp.groups + grp) * G;
        const int d = dl[li].d;
        const unsigned slot = (unsigned)(dl[li].off + (t & (d - 1)));
#pragma unroll
        for (int bt = 0; bt < G; bt++) {
            const char* cp0 = (const char*)p.cond + ((size_t)(tile0 + bt) * NW + w) * C::COND_FR * 1024 +
                              ((size_t)t * L + (l0 + li)) * condStride;
            const char* rp = (const char*)p.ring + (size_t)(tile0 + bt) * ringTile + (size_t)slot * (KF_R * 1024);
#pragma unroll
            for (int k = 0; k < C::COND_FR; k++) cd[bt][k] = __builtin_nontemporal_load((const frag*)(cp0 + k * 1024 + laneOff));
#pragma unroll
            for (int k = 0; k < KF_R; k++) xp[bt][k] = __builtin_nontemporal_load((const frag*)(rp + (size_t)k * 1024 + laneOff));
        }
    };

    __syncthreads();
    const int tEnd = p.initSample + p.count;
    // groups this chain actually serves; with a single group the next step's operands are this step's ring stores:
    // they are requested at the step start, behind the store-completion barrier, instead of one layer ahead
    int myGroups = 0;
    for (int grp = 0; grp < pp.groups; grp++) myGroups += (chain * pp.groups + grp) * G < pp.tiles ? 1 : 0;
    const bool lateFirst = myGroups <= 1;
    if (!lateFirst) prefetch(p.initSample, 0, 0);
    for (int t = p.initSample; t < tEnd; t++) {
        const unsigned tag = (unsigned)(t - p.initSample) + 1u;
        const bool dumpNow = DUMP && p.dump && (t == tEnd - 1);
        for (int grp = 0; grp < pp.groups; grp++) {
            const int tile0 = (chain * pp.groups + grp) * G;
            if (tile0 >= pp.tiles) continue;                       // (a chain's last groups may be empty)
            const unsigned long long* const xin = boxOf(stage, grp);
            const unsigned long long* const skin = xin + (size_t)G * PC::XG;
            unsigned long long* const xout = boxOf(stage + 1, grp);
            unsigned long long* const skout = xout + (size_t)G * PC::XG;
            // ring stores of this group's previous sample (and the weights / biases of the prologue) are complete
            // and visible to the whole workgroup; h images of the previous step are no longer read
            WN_CT_DECL
            WN_CT(0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_barrier();
            WN_CT(1)
            if (lateFirst) prefetch(t, grp, 0);

            // ---- the group's x tiles arrive (each wave its own tiles of all G tiles) ----------------------
            floatx4 x[G][HTW];
            if (!recv_group<HTW, NW, G>(xin, PC::XG, w, lane, tag, x, status, 0x100u + (unsigned)stage)) return;
#pragma unroll
            for (int bt = 0; bt < G; bt++)
#pragma unroll
                for (int i = 0; i < HTW; i++) lds_put_tile<F16>(xbuf + bt * C::XBUF, w + NW * i, lane, x[bt][i]);
            wg_barrier();
            WN_CT(2)

#pragma unroll
            for (int li = 0; li < LP; li++) {
                if (li < nl) {
                    const int l = l0 + li;
                    const float* bl = biasLds + li * 3 * R;
                    const int d = dl[li].d;
                    const bool havePrev = t >= d;
                    frag xb[G][KF_R];
                    floatx4 acc[G][2 * HTW];
#pragma unroll
                    for (int bt = 0; bt < G; bt++) {
                        lds_get_frags<F16, KF_R>(xbuf + bt * C::XBUF, lane, xb[bt]);
#pragma unroll
                        for (int i = 0; i < HTW; i++) {
                            acc[bt][2 * i] = *(const floatx4*)(bl + (w + NW * i) * 16 + g4 * 4);
                            acc[bt][2 * i + 1] = *(const floatx4*)(bl + (w + NW * i + RT) * 16 + g4 * 4);
                        }
                        // + conditioning, + dilated tap (zero before the start of the utterance)
                        if constexpr (F16) {
#pragma unroll
                            for (int k = 0; k < C::COND_FR; k++)
#pragma unroll
                                for (int tt = 0; tt < P::TPF; tt++)
                                    acc[bt][k * P::TPF + tt] = mma(selA[tt], cd[bt][k], acc[bt][k * P::TPF + tt]);
                        } else {
#pragma unroll
                            for (int k = 0; k < C::COND_FR; k++)
#pragma unroll
                                for (int e = 0; e < P::EPL; e++) acc[bt][k * P::TPF + (e >> 2)][e & 3] += (float)cd[bt][k][e];
                        }
                        if (!havePrev) {
#pragma unroll
                            for (int k = 0; k < KF_R; k++)
#pragma unroll
                                for (int e = 0; e < P::EPL; e++) xp[bt][k][e] = (elem)0.f;
                        }
                    }
                    if (li == 0) WN_CT(8)
                    gemm_p<F16, PC, G, 2 * HTW, KF_R>(wag, wlds, laneOff, li * FLW + C::O_PREV, acc, xp);
                    if (li == 0) WN_CT(9)
                    // x_l[t] replaces x_l[t-d] in the ring (each wave stores its share of the fragments)
#pragma unroll
                    for (int bt = 0; bt < G; bt++) {
                        char* rp = (char*)p.ring + (size_t)(tile0 + bt) * ringTile + (size_t)(unsigned)(dl[li].off + (t & (d - 1))) * (KF_R * 1024);
#pragma unroll
                        for (int i = 0; i < C::XPW; i++) {
                            const int k = w + NW * i;
                            if (k < KF_R)
                                __builtin_nontemporal_store(*(const frag*)(xbuf + bt * C::XBUF + (size_t)k * 1024 + laneOff),
                                                            (frag*)(rp + (size_t)k * 1024 + laneOff));
                        }
                    }
                    // the operands of the NEXT own layer (or of the next group's first layer) are requested now,
                    // into the registers this layer's have just left
                    {
                        int nli = li + 1, ngrp = grp, nt = t;
                        if (nli >= nl) {
                            nli = 0;
                            ngrp = grp + 1;
                            if (ngrp >= pp.groups || (chain * pp.groups + ngrp) * G >= pp.tiles) {
                                ngrp = 0;
                                nt = t + 1;
                            }
                        }
                        if (nt < tEnd) {
                            // (the layer index must be a compile-time constant of the unrolled body: both cases spelled out)
                            if (li + 1 < nl) prefetch(nt, ngrp, li + 1 < LP ? li + 1 : 0);
                            else if (!lateFirst) prefetch(nt, ngrp, 0);
                        }
                    }
                    if (li == 0) WN_CT(10)
                    // tile by tile: the gate (VALU) of one tile runs under the current-tap MFMAs of the next
#pragma unroll
                    for (int bt = 0; bt < G; bt++) {
                        gemm_p1<F16, PC, 2 * HTW, KF_R>(wag, wlds, laneOff, li * FLW + C::O_CUR, acc[bt], xb[bt]);
#pragma unroll
                        for (int i = 0; i < HTW; i++) {
                            const floatx4 hv = gate4<F16>(acc[bt][2 * i], acc[bt][2 * i + 1]);
                            lds_put_tile<F16>(hbuf + (li * G + bt) * C::HBUF, w + NW * i, lane, hv);
                        }
                    }
                    if (li == 0) WN_CT(11)
                    wg_barrier();   // h of all G tiles complete
                    if (li == 0) WN_CT(12)
                    frag hb[G][KF_R];
                    floatx4 xa[G][HTW];
#pragma unroll
                    for (int bt = 0; bt < G; bt++) {
                        lds_get_frags<F16, KF_R>(hbuf + (li * G + bt) * C::HBUF, lane, hb[bt]);
#pragma unroll
                        for (int i = 0; i < HTW; i++) xa[bt][i] = *(const floatx4*)(bl + 2 * R + (w + NW * i) * 16 + g4 * 4) + x[bt][i];
                    }
                    gemm_p<F16, PC, G, HTW, KF_R>(wag, wlds, laneOff, li * FLW + C::O_RES, xa, hb);
#pragma unroll
                    for (int bt = 0; bt < G; bt++) {
#pragma unroll
                        for (int i = 0; i < HTW; i++) x[bt][i] = xa[bt][i];
                        const int b = (tile0 + bt) * 16 + j;
                        if (dumpNow && b < p.batch) {
#pragma unroll
                            for (int i = 0; i < HTW; i++)
                                *(floatx4*)(p.xtOut + ((size_t)l * p.maxBatch + b) * R + (w + NW * i) * 16 + g4 * 4) = x[bt][i];
                        }
                    }
                    if (li == 0) WN_CT(13)
                    if (li + 1 < nl) {
#pragma unroll
                        for (int bt = 0; bt < G; bt++)
#pragma unroll
                            for (int i = 0; i < HTW; i++) lds_put_tile<F16>(xbuf + bt * C::XBUF, w + NW * i, lane, x[bt][i]);
                        wg_barrier();   // x of all G tiles complete
                        if (li == 0) WN_CT(14)
                    }
                }
            }
            WN_CT(3)
            if (!lastLayerStage) {
#pragma unroll
                for (int bt = 0; bt < G; bt++) send_tiles<HTW, NW>(xout + (size_t)bt * PC::XG, w, lane, tag, x[bt], sameXcd);
            }
            WN_CT(4)

            // ---- running skip sums of the G tiles: every Wskip fragment feeds G MFMAs -------------------------
            floatx4 sk[G][STW];
            if (stage == 0) {
#pragma unroll
                for (int bt = 0; bt < G; bt++)
#pragma unroll
                    for (int i = 0; i < STW; i++) sk[bt][i] = floatx4{0.f, 0.f, 0.f, 0.f};
            } else {
                constexpr int HG = G >= 2 ? 2 : 1;          // tiles per sweep: 2 x 16 granules = 64 registers in flight
#pragma unroll
                for (int b0 = 0; b0 < G; b0 += HG) {
                    floatx4 part[HG][STW];
                    if (!recv_group<STW, NW, HG>(skin + (size_t)b0 * PC::SG, PC::SG, w, lane, tag, part, status, 0x200u + (unsigned)stage)) return;
#pragma unroll
                    for (int h = 0; h < HG; h++)
#pragma unroll
                        for (int i = 0; i < STW; i++)
                            if (b0 + h < G) sk[b0 + h][i] = part[h][i];
                }
            }
            WN_CT(5)
#pragma unroll
            for (int li = 0; li < LP; li++) {
                if (li < nl) {
                    frag hb[G][KF_R];
#pragma unroll
                    for (int bt = 0; bt < G; bt++) lds_get_frags<F16, KF_R>(hbuf + (li * G + bt) * C::HBUF, lane, hb[bt]);
                    gemm_p<F16, PC, G, STW, KF_R>(wag, wlds, laneOff, li * FLW + C::O_SKIP, sk, hb);
                    if (dumpNow && l0 + li < L - 1) {
#pragma unroll
                        for (int bt = 0; bt < G; bt++) {
                            const int b = (tile0 + bt) * 16 + j;
                            if (b < p.batch) {
#pragma unroll
                                for (int i = 0; i < STW; i++) {
                                    const int row = (w + NW * i) * 16 + g4 * 4;
                                    floatx4 run = *(const floatx4*)(p.bias + 3 * R + row);
                                    for (int l = 1; l <= l0 + li; l++) run += *(const floatx4*)(p.bias + (size_t)l * C::BIAS_L + 3 * R + row);
                                    *(floatx4*)(p.skipOut + ((size_t)(l0 + li) * p.maxBatch + b) * S + row) = sk[bt][i] + run;
                                }
                            }
                        }
                    }
                }
            }
            WN_CT(6)
#pragma unroll
            for (int bt = 0; bt < G; bt++) send_tiles<STW, NW>(skout + (size_t)bt * PC::SG, w, lane, tag, sk[bt], sameXcd);
            WN_CT(7)
            if (grp == 0 && chain == 0) WN_CT_FLUSH(stage, t - p.initSample)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// head stage
// ------------------------------------------------------------------------------------------------
template <bool F16, int R, int S, int A, bool DUMP>
WN_DEV void pipe_head(const Params& p, const PipeParams& pp, char* lds, int chain) {
    using PC = PCfg<F16, R, S, A>;
    using C = typename PC::C;
    using P = Prec<F16>;
    using frag = typename P::frag;
    using quad = typename P::quad;
    using elem = typename P::elem;
    constexpr int G = PC::G, NW = C::NW;
    constexpr int HTW = C::HTW, STW = C::STW, ATW = C::ATW, KF_S = C::KF_S, KF_A = C::KF_A, FHW = C::FHW;

    char* const skbuf = lds + PC::OFF_HSK;                 // [G]
    char* const zsbuf = lds + PC::OFF_HZS;                 // [G]
    float* const lgbuf = (float*)(lds + PC::OFF_HLG);      // one tile at a time
    int* const hist = (int*)(lds + PC::OFF_HY);            // [groups][G][16][prev, cur]: the two last indices of every utterance
    float* const fsb = (float*)(lds + PC::OFF_HB);
    float* const headBias = fsb + S;
    elem* const embLds = (elem*)(lds + PC::OFF_HE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, j = lane & 15;
    const int L = p.numLayers;
    const int stage = pp.stages - 1;
    const unsigned laneOff = (unsigned)lane * 16u;
    const int su = tid / C::LPU, sq = tid % C::LPU;

    unsigned long long* const boxes = pp.mail + PC::placeWords(pp.chains, pp.stages);
    auto boxOf = [&](int st, int grp) { return boxes + (((size_t)chain * pp.stages + st) * pp.groups + grp) * PC::boxGranules(); };
    gu32* const status = (gu32*)pp.status;
    bool sameXcd = false;
    if (!chain_place(pp.mail, chain * pp.stages + stage, chain * pp.stages, status, sameXcd)) return;

    for (int s0 = tid; s0 < S; s0 += C::THREADS) {
        float run = p.bias[3 * R + s0];
        for (int l = 1; l < L; l++) run += p.bias[(size_t)l * C::BIAS_L + 3 * R + s0];
        fsb[s0] = run;
    }
    for (int i = tid; i < 2 * A; i += C::THREADS) headBias[i] = p.bias[(size_t)L * C::BIAS_L + i];
    const int nEmb = p.embLds;
    if (nEmb > 0) {
        const floatx4* s0 = (const floatx4*)p.embCur;
        const floatx4* s1 = (const floatx4*)p.embPrev;
        constexpr int CH = (int)(A * R * sizeof(elem) / 16);
        for (int i = tid; i < CH; i += C::THREADS) {
            ((floatx4*)embLds)[i] = s0[i];
            if (nEmb > 1) ((floatx4*)embLds)[CH + i] = s1[i];
        }
    }
    const elem* const gEmbPrev = (const elem*)p.embPrev;
    const elem* const gEmbCur = (const elem*)p.embCur;
    auto rowCur = [&](int y, int tile16) -> floatx4 {
        const size_t off = (size_t)y * R + tile16 * 16 + g4 * 4;
        if (nEmb > 0) return quad_to_f32(*(const quad*)(embLds + off));
        return quad_to_f32(*(const quad*)(gEmbCur + off));
    };
    auto rowPrev = [&](int y, int tile16) -> floatx4 {
        const size_t off = (size_t)y * R + tile16 * 16 + g4 * 4;
        if (nEmb > 1) return quad_to_f32(*(const quad*)(embLds + (size_t)A * R + off));
        return quad_to_f32(*(const quad*)(gEmbPrev + off));
    };

    const char* const whead = (const char*)p.wblob + (size_t)w * C::waveStreamFrags(L) * 1024 + C::headOffsetFrags(L) * 1024;
    floatx4 hw[FHW];
#pragma unroll
    for (int i = 0; i < FHW; i++) hw[i] = agpr_pin(__builtin_bit_cast(floatx4, *(const frag*)(whead + (size_t)C::headFrag(i) * 1024 + laneOff)));
    __syncthreads();

    for (int i = tid; i < pp.groups * G * 16; i += C::THREADS) {
        int b = chain * pp.groups * G * 16 + i;
        b = b < p.batch ? b : p.batch - 1;
        hist[2 * i] = p.yInPrev[b];
        hist[2 * i + 1] = p.yInCur[b];
    }
    __syncthreads();
    auto embed_and_send = [&](int grp, unsigned tag) {
        const int tile0 = (chain * pp.groups + grp) * G;
        unsigned long long* const xout = boxOf(0, grp);
#pragma unroll
        for (int bt = 0; bt < G; bt++) {
            const int hi = ((grp * G + bt) * 16 + j) * 2;
            const int yPrev = hist[hi], yCur = hist[hi + 1];
            floatx4 x0[HTW];
#pragma unroll
            for (int i = 0; i < HTW; i++) {
                const int tile16 = w + NW * i;
                floatx4 v = rowPrev(yPrev, tile16) + rowCur(yCur, tile16);
                if (p.tanhEmbed) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = tanh_t<F16>(v[r]);
                }
                x0[i] = v;
            }
            send_tiles<HTW, NW>(xout + (size_t)bt * PC::XG, w, lane, tag, x0, sameXcd);
        }
    };
    for (int grp = 0; grp < pp.groups; grp++)
        if ((chain * pp.groups + grp) * G < pp.tiles) embed_and_send(grp, 1u);

    const int tEnd = p.initSample + p.count;
    for (int t = p.initSample; t < tEnd; t++) {
        const unsigned tag = (unsigned)(t - p.initSample) + 1u;
        const bool dumpNow = DUMP && p.dump && (t == tEnd - 1);
        for (int grp = 0; grp < pp.groups; grp++) {
            const int tile0 = (chain * pp.groups + grp) * G;
            if (tile0 >= pp.tiles) continue;
            const unsigned long long* const skin = boxOf(stage, grp) + (size_t)G * PC::XG;
            // ---- skip sums of all layers, G tiles; + biases, ReLU -> B fragments ------------------------
#pragma unroll
            for (int bt = 0; bt < G; bt++) {
                floatx4 sk[STW];
                if (!recv_tiles<STW, NW>(skin + (size_t)bt * PC::SG, w, lane, tag, sk, status, 0x300u)) return;
                const int b = (tile0 + bt) * 16 + j;
#pragma unroll
                for (int i = 0; i < STW; i++) {
                    floatx4 v = sk[i] + *(const floatx4*)(fsb + (w + NW * i) * 16 + g4 * 4);
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = __builtin_fmaxf(v[r], 0.f);
                    lds_put_tile<F16>(skbuf + bt * C::SKBUF, w + NW * i, lane, v);
                    if (dumpNow && b < p.batch)
                        *(floatx4*)(p.skipOut + ((size_t)(L - 1) * p.maxBatch + b) * S + (w + NW * i) * 16 + g4 * 4) = v;
                }
            }
            wg_barrier();
            {
                floatx4 zs[G][ATW];
                frag sbf[G][KF_S];
#pragma unroll
                for (int bt = 0; bt < G; bt++) {
                    lds_get_frags<F16, KF_S>(skbuf + bt * C::SKBUF, lane, sbf[bt]);
#pragma unroll
                    for (int i = 0; i < ATW; i++) zs[bt][i] = *(const floatx4*)(headBias + (w + NW * i) * 16 + g4 * 4);
                }
                gemm_pinned_bt<F16, G, ATW, KF_S>(hw, 0, zs, sbf);
#pragma unroll
                for (int bt = 0; bt < G; bt++) {
                    const int b = (tile0 + bt) * 16 + j;
#pragma unroll
                    for (int i = 0; i < ATW; i++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) zs[bt][i][r] = __builtin_fmaxf(zs[bt][i][r], 0.f);
                        lds_put_tile<F16>(zsbuf + bt * C::ZSBUF, w + NW * i, lane, zs[bt][i]);
                        if (dumpNow && b < p.batch) *(floatx4*)(p.zs + (size_t)b * A + (w + NW * i) * 16 + g4 * 4) = zs[bt][i];
                    }
                }
            }
            wg_barrier();
            floatx4 za[G][ATW];
            {
                frag zb[G][KF_A];
#pragma unroll
                for (int bt = 0; bt < G; bt++) {
                    lds_get_frags<F16, KF_A>(zsbuf + bt * C::ZSBUF, lane, zb[bt]);
#pragma unroll
                    for (int i = 0; i < ATW; i++) za[bt][i] = *(const floatx4*)(headBias + A + (w + NW * i) * 16 + g4 * 4);
                }
                gemm_pinned_bt<F16, G, ATW, KF_A>(hw, C::FW_ZS, za, zb);
            }
            // ---- softmax + pick, one tile at a time through the one logits image ---------------------------
#pragma unroll
            for (int bt = 0; bt < G; bt++) {
                const int b = (tile0 + bt) * 16 + j;
#pragma unroll
                for (int i = 0; i < ATW; i++) {
                    *(floatx4*)(lgbuf + j * C::LROW + (w + NW * i) * 16 + g4 * 4) = za[bt][i];
                    if (dumpNow && b < p.batch) *(floatx4*)(p.za + (size_t)b * A + (w + NW * i) * 16 + g4 * 4) = za[bt][i];
                }
                wg_barrier();
                int sb = (tile0 + bt) * 16 + su;
                const bool sbValid = sb < p.batch;
                sb = sbValid ? sb : p.batch - 1;
                const float selv = p.useRng ? philox_selector(p.rngKey0, p.rngKey1, (unsigned)t, (unsigned)sb)
                                            : p.sel[(size_t)t * p.maxBatch + sb];
                float e[C::RPL];
                float total;
                const int pick = softmax_pick<A, C::LPU, C::RPL>(lgbuf + su * C::LROW + sq * C::RPL, sq, lane, selv, e, total);
                if (sq == 0) {
                    const int hi = ((grp * G + bt) * 16 + su) * 2;
                    hist[hi] = hist[hi + 1];
                    hist[hi + 1] = pick;
                    if (sbValid) p.yOut[(size_t)sb * p.numSamples + t] = pick;
                }
                if (dumpNow && sbValid) {
                    const float inv = 1.0f / total;
#pragma unroll
                    for (int i = 0; i < C::RPL / 4; i++)
                        *(floatx4*)(p.p + (size_t)sb * A + sq * C::RPL + i * 4) =
                            floatx4{e[i * 4] * inv, e[i * 4 + 1] * inv, e[i * 4 + 2] * inv, e[i * 4 + 3] * inv};
                }
                wg_barrier();   // the logits image is free again; the history is written
            }
            if (t + 1 < tEnd) embed_and_send(grp, tag + 1u);
        }
    }
    __syncthreads();
    for (int i = tid; i < pp.groups * G * 16; i += C::THREADS) {
        const int b = chain * pp.groups * G * 16 + i;
        if (b < p.batch) {
            p.yInPrev[b] = hist[2 * i];
            p.yInCur[b] = hist[2 * i + 1];
        }
    }
}

template <bool F16, int R, int S, int A, bool DUMP>
__global__ __launch_bounds__((Cfg<F16, R, S, A, 1>::THREADS), 1) void wavenet_pipe(const Params p, const PipeParams pp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if constexpr (!PCfg<F16, R, S, A>::SUPPORTED) return;
    const int bidx = blockIdx.x;
    const int xcd = bidx & 7, q = bidx >> 3;
    const int stage = q % pp.stages;
    const int chain = (q / pp.stages) * 8 + xcd;
    if (chain >= pp.chains) return;
    if (stage == pp.stages - 1) pipe_head<F16, R, S, A, DUMP>(p, pp, lds, chain);
    else pipe_layers<F16, R, S, A, DUMP>(p, pp, lds, chain, stage);
}

}  // namespace wn
