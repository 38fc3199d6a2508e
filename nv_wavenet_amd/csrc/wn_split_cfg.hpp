// wn_split_cfg.hpp -- layout constants (weight streams, LDS) and the declaration of wn::wavenet_split; the kernel itself is
// in wn_split.hpp and is compiled in its own translation units (split_inst.hip), one per shape.
#pragma once

#include "wn_kernels.hpp"

namespace wn {

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int clcm(int a, int b) { return a / cgcd(a, b) * b; }
constexpr int round_up(int x, int m) { return (x + m - 1) / m * m; }
constexpr int cmax2(int a, int b) { return a > b ? a : b; }

#ifndef WN_SPLIT_PFA
#define WN_SPLIT_PFA 12      // depth bound of role A's weight prefetch ring (fragments): two layers (its queue carries the HBM loads)
#endif
#ifndef WN_SPLIT_PFB
#define WN_SPLIT_PFB 12      // ... of role B's: one layer
#endif
#ifndef WN_SPLIT_SKIPCUT
#define WN_SPLIT_SKIPCUT 4   // eighths of role B's skip GEMM issued before the h barrier of a layer
#endif

template <int R, int S, int A, int BT>
struct SCfg {
    using C = Cfg<true, R, S, A, BT>;
    static constexpr int RT = C::RT, ST = C::ST, AT = C::AT;
    static constexpr bool SUPPORTED = R == 64 && AT % 8 == 0 && ST % 4 == 0;
    static constexpr int NWR = 4;                          // waves per role
    static constexpr int HTW = C::HTW, STW = C::STW, ATW8 = AT / 8;
    static constexpr int KF_R = C::KF_R, KF_S = C::KF_S, KF_A = C::KF_A;
    static constexpr int FW_GATE = C::FW_GATE, FW_RES = C::FW_RES, FW_SKIP = C::FW_SKIP;
    // fragments per layer in the stream of a role A wave (cur | res) and of a role B wave (prev | skip)
    static constexpr int FLA = FW_GATE + FW_RES, FLB = FW_GATE + FW_SKIP;
    // (role A's ring may span several layers: a multiple of its layer length)
    static constexpr int PFA = WN_SPLIT_PFA >= FLA ? WN_SPLIT_PFA / FLA * FLA : pick_pf(FLA, WN_SPLIT_PFA);
    static constexpr int PFB = pick_pf(FLB, WN_SPLIT_PFB);
    // Role B consumes   prev(1) | prev(2) skip(0) | prev(3) skip(1) | ... | prev(0) skip(L-2) | skip(L-1) | head
    // per sample, so every code site but the first starts FW_GATE fragments into a ring turn:
    static constexpr int PHB = FW_GATE % PFB;              // ring phase of those sites
    static constexpr int BASEB = FW_GATE - PHB;            // (a whole number of ring turns)
    // head: each wave owns the output tiles w8, w8 + 8, ...; both matrices padded with zero fragments to a whole number of
    // turns of either ring (the rings come back in phase for the next sample by themselves)
    static constexpr int FW_ZS = ATW8 * KF_S, FW_ZA = ATW8 * KF_A;
    static constexpr int PADM = clcm(PFA, PFB);
    static constexpr int ZSP = round_up(FW_ZS, PADM), ZAP = round_up(FW_ZA, PADM);
    static constexpr int O_ZS = 0, O_ZA = ZSP, HEADP = ZSP + ZAP;
    static_assert(HEADP >= PFA && HEADP >= PFB, "the head part of a stream is at least one ring turn");
    // one stream per wave: [L layers][head]; waves 0..3 role A (layer stride FLA), waves 4..7 role B (FLB)
    __host__ __device__ static constexpr size_t waveStrideFrags(int L) { return (size_t)L * cmax2(FLA, FLB) + HEADP; }
    __host__ __device__ static constexpr size_t headOffA(int L) { return (size_t)L * FLA; }
    __host__ __device__ static constexpr size_t headOffB(int L) { return (size_t)L * FLB; }
    __host__ __device__ static constexpr size_t posCur(int l) { return (size_t)l * FLA; }
    __host__ __device__ static constexpr size_t posRes(int l) { return (size_t)l * FLA + FW_GATE; }
    __host__ __device__ static constexpr size_t posPrev(int l, int L) {
        return l == 1 ? 0 : (size_t)FW_GATE + (size_t)FLB * ((l == 0 ? L : l) - 2);
    }
    __host__ __device__ static constexpr size_t posSkip(int l, int L) {
        return l == L - 1 ? (size_t)FW_GATE + (size_t)FLB * (L - 1) : (size_t)FW_GATE + (size_t)FLB * l + FW_GATE;
    }
    // ---- LDS (bytes) ----
    // h, the staged taps and the staged conditioning are double-buffered by layer parity
    static constexpr int XBUF = BT * KF_R * 1024, HBUF1 = XBUF, HBUF = 2 * HBUF1, XPBUF1 = XBUF, XPBUF = 2 * XPBUF1;
    static constexpr int ACCBUF = BT * NWR * 2 * HTW * 1024;          // fp32 accumulator tiles, MFMA D layout
    static constexpr int CONDBUF1 = BT * NWR * C::COND_FR * 1024;     // conditioning of one layer: packed fragments or raw fp16 quads
    static constexpr int CONDBUF = 2 * CONDBUF1;
    static constexpr int SKBUF = BT * KF_S * 1024, ZSBUF = BT * KF_A * 1024;
    static constexpr int LROW = A + 4;
    static constexpr int LGBUF = BT * 16 * LROW * 4;                   // logits: take the place of the zs and skip images
    static constexpr int HEADBUF = cmax2(SKBUF + ZSBUF, LGBUF);
    static constexpr int YBUF = align16(BT * 16 * 4);
    static constexpr int OFF_X = 0, OFF_H = OFF_X + XBUF, OFF_XP = OFF_H + HBUF, OFF_ACC = OFF_XP + XPBUF;
    static constexpr int OFF_COND = OFF_ACC + ACCBUF;
    static constexpr int OFF_ZS = OFF_COND + CONDBUF, OFF_SK = OFF_ZS + ZSBUF, OFF_LG = OFF_ZS;
    static constexpr int OFF_Y = OFF_ZS + HEADBUF;
    static constexpr int LDS_FIXED = OFF_Y + YBUF;
    // bias table in LDS: [L][Bh 2R | Bres R] | sum of the skip biases [S] | Bzs [A] | Bza [A]
    static constexpr int BIAS_L = 3 * R;
    __host__ __device__ static size_t biasFloats(int L) { return (size_t)L * BIAS_L + S + 2 * A; }
    static size_t ldsBytes(int L, int embTables) {
        return (size_t)LDS_FIXED + biasFloats(L) * sizeof(float) + (size_t)embTables * A * R * sizeof(_Float16);
    }
    static constexpr int NPASS = (BT * 16 + 31) / 32;      // softmax passes: 32 utterances at a time (16 lanes each)
};

// EMBLDS: the current tap's embedding table lives in LDS.  RAW: see wavenet_wg.
template <int R, int S, int A, int BT, bool EMBLDS, int RAW>
__global__ __launch_bounds__(512) void wavenet_split(const Params p);

// shapes with a split_inst object in the library (Makefile: SPLIT_SHAPES)
constexpr bool split_built(int R, int S, int A) { return R == 64 && A == 256 && (S == 128 || S == 256); }

}  // namespace wn
