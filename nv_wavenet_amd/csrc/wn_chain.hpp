// wn_chain.hpp -- the MULTI-CU organisation of the engine: a chain of workgroups per utterance tile,
// every workgroup on its own CU with its share of the model RESIDENT on chip for the whole launch.
//
// Role of the reference's model-split variants (nv_wavenet_dualblock.cuh:99-161,289-293: layers block +
// skip/head block; nv_wavenet_persistent.cuh:464-568: one block per layer / output role, weights held
// in registers, blocks hand activations on through global memory).  Nothing of their structure is kept
// (thread-per-row mat-vecs, -0.0 sentinels, volatile polling): this is a CDNA4 design.
//
//   * wavenet_wg (wn_kernels.hpp) streams the whole model (1.7 MB at C3, 7.2 MB at C4) through ONE CU's
//     vector-memory path every sample; that stream, not the arithmetic, is its floor (C4: 52 us per
//     sample > the 41.7 us of 24 kHz).  A CU can hold ~460 KB of weights on chip: the 256 accumulator
//     registers of every lane (64 fragments per wave, read in place by the MFMAs once pinned there, agpr_pin),
//     a few architectural VGPRs, and ~130 KB of LDS (one ds_read_b128 per use).  So the layer stack is cut
//     into STAGES of lpc consecutive layers whose weights stay on chip (C3 fp16: 5 layers per CU, C4 fp16:
//     2), one workgroup = 4 waves per stage, plus one HEAD stage (skip ReLU -> Zs -> Za -> softmax -> pick
//     -> embedding of the next sample).  No weight is read from memory after the prologue.
//   * A sample travels down the chain: x_l (fp32, MFMA D layout, the residual stream) from stage to
//     stage, and behind it the running skip sums (fp32); the head closes the loop by handing the
//     embedded next sample to stage 0.  Arithmetic, operand rounding and summation order are exactly
//     those of wavenet_wg, so both organisations produce bit-identical samples in fp16 and fp32.
//   * While a stage waits for the sample to arrive it does everything that does not depend on it:
//     conditioning loads and the dilated-tap GEMMs  bias + Wprev x_l[t-d]  of all its layers
//     (the reference's pipelined nv_wavenet_prev, nv_wavenet.cuh:87-129), so the arrival-to-departure
//     path is  Wcur x -> gate -> Wres h  per layer only.  The skip GEMMs of a stage run after its x has
//     left (off the critical path except in the last stage).
//   * Hand-off = data-tagged 8-byte granules {value, tag = sample number in this launch + 1}, one relaxed
//     atomic store each -- agent scope (write-through to the fabric: per-XCD L2s are not coherent) or, when the
//     placement exchange at launch found producer and consumer on one XCD, workgroup scope (the line stays in
//     the L2 they share) -- swept by the consumer wave that needs them with agent-scope relaxed loads (a CU's
//     L1 is never refreshed by another CU's stores: they bypass it) until every tag matches; two sweep passes
//     are kept in flight on the x path.  No flag, no fence, no ordering between granules needed
//     (MI355X_MICROARCH.md "handoff-1to1"); measured 0.45-0.55 us from stored to received.  Mailboxes are single
//     slots: the autoregressive loop itself is the flow control (stage s cannot receive sample t+1
//     before the head has finished sample t, i.e. after every stage has consumed sample t).  Mailboxes
//     are zeroed by the host before every launch, tags count within the launch.  Every spin is bounded
//     (wall clock); a time-out raises a status word that makes every other poller of the launch give
//     up as well, and the host reports the launch as failed.
//   * Placement is a speed matter only: the stages of a chain are put on one XCD (workgroup b runs on
//     XCD b % 8 as observed on MI355X) so that granules travel through one L2.
//
// Layouts: weights, biases, conditioning, dilation ring, embeddings, selectors and outputs are those of
// wavenet_wg (wn_kernels.hpp); the chain adds only the mailboxes.
#pragma once

#include "wn_kernels.hpp"

namespace wn {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct ChainParams {
    unsigned long long* mail;   // [chain][stage] placement words (padded to 64), then [chain][stage][x granules | skip granules]; zeroed before every launch
    unsigned* status;           // [0]: 0 = fine, else the code of the first time-out (host checks after the launch)
    int stages;                 // layer stages + 1 (head)
    int lpc;                    // layers per layer stage (<= CCfg::LPC)
    int chains;                 // chains of this launch
    int tile0;                  // first tile of this launch
    int ntiles;                 // tiles of this launch (<= chains * tpc): chain c serves tiles tile0 + q * chains + c, q = 0 .. tpc-1
    int tpc;                    // tiles per chain (round 5): every stage works through its chain's tiles in turn, sample by sample
    long long timeoutTicks;     // bound of every spin, in ticks of the 100 MHz wall clock
};

constexpr int cmin(int a, int b) { return a < b ? a : b; }

template <bool F16, int R, int S, int A>
struct CCfg {
    using C = Cfg<F16, R, S, A, 1>;
    static constexpr int NW = C::NW, FLW = C::FLW, FHW = C::FHW;
    static constexpr int LDS_MAX = 160 * 1024;
    static constexpr int MAX_LPC = 8;
    // ---- layer stage with n layers.  Where a weight fragment lives decides what it costs to use:
    //   * accumulator registers (AGPRs): free -- an MFMA reads its A operand straight from them -- IF the value
    //     is pinned there (agpr_pin below).  Left to place 300 registers of weights itself, the compiler keeps
    //     them in the VGPR class, spills them to AGPRs and copies every fragment back in front of its MFMA
    //     (4 v_accvgpr_read + wait states: the cur GEMM of a C4 layer took 0.315 us for 20 MFMAs, 0.19 pinned);
    //   * architectural VGPRs: free, but the file (256) also holds the whole working set;
    //   * LDS: one ds_read_b128 per use; at 1 KiB per MFMA and wave the four SIMDs together would need all of
    //     the LDS bandwidth (256 B/clk), so LDS is for the GEMMs that run while the stage waits.
    // So the weights of the arrival-to-departure path (class A: Wcur, Wres) take the AGPRs first, then VGPRs,
    // then LDS; the weights used while the stage waits (class B: Wprev, Wskip) take what AGPRs are left,
    // then LDS, then VGPRs.
    static constexpr int FA = C::FW_GATE + C::FW_RES;      // class A fragments per layer per wave (cur | res)
    static constexpr int FB = C::FW_GATE + C::FW_SKIP;     // class B (prev | skip)
#ifndef WN_CHAIN_AGPR_FRAGS
#define WN_CHAIN_AGPR_FRAGS 64                             // fp16: the whole accumulator file (the accumulators live in VGPRs:
#endif                                                     // -amdgpu-mfma-vgpr-form); fp32 builds keep theirs in AGPRs
#ifndef WN_CHAIN_VGPR_FRAGS
#define WN_CHAIN_VGPR_FRAGS 16                             // weight fragments a wave may keep in VGPRs (64 registers)
#endif
    // (fp32 = the parity mode: its accumulators need part of the AGPR file, and it may keep more weights in the
    //  VGPR class than fit beside the working set -- the compiler then spills, which only costs time)
    static constexpr int AGPR_FRAGS = F16 ? WN_CHAIN_AGPR_FRAGS : 44, VGPR_FRAGS = F16 ? WN_CHAIN_VGPR_FRAGS : 36;
    // LDS: x image | n h images | n x (Bh, Bres) | weights
    static constexpr int fixedLds(int n) { return C::XBUF + n * C::HBUF + n * 3 * R * 4; }
    static constexpr int ldsAvail(int n) {                // weight fragments per layer per wave that fit in LDS
        int avail = (LDS_MAX - fixedLds(n)) / (n * NW * 1024);
        return avail < 0 ? 0 : avail;
    }
    struct Split {
        int aa, va, la;   // class A: AGPR / VGPR / LDS
        int ab, lb, vb;   // class B: AGPR / LDS / VGPR
        bool ok;
    };
    static constexpr Split split(int n) {
        Split s{};
        const int per = AGPR_FRAGS / n, avail = ldsAvail(n), vper = VGPR_FRAGS / n;
        s.aa = cmin(FA, per);
        s.ab = cmin(FB, per - s.aa);
        s.va = cmin(FA - s.aa, vper);
        s.la = FA - s.aa - s.va;
        s.lb = cmin(FB - s.ab, avail - s.la > 0 ? avail - s.la : 0);
        s.vb = FB - s.ab - s.lb;
        s.ok = LDS_MAX > fixedLds(n) && s.la <= avail && s.va + s.vb <= vper;
        return s;
    }
    static constexpr int pickLpc() {
        int best = 0;
        for (int n = 1; n <= MAX_LPC; n++)
            if (split(n).ok) best = n;
        return best;
    }
    static constexpr int LPC = pickLpc();                  // 0: not even one layer fits a CU (no chain for this shape)
    static constexpr bool SUPPORTED = LPC > 0;
    static constexpr int LP = LPC > 0 ? LPC : 1;
    static constexpr Split SP = split(LP);
    static constexpr int NAA = SP.aa, NVA = SP.va, NLA = SP.la;    // class A positions [0,NAA) AGPR, then VGPR, then LDS
    static constexpr int NAB = SP.ab, NLB = SP.lb, NVB = SP.vb;    // class B positions [0,NAB) AGPR, then LDS, then VGPR
    static constexpr int NAG = NAA + NAB, NVG = NVA + NVB, NLD = NLA + NLB;   // per layer per wave: AGPR [A|B], VGPR [A|B], LDS [A|B]
    static constexpr int OFF_LX = 0, OFF_LH = C::XBUF, OFF_LB = OFF_LH + LP * C::HBUF;
    static constexpr int OFF_LW = (OFF_LB + LP * 3 * R * 4 + 15) & ~15;
    static constexpr int LAYER_LDS = OFF_LW + LP * NW * NLD * 1024;
    // ---- head stage: skip image | zs image | logits | picks | biases (final skip bias, Bzs, Bza) | embeddings ----
    // the head's weights stay in registers when they fit (64 fragments at C3 / C4), else they are streamed
    // through the prefetch ring like wavenet_wg does (large A, fp32)
    static constexpr int HEADREGS = F16 ? 256 : 144;
    static constexpr int HR = FHW * 4 <= HEADREGS ? FHW : (C::FW_ZA * 4 <= HEADREGS ? C::FW_ZA : 0);
    static constexpr int HS = FHW - HR;
    static constexpr int OFF_HSK = 0, OFF_HZS = C::SKBUF;
    static constexpr int OFF_HLG = C::ALIAS_LG ? OFF_HZS : OFF_HZS + C::ZSBUF;
    static constexpr int OFF_HY = C::ALIAS_LG ? OFF_HZS + (C::ZSBUF > C::LGBUF ? C::ZSBUF : C::LGBUF) : OFF_HLG + C::LGBUF;
    static constexpr int OFF_HB = OFF_HY + C::YBUF;
    // tiles a chain may keep in flight (the head keeps their sample history in LDS: [TPC_MAX][older | current][16 utterances])
    static constexpr int TPC_MAX = 8;
    static constexpr int OFF_HH = (OFF_HB + (S + 2 * A) * 4 + 15) & ~15;
    static constexpr int OFF_HE = OFF_HH + TPC_MAX * 2 * 16 * 4;
    static size_t headLds(int embTables) { return (size_t)OFF_HE + (size_t)embTables * A * R * sizeof(typename C::P::elem); }
    static int embTables() { return headLds(2) <= (size_t)LDS_MAX ? 2 : headLds(1) <= (size_t)LDS_MAX ? 1 : 0; }
    static size_t ldsBytes() {
        const size_t h = headLds(embTables());
        return h > (size_t)LAYER_LDS ? h : (size_t)LAYER_LDS;
    }
    // mailboxes of one stage, in granules
    static constexpr int XG = R * 16, SG = S * 16;
    static constexpr size_t placeWords(int chains, int stages) { return ((size_t)chains * stages + 63) & ~(size_t)63; }
    static constexpr size_t mailGranules(int chains, int stages, int tpc = 1) { return placeWords(chains, stages) + (size_t)chains * stages * tpc * (XG + SG); }
};

// Experiment build (-DWN_CHAIN_TIMING): wave 0 of every stage stamps the 100 MHz wall clock (one counter for
// the whole chip) at its phase boundaries for samples 8..15 of the launch into p.p, read as
// unsigned long long [stage][tile of the chain q][8 samples][16 events] (chain 0 only; scripts/chain_phase.py).
#ifdef WN_CHAIN_TIMING
#define WN_CT_DECL unsigned long long cts[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define WN_CT(ev) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); cts[ev] = __builtin_amdgcn_s_memrealtime(); }
#define WN_CT_FLUSH(stageIdx, tt)                                                                           \
    if (w == 0 && lane == 0 && chainIdx == 0 && (tt) >= 8 && (tt) < 16) {                                     \
        unsigned long long* dbg = (unsigned long long*)p.p + (((size_t)(stageIdx) * nq + q) * 8 + ((tt) - 8)) * 16; \
        _Pragma("unroll") for (int q_ = 0; q_ < 16; q_++) dbg[q_] = cts[q_];                                 \
    }
#else
#define WN_CT_DECL
#define WN_CT(ev) {}
#define WN_CT_FLUSH(stageIdx, tt) {}
#endif

// ---- hand-off primitives -----------------------------------------------------------------------
constexpr long long kChainTimeoutTicks = 150000000LL;     // default bound: 1.5 s of the 100 MHz wall clock

struct Spin {
    gu32* status;
    long long t0;
    unsigned spins;
    long long limit;
};
// false: give up (this wave timed out, or another one did)
WN_DEV bool spin_more(Spin& s, unsigned code) {
    if ((++s.spins & 127u) == 0u) {
        if (__hip_atomic_load(s.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if ((long long)wall_clock64() - s.t0 > s.limit) {
            __hip_atomic_store(s.status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
#ifndef WN_CHAIN_SLEEP
#define WN_CHAIN_SLEEP 1
#endif
    if (WN_CHAIN_SLEEP > 0) __builtin_amdgcn_s_sleep(WN_CHAIN_SLEEP);
    return true;
}

// tile `tile`, register r of lane `lane`: granule index inside a mailbox (lanes contiguous: one 512-byte
// coalesced access per wave instruction)
WN_DEV int granule_at(int tile, int r, int lane) { return (tile * 4 + r) * 64 + lane; }

// this wave's tiles w, w+NW, ... of a vector in MFMA D layout -> the consumer's mailbox.
// sameXcd: the consumer was found on this XCD (chain_place below).  Then the granules are stored at workgroup
// scope: they go through this CU's write-through L1 into the L2 both CUs share and STAY there, so the
// consumer's L1-bypassing sweep is served by L2; an agent-scope (sc1) store writes through to the fabric and
// drops the line from L2, and every sweep pass of the consumer goes out to memory
// (MI355X_MICROARCH.md, "stores of each flavour").  Across XCDs the agent-scope store is the only valid form.
template <int NT, int NW, int SCOPE>
WN_DEV void send_tiles_scope(gu64* g, int w, int lane, unsigned tag, const floatx4 (&v)[NT]) {
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // (through a scalar: __builtin_bit_cast applied to the vector-element lvalue v[i][r] itself reads element 0)
            const float f = v[i][r];
            __hip_atomic_store(g + granule_at(w + NW * i, r, lane), ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(f),
                               __ATOMIC_RELAXED, SCOPE);
        }
}
template <int NT, int NW>
WN_DEV void send_tiles(unsigned long long* mbox, int w, int lane, unsigned tag, const floatx4 (&v)[NT], bool sameXcd) {
    if (sameXcd) send_tiles_scope<NT, NW, __HIP_MEMORY_SCOPE_WORKGROUP>((gu64*)mbox, w, lane, tag, v);
    else send_tiles_scope<NT, NW, __HIP_MEMORY_SCOPE_AGENT>((gu64*)mbox, w, lane, tag, v);
}
// one sweep pass: issue the loads of this wave's granules (agent scope: L1 is bypassed)
template <int NT, int NW>
WN_DEV void sweep_issue(const unsigned long long* mbox, int w, int lane, unsigned long long (&q)[NT * 4]) {
    gu64* g = (gu64*)mbox;
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            q[i * 4 + r] = __hip_atomic_load(g + granule_at(w + NW * i, r, lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and its check: true when every granule of the wave carries `tag` (then v holds the payload)
template <int NT>
WN_DEV bool sweep_check(const unsigned long long (&q)[NT * 4], unsigned tag, floatx4 (&v)[NT]) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[i][r] = __uint_as_float((unsigned)q[i * 4 + r]);
            ok &= (unsigned)(q[i * 4 + r] >> 32) == tag;
        }
    return __all(ok);
}
// the same tiles, swept until every granule carries `tag`
template <int NT, int NW>
WN_DEV bool recv_tiles(const unsigned long long* mbox, int w, int lane, unsigned tag, floatx4 (&v)[NT], gu32* status,
                       unsigned code, long long limit) {
    Spin s{status, (long long)wall_clock64(), 0u, limit};
    for (;;) {
        unsigned long long q[NT * 4];
        sweep_issue<NT, NW>(mbox, w, lane, q);
        if (sweep_check<NT>(q, tag, v)) return true;
        if (!spin_more(s, code)) return false;
    }
}
// Latency-critical variant (the x hand-off): two passes in flight, a new one issued every WN_CHAIN_GAP sleeps,
// so the time from the granules landing to their detection is a fraction of a memory round trip instead of
// half a round trip on average.
#ifndef WN_CHAIN_GAP
#define WN_CHAIN_GAP 3
#endif
template <int NT, int NW>
WN_DEV bool recv_tiles_fast(const unsigned long long* mbox, int w, int lane, unsigned tag, floatx4 (&v)[NT], gu32* status,
                            unsigned code, long long limit) {
    Spin s{status, (long long)wall_clock64(), 0u, limit};
    unsigned long long qa[NT * 4], qb[NT * 4];
    sweep_issue<NT, NW>(mbox, w, lane, qa);
    for (;;) {
        __builtin_amdgcn_s_sleep(WN_CHAIN_GAP);
        sweep_issue<NT, NW>(mbox, w, lane, qb);
        if (sweep_check<NT>(qa, tag, v)) return true;
        __builtin_amdgcn_s_sleep(WN_CHAIN_GAP);
        sweep_issue<NT, NW>(mbox, w, lane, qa);
        if (sweep_check<NT>(qb, tag, v)) return true;
        if (!spin_more(s, code)) return false;
    }
}

// Placement exchange at the start of a launch: every stage publishes the XCD it runs on and reads its
// consumer's.  A speed matter only: the answer selects the store flavour of send_tiles, both are valid.
WN_DEV unsigned my_xcd() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu; }   // HW_REG_XCC_ID[3:0]
WN_DEV bool chain_place(unsigned long long* place, int mine, int consumer, gu32* status, bool& sameXcd, long long limit) {
    gu64* g = (gu64*)place;
    const unsigned xcd = my_xcd();
    if (threadIdx.x == 0) __hip_atomic_store(g + mine, 0xC0DE00000000ull | xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Spin s{status, (long long)wall_clock64(), 0u, limit};
    for (;;) {
        const unsigned long long c = __hip_atomic_load(g + consumer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((c >> 32) == 0xC0DEull) {
            sameXcd = (unsigned)(c & 0xfu) == xcd;
            return true;
        }
        if (!spin_more(s, 0x400u + (unsigned)mine)) return false;
    }
}

// ---- weight fragments pinned in accumulator registers: agpr_pin (wn_kernels.hpp) ----------------------------

// acc[mt] += W(tile slot mt) * b  with AGPR-pinned fragments wres[pos0 ...] (the head's resident weights)
template <bool F16, int MT, int KF, int NFR>
WN_DEV void gemm_pinned(const floatx4 (&wres)[NFR], int pos0, floatx4 (&acc)[MT], const typename Prec<F16>::frag (&b)[KF]) {
    using frag = typename Prec<F16>::frag;
    constexpr int G = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / G; mg++)
#pragma unroll
        for (int kf = 0; kf < KF; kf++)
#pragma unroll
            for (int mi = 0; mi < G; mi++)
                acc[mg * G + mi] = mma(__builtin_bit_cast(frag, wres[pos0 + (mg * KF + kf) * G + mi]), b[kf], acc[mg * G + mi]);
}

// acc[mt] += W(tile slot mt) * b  with the stage's resident weights (fragment order of gemm(), wn_kernels.hpp).
// CLS 0 = class A (cur: POS0 = 0, res: POS0 = FW_GATE), CLS 1 = class B (prev: POS0 = 0, skip: POS0 = FW_GATE).
// A fragment at class position pos lives, in this order,  A: AGPR [0,NAA) | VGPR | LDS   B: AGPR | LDS | VGPR.
template <bool F16, typename CC, int CLS, int POS0, int MT, int KF>
WN_DEV void gemm_w(const floatx4 (&wag)[CC::NAG ? CC::NAG : 1], const typename Prec<F16>::frag (&wvg)[CC::NVG ? CC::NVG : 1],
                   const char* wl, unsigned laneOff, floatx4 (&acc)[MT], const typename Prec<F16>::frag (&b)[KF]) {
    using frag = typename Prec<F16>::frag;
    constexpr int G = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / G; mg++)
#pragma unroll
        for (int kf = 0; kf < KF; kf++)
#pragma unroll
            for (int mi = 0; mi < G; mi++) {
                const int pos = POS0 + (mg * KF + kf) * G + mi;
                frag a;
                if (CLS == 0) {
                    if (pos < CC::NAA) a = __builtin_bit_cast(frag, wag[pos < CC::NAA ? pos : 0]);
                    else if (pos < CC::NAA + CC::NVA) a = wvg[pos < CC::NAA + CC::NVA ? pos - CC::NAA : 0];
                    else a = *(const frag*)(wl + (size_t)(pos - CC::NAA - CC::NVA) * 1024 + laneOff);
                } else {
                    if (pos < CC::NAB) a = __builtin_bit_cast(frag, wag[pos < CC::NAB ? CC::NAA + pos : 0]);
                    else if (pos < CC::NAB + CC::NLB) a = *(const frag*)(wl + (size_t)(CC::NLA + pos - CC::NAB) * 1024 + laneOff);
                    else a = wvg[pos >= CC::NAB + CC::NLB ? CC::NVA + pos - CC::NAB - CC::NLB : 0];
                }
                acc[mg * G + mi] = mma(a, b[kf], acc[mg * G + mi]);
            }
}

// ------------------------------------------------------------------------------------------------
// layer stage: layers l0 .. l0+nl-1 of tile `tile`
// ------------------------------------------------------------------------------------------------
// HOIST (round 6; launches with several tiles per chain): the packed conditioning of ALL own layers is requested in front of the
// first layer's use instead of layer by layer -- the HBM part of a unit's "idle work", which a saturated stage pays once per tile
// and sample (measured in round 5 as an experiment build at C4: five tiles per chain 24.0 -> 25.0 kHz, one and four tiles unchanged,
// but the registers it holds across the idle work cost the one-tile launches 4 % -- C2 B = 4 84.1 -> 80.5 kHz --: hence a separate
// instantiation, launched only when a chain serves more than four tiles: round 6, A/B in one GPU call, steady-state kHz per utterance at C4
// with / without: 4 tiles 27.5 / 27.7, 5 tiles 24.4 - 24.7 / 23.5 - 23.8 -- 1 280 utterances per GPU in real time --, 6 tiles 20.8 / 20.0).
template <bool F16, int R, int S, int A, bool DUMP, bool HOIST>
WN_DEV void chain_layers(const Params& p, const ChainParams& cp, char* lds, int chainIdx, int stage) {
    using CC = CCfg<F16, R, S, A>;
    using C = typename CC::C;
    using P = Prec<F16>;
    using frag = typename P::frag;
    using elem = typename P::elem;
    constexpr int LP = CC::LP, NLD = CC::NLD, NW = C::NW, FLW = C::FLW;
    constexpr int RT = C::RT, HTW = C::HTW, STW = C::STW, KF_R = C::KF_R;

    char* const xbuf = lds + CC::OFF_LX;
    char* const hbuf = lds + CC::OFF_LH;
    float* const biasLds = (float*)(lds + CC::OFF_LB);          // [LP][Bh 2R | Bres R]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int L = p.numLayers;
    const int l0 = stage * cp.lpc;
    const int nl = cmin(cp.lpc, L - l0);
    const bool lastLayerStage = stage == cp.stages - 2;
    const unsigned laneOff = (unsigned)lane * 16u;
    char* const wlds = lds + CC::OFF_LW + (size_t)w * LP * NLD * 1024;     // this wave's slice

    // Several tiles per chain (round 5): the stage works through its chain's tiles q = 0 .. nq-1 in turn, sample by sample -- tile
    // q's sample t+1 cannot arrive before the head has finished its sample t, a whole trip round the chain later, and the stage
    // serves the other tiles meanwhile.  Every tile has its own single-slot mailboxes; the weights are resident once.
    const int nq = (cp.ntiles - chainIdx + cp.chains - 1) / cp.chains;
    unsigned long long* const boxes = cp.mail + CC::placeWords(cp.chains, cp.stages);
    unsigned long long* const mstage = boxes + ((size_t)chainIdx * cp.stages + stage) * cp.tpc * (CC::XG + CC::SG);
    gu32* const status = (gu32*)cp.status;
    bool sameXcd = false;
    if (!chain_place(cp.mail, chainIdx * cp.stages + stage, chainIdx * cp.stages + stage + 1, status, sameXcd, cp.timeoutTicks)) return;

    // ---- gate and residual biases of the own layers -> LDS (the skip biases are added by the head) ----
    for (int i = tid; i < nl * 3 * R; i += C::THREADS)
        biasLds[i] = p.bias[(size_t)(l0 + i / (3 * R)) * C::BIAS_L + i % (3 * R)];

    // ---- resident weights (per layer stream: prev | cur | res | skip) ---------------------------------
    const char* const wbase = (const char*)p.wblob + (size_t)w * C::waveStreamFrags(L) * 1024;
    floatx4 wag[LP][CC::NAG ? CC::NAG : 1];   // AGPRs: [class A | class B]
    frag wvg[LP][CC::NVG ? CC::NVG : 1];      // VGPRs: [class A | class B]
#pragma unroll
    for (int li = 0; li < LP; li++) {
        const int lw = l0 + (li < nl ? li : 0);            // (the stream is in wavenet_wg's consumption order: Cfg::streamPos)
        char* const myl = wlds + (size_t)li * NLD * 1024;
#pragma unroll
        for (int a = 0; a < CC::FA; a++) {                 // class A: cur | res
            const frag f = *(const frag*)(wbase + C::streamPos(lw, C::O_CUR + a, L) * 1024 + laneOff);
            if (a < CC::NAA) wag[li][a < CC::NAA ? a : 0] = agpr_pin(__builtin_bit_cast(floatx4, f));
            else if (a < CC::NAA + CC::NVA) wvg[li][a < CC::NAA + CC::NVA ? a - CC::NAA : 0] = f;
            else if (li < nl) *(frag*)(myl + (size_t)(a - CC::NAA - CC::NVA) * 1024 + laneOff) = f;
        }
#pragma unroll
        for (int bq = 0; bq < CC::FB; bq++) {              // class B: prev, then skip
            const int idx = bq < C::FW_GATE ? C::O_PREV + bq : C::O_SKIP + (bq - C::FW_GATE);
            const frag f = *(const frag*)(wbase + C::streamPos(lw, idx, L) * 1024 + laneOff);
            if (bq < CC::NAB) wag[li][bq < CC::NAB ? CC::NAA + bq : 0] = agpr_pin(__builtin_bit_cast(floatx4, f));
            else if (bq < CC::NAB + CC::NLB) {
                if (li < nl) *(frag*)(myl + (size_t)(CC::NLA + bq - CC::NAB) * 1024 + laneOff) = f;
            } else wvg[li][bq >= CC::NAB + CC::NLB ? CC::NVA + bq - CC::NAB - CC::NLB : 0] = f;
        }
    }

    // dilation and first ring slot of the own layers
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
    const size_t condStride = (size_t)p.tiles * NW * C::COND_FR * 1024;             // one (sample, layer) row
    const size_t ringTile = (size_t)p.ringSlots * KF_R * 1024;

    frag selA[P::TPF];
#pragma unroll
    for (int tt = 0; tt < P::TPF; tt++)
#pragma unroll
        for (int e = 0; e < P::EPL; e++) selA[tt][e] = (elem)(((e >> 2) == tt && g * 4 + (e & 3) == j) ? 1.0f : 0.0f);

    const int tEnd = p.initSample + p.count;
    for (int t = p.initSample; t < tEnd; t++)
    for (int q = 0; q < nq; q++) {
        const unsigned tag = (unsigned)(t - p.initSample) + 1u;
        const bool dumpNow = DUMP && p.dump && (t == tEnd - 1);
        // this unit's tile: utterances, mailboxes (own inputs; the next stage's inputs), conditioning, ring
        const int tile = cp.tile0 + q * cp.chains + chainIdx;
        const int b = tile * 16 + j;
        const bool uvalid = b < p.batch;
        const int ub = uvalid ? b : p.batch - 1;
        unsigned long long* const mbase = mstage + (size_t)q * (CC::XG + CC::SG);
        const unsigned long long* const xin = mbase;
        const unsigned long long* const skin = mbase + CC::XG;
        unsigned long long* const xout = mbase + (size_t)cp.tpc * (CC::XG + CC::SG);
        unsigned long long* const skout = xout + CC::XG;
        const char* const condMine = (const char*)p.cond + ((size_t)tile * NW + w) * C::COND_FR * 1024;
        char* const ringMine = (char*)p.ring + (size_t)tile * ringTile;
        // every ring store of the previous unit has completed (and is visible to the whole workgroup:
        // one L1 per CU); also orders the reuse of the h images and of the bias table after the prologue
        // (round 5, measured and dropped at C4 with 4-6 tiles per chain: draining one unit later, in front of a unit's own ring stores:
        //  no difference; requesting the next unit's conditioning and taps one unit ahead, behind the x hand-off -- the idle work is
        //  2.6-4 us of a saturated stage's 8 us per unit -- needs 48 more registers: 24 spilled, 27.3 -> 19.7 kHz at 1024 utterances)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
        WN_CT_DECL
        WN_CT(0)

        // ---- while the sample is on its way: conditioning + dilated-tap GEMMs of all own layers ------
        floatx4 acc[LP][2 * HTW];
        frag cdAll[HOIST ? LP : 1][C::COND_FR];
        if constexpr (HOIST) {
            if (p.condRawKind == 0) {
#pragma unroll
                for (int li = 0; li < LP; li++)
                    if (li < nl) {
                        const char* cp0 = condMine + ((size_t)t * L + (l0 + li)) * condStride;
#pragma unroll
                        for (int k = 0; k < C::COND_FR; k++) cdAll[li][k] = *(const frag*)(cp0 + k * 1024 + laneOff);
                    }
            }
        }
#pragma unroll
        for (int li = 0; li < LP; li++) {
            if (li < nl) {
                frag cd[1][C::COND_FR];
                const int l = l0 + li;
                const float* bl = biasLds + li * 3 * R;
                const char* cp0 = condMine + ((size_t)t * L + l) * condStride;
                if (p.condRawKind != 0) {
                    // the caller's [sample][L][maxBatch][2R] tensor read in place: 4 channels of this lane's utterance per gate
                    // tile, scaled and rounded like pack_cond_tiled_kernel would have (bit-identical to packed runs)
                    const int tc = t < p.condSamples ? t : p.condSamples - 1;
                    const size_t at = (((size_t)tc * L + l) * p.maxBatch + ub) * (2 * R) + g * 4;
                    auto slotAt = [&](int it) { return (w + NW * (it >> 1) + (it & 1) * RT) * 16; };
                    if (!F16 || p.condRawKind == 1) {
                        const float* rb = (const float*)p.condRaw + at;
                        frag raw[(F16 ? 2 : 1) * C::COND_FR];
#pragma unroll
                        for (int it = 0; it < 2 * HTW; it++) raw[it] = __builtin_bit_cast(frag, *(const floatx4*)(rb + slotAt(it)));
#pragma unroll
                        for (int k = 0; k < C::COND_FR; k++) cd[0][k] = cond_frag<F16, 1>(raw, k);
                    } else if constexpr (F16) {
                        const _Float16* rb = (const _Float16*)p.condRaw + at;
                        frag raw[C::COND_FR];
#pragma unroll
                        for (int k = 0; k < C::COND_FR; k++) {
                            const half4 qa = *(const half4*)(rb + slotAt(2 * k)), qb = *(const half4*)(rb + slotAt(2 * k + 1));
                            raw[k] = half8{qa[0], qa[1], qa[2], qa[3], qb[0], qb[1], qb[2], qb[3]};
                        }
#pragma unroll
                        for (int k = 0; k < C::COND_FR; k++) cd[0][k] = cond_frag<F16, 2>(raw, k);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < C::COND_FR; k++) {
                        if constexpr (HOIST) cd[0][k] = cdAll[li][k];
                        else cd[0][k] = *(const frag*)(cp0 + k * 1024 + laneOff);
                    }
                }
#pragma unroll
                for (int i = 0; i < HTW; i++) {
                    acc[li][2 * i] = *(const floatx4*)(bl + (w + NW * i) * 16 + g * 4);
                    acc[li][2 * i + 1] = *(const floatx4*)(bl + (w + NW * i + RT) * 16 + g * 4);
                }
                // + conditioning (a B-layout fragment = the D layout of TPF result tiles: wn::cond_add, shared with wavenet_wg)
#pragma unroll
                for (int k = 0; k < C::COND_FR; k++) cond_add<F16>(&acc[li][k * P::TPF], cd[0][k], selA);
                const int d = dl[li].d;
                const bool havePrev = t >= d;
                frag xp[KF_R];
                const char* rp = ringMine + (size_t)(unsigned)(dl[li].off + (t & (d - 1))) * (KF_R * 1024);
#pragma unroll
                for (int k = 0; k < KF_R; k++) {
                    if (havePrev) xp[k] = *(const frag*)(rp + (size_t)k * 1024 + laneOff);
                    else {
#pragma unroll
                        for (int e = 0; e < P::EPL; e++) xp[k][e] = (elem)0.f;
                    }
                }
                gemm_w<F16, CC, 1, 0, 2 * HTW, KF_R>(wag[li], wvg[li], wlds + (size_t)li * NLD * 1024, laneOff, acc[li], xp);
            }
        }

        // ---- the sample arrives: x_l0[t], this wave's tiles (fp32) ----------------------------------
        floatx4 x[HTW];
        frag xring[LP][C::XPW];
        WN_CT(1)
        if (!recv_tiles_fast<HTW, NW>(xin, w, lane, tag, x, status, 0x100u + (unsigned)stage, cp.timeoutTicks)) return;
        unsigned long long skq[STW * 4];
        WN_CT(2)
#pragma unroll
        for (int i = 0; i < HTW; i++) lds_put_tile<F16>(xbuf, w + NW * i, lane, x[i]);
        wg_barrier();

#pragma unroll
        for (int li = 0; li < LP; li++) {
            if (li < nl) {
                const int l = l0 + li;
                const float* bl = biasLds + li * 3 * R;
                const char* wl = wlds + (size_t)li * NLD * 1024;
                char* const hb_img = hbuf + li * C::HBUF;
                frag xb[KF_R];
                lds_get_frags<F16, KF_R>(xbuf, lane, xb);
                if (li == 0) WN_CT(8)
                // x_l[t] will replace x_l[t-d] in the ring (each wave stores its share of the fragments): kept here, stored behind
                // the x hand-off (round 5: nothing that can wait is queued in front of the stores of the hand-off)
#pragma unroll
                for (int i = 0; i < C::XPW; i++) {
                    const int k = w + NW * i;              // wave-uniform: no select over the xb registers
                    if (k < KF_R) xring[li][i] = *(const frag*)(xbuf + (size_t)k * 1024 + laneOff);
                }
                gemm_w<F16, CC, 0, 0, 2 * HTW, KF_R>(wag[li], wvg[li], wl, laneOff, acc[li], xb);
                if (li == 0) {
#ifdef WN_CHAIN_TIMING
                    asm volatile("s_nop 0" ::"v"(acc[li][0][0]), "v"(acc[li][1][0]));
#endif
                    WN_CT(9)
                }
#pragma unroll
                for (int i = 0; i < HTW; i++) {
                    const floatx4 hv = gate4<F16>(acc[li][2 * i], acc[li][2 * i + 1]);
                    lds_put_tile<F16>(hb_img, w + NW * i, lane, hv);
                }
                if (li == 0) WN_CT(10)
                wg_barrier();   // h complete
                if (li == 0) WN_CT(11)
                frag hb[KF_R];
                lds_get_frags<F16, KF_R>(hb_img, lane, hb);
                if (li == 0) WN_CT(12)
                // the skip sums of the stage before are usually on their way by now; their first sweep pass is issued right
                // behind the x hand-off
                // (round 5: NOT before the x hand-off any more -- vector-memory instructions issue in order, and the sixteen L1-bypassing
                //  loads of this sweep in front of the x stores cost every stage 0.2 us of its arrival-to-departure path: 25.6 -> 26.8 kHz
                //  at C4, and 23.6 -> 26.3 with four tiles per chain, where the mailboxes no longer sit in a quiet L2)
                floatx4 xa[HTW];
#pragma unroll
                for (int i = 0; i < HTW; i++) xa[i] = *(const floatx4*)(bl + 2 * R + (w + NW * i) * 16 + g * 4) + x[i];
                gemm_w<F16, CC, 0, C::FW_GATE, HTW, KF_R>(wag[li], wvg[li], wl, laneOff, xa, hb);
#pragma unroll
                for (int i = 0; i < HTW; i++) x[i] = xa[i];
                if (li == 0) {
#ifdef WN_CHAIN_TIMING
                    asm volatile("s_nop 0" ::"v"(x[0][0]));
#endif
                    WN_CT(13)
                }
                if (dumpNow && uvalid) {
#pragma unroll
                    for (int i = 0; i < HTW; i++)
                        *(floatx4*)(p.xtOut + ((size_t)l * p.maxBatch + ub) * R + (w + NW * i) * 16 + g * 4) = x[i];
                }
                if (li + 1 < nl) {   // the next own layer reads x through LDS; the last one hands it on
#pragma unroll
                    for (int i = 0; i < HTW; i++) lds_put_tile<F16>(xbuf, w + NW * i, lane, x[i]);
                    wg_barrier();   // x complete
                    if (li == 0) WN_CT(14)
                }
            }
        }
        WN_CT(3)
        if (!lastLayerStage) send_tiles<HTW, NW>(xout, w, lane, tag, x, sameXcd);   // (the last layer's output is unused)
        if (stage != 0) sweep_issue<STW, NW>(skin, w, lane, skq);
#pragma unroll
        for (int li = 0; li < LP; li++) {
            if (li < nl) {
                const int d = dl[li].d;
                char* rp = ringMine + (size_t)(unsigned)(dl[li].off + (t & (d - 1))) * (KF_R * 1024);
#pragma unroll
                for (int i = 0; i < C::XPW; i++) {
                    const int k = w + NW * i;
                    if (k < KF_R) *(frag*)(rp + (size_t)k * 1024 + laneOff) = xring[li][i];
                }
            }
        }
        WN_CT(4)

        // ---- behind the sample: running skip sums  skip <- Wskip_l h_l + skip  ------------------------
        // fp16 engine: onto the sums received from the stage before, in layer order -- the order of wavenet_wg's accumulators, which keeps
        // the organisations bit-identical (tested).  fp32 engine (round 6): the own layers' contribution is summed FIRST, from zero, while
        // the sums of the stages before are still on their way, and added to them when they arrive -- the skip sums then trail x by one
        // stage's skip GEMM instead of accumulating every stage's behind the last layer: 8.9 -> 5.2 us from the last stage's x to its
        // skip hand-off at C3 (scripts/chain_phase.py), 23.2 -> 25 kHz through the reference's PyTorch entry.  (fp32 sums in another
        // association: samples exact against the oracle like before, the dumped skipOut within its bar.)
        constexpr bool OWN_FIRST = !F16;
        floatx4 sk[STW];
        bool haveIn = stage == 0;
        auto receive_sums = [&]() -> bool {
            if (!haveIn && !sweep_check<STW>(skq, tag, sk)) {
                if (!recv_tiles<STW, NW>(skin, w, lane, tag, sk, status, 0x200u + (unsigned)stage, cp.timeoutTicks)) return false;
            }
            haveIn = true;
            return true;
        };
        if (stage == 0) {
#pragma unroll
            for (int i = 0; i < STW; i++) sk[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        } else if (!OWN_FIRST || dumpNow) {      // (the per-layer dump needs the incoming sums under every layer)
            if (!receive_sums()) return;
        }
        WN_CT(5)
        floatx4 own[OWN_FIRST ? STW : 1];
        if constexpr (OWN_FIRST) {
#pragma unroll
            for (int i = 0; i < STW; i++) own[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int li = 0; li < LP; li++) {
            if (li < nl) {
                frag hb[KF_R];
                lds_get_frags<F16, KF_R>(hbuf + li * C::HBUF, lane, hb);
                if constexpr (OWN_FIRST) gemm_w<F16, CC, 1, C::FW_GATE, STW, KF_R>(wag[li], wvg[li], wlds + (size_t)li * NLD * 1024, laneOff, own, hb);
                else gemm_w<F16, CC, 1, C::FW_GATE, STW, KF_R>(wag[li], wvg[li], wlds + (size_t)li * NLD * 1024, laneOff, sk, hb);
                // (the last layer's skipOut is dumped by the head, after the ReLU)
                if (dumpNow && uvalid && l0 + li < L - 1) {
#pragma unroll
                    for (int i = 0; i < STW; i++) {
                        // running sum of the skip biases up to this layer, in layer order like wavenet_wg
                        const int row = (w + NW * i) * 16 + g * 4;
                        floatx4 run = *(const floatx4*)(p.bias + 3 * R + row);
                        for (int l = 1; l <= l0 + li; l++) run += *(const floatx4*)(p.bias + (size_t)l * C::BIAS_L + 3 * R + row);
                        floatx4 v = sk[i];
                        if constexpr (OWN_FIRST) v += own[i];
                        *(floatx4*)(p.skipOut + ((size_t)(l0 + li) * p.maxBatch + ub) * S + row) = v + run;
                    }
                }
            }
        }
        if constexpr (OWN_FIRST) {
            if (!receive_sums()) return;
#pragma unroll
            for (int i = 0; i < STW; i++) sk[i] += own[i];
        }
        WN_CT(6)
        send_tiles<STW, NW>(skout, w, lane, tag, sk, sameXcd);
        WN_CT(7)
        WN_CT_FLUSH(stage, t - p.initSample)
    }
}

// ------------------------------------------------------------------------------------------------
// head stage: final skip -> Zs -> Za -> softmax -> pick -> embedding of the next sample
// ------------------------------------------------------------------------------------------------
template <bool F16, int R, int S, int A, bool DUMP>
WN_DEV void chain_head(const Params& p, const ChainParams& cp, char* lds, int chainIdx) {
    using CC = CCfg<F16, R, S, A>;
    using C = typename CC::C;
    using P = Prec<F16>;
    using frag = typename P::frag;
    using quad = typename P::quad;
    using elem = typename P::elem;
    constexpr int NW = C::NW, PF = C::PF;
    constexpr int HTW = C::HTW, STW = C::STW, ATW = C::ATW;
    constexpr int KF_S = C::KF_S, KF_A = C::KF_A;
    constexpr int HR = CC::HR, HS = CC::HS;
    constexpr int HSP = HS == 0 ? 0 : (HS == C::FW_ZS ? C::FW_ZS + C::PAD1 : C::FHWP);   // physical length of the streamed part (wn_kernels.hpp: zs | PAD1 | za | PAD2)

    char* const skbuf = lds + CC::OFF_HSK;
    char* const zsbuf = lds + CC::OFF_HZS;
    float* const lgbuf = (float*)(lds + CC::OFF_HLG);
    int* const ybuf = (int*)(lds + CC::OFF_HY);
    float* const fsb = (float*)(lds + CC::OFF_HB);           // final running skip bias [S], then Bzs[A], Bza[A]
    float* const headBias = fsb + S;
    elem* const embLds = (elem*)(lds + CC::OFF_HE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int L = p.numLayers;
    const int stage = cp.stages - 1;
    const unsigned laneOff = (unsigned)lane * 16u;

    const int su = tid / C::LPU, sq = tid % C::LPU;            // softmax role

    const int nq = (cp.ntiles - chainIdx + cp.chains - 1) / cp.chains;      // tiles of this chain (see chain_layers)
    unsigned long long* const boxes = cp.mail + CC::placeWords(cp.chains, cp.stages);
    unsigned long long* const mstage = boxes + ((size_t)chainIdx * cp.stages + stage) * cp.tpc * (CC::XG + CC::SG);
    unsigned long long* const mstage0 = boxes + ((size_t)chainIdx * cp.stages) * cp.tpc * (CC::XG + CC::SG);   // stage 0
    gu32* const status = (gu32*)cp.status;
    // per-tile state of the head between two visits of a tile: the sample history
    int* const hist = (int*)(lds + CC::OFF_HH);                                              // [TPC_MAX][older | current][16]
    bool sameXcd = false;
    if (!chain_place(cp.mail, chainIdx * cp.stages + stage, chainIdx * cp.stages, status, sameXcd, cp.timeoutTicks)) return;

    // ---- biases: sum of all skip biases (layer order, like wavenet_wg's running sums), Bzs, Bza ----
    for (int s0 = tid; s0 < S; s0 += C::THREADS) {
        float run = p.bias[3 * R + s0];
        for (int l = 1; l < L; l++) run += p.bias[(size_t)l * C::BIAS_L + 3 * R + s0];
        fsb[s0] = run;
    }
    for (int i = tid; i < 2 * A; i += C::THREADS) headBias[i] = p.bias[(size_t)L * C::BIAS_L + i];
    // ---- embedding tables -> LDS (p.embLds = 2: both, 1: the current tap's only) ------------------
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
        const size_t off = (size_t)y * R + tile16 * 16 + g * 4;
        if (nEmb > 0) return quad_to_f32(*(const quad*)(embLds + off));
        return quad_to_f32(*(const quad*)(gEmbCur + off));
    };
    auto rowPrev = [&](int y, int tile16) -> floatx4 {
        const size_t off = (size_t)y * R + tile16 * 16 + g * 4;
        if (nEmb > 1) return quad_to_f32(*(const quad*)(embLds + (size_t)A * R + off));
        return quad_to_f32(*(const quad*)(gEmbPrev + off));
    };

    // ---- head weights: resident fragments, prefetch ring for the streamed part ---------------------
    const char* const wbase = (const char*)p.wblob + (size_t)w * C::waveStreamFrags(L) * 1024;
    const char* const whead = wbase + C::headOffsetFrags(L) * 1024;
    // resident head fragments: pinned in the accumulator file, read in place by the MFMAs (agpr_pin)
    floatx4 hw[HR ? HR : 1];
    if constexpr (HR > 0) {
#pragma unroll
        for (int i = 0; i < HR; i++)
            hw[i] = agpr_pin(__builtin_bit_cast(floatx4, *(const frag*)(whead + (size_t)C::headFrag(HS + i) * 1024 + laneOff)));
    }
    WStream<F16, PF> ws;
    if constexpr (HS > 0) {
        static_assert(HSP >= PF, "streamed head shorter than the prefetch ring");
#pragma unroll
        for (int i = 0; i < PF; i++) ws.buf[i] = *(const frag*)(whead + (size_t)i * 1024 + laneOff);
    }

    if (w == 0 && g == 0) {
        for (int q = 0; q < nq; q++) {
            int b0 = (cp.tile0 + q * cp.chains + chainIdx) * 16 + j;
            b0 = b0 < p.batch ? b0 : p.batch - 1;
            hist[(q * 2 + 0) * 16 + j] = p.yInPrev[b0];
            hist[(q * 2 + 1) * 16 + j] = p.yInCur[b0];
        }
    }
    __syncthreads();   // tables, biases and history complete

    // embedding of the sample after (yPrev, yCur) -> stage 0 (nv_wavenet_reference.cpp:42-56)
    // ep: the older tap's row of the NEXT sample is the current tap's index of this one: gathered a whole
    // sample early (it may come from global memory when only one table fits in LDS)
    floatx4 ep[HTW];
    auto embed_and_send = [&](int q, unsigned tag, int yCur) {
        floatx4 x0[HTW];
#pragma unroll
        for (int i = 0; i < HTW; i++) {
            const int tile16 = w + NW * i;
            floatx4 v = ep[i] + rowCur(yCur, tile16);
            if (p.tanhEmbed) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = tanh_t<F16>(v[r]);
            }
            x0[i] = v;
        }
        send_tiles<HTW, NW>(mstage0 + (size_t)q * (CC::XG + CC::SG), w, lane, tag, x0, sameXcd);
    };
    for (int q = 0; q < nq; q++) {
#pragma unroll
        for (int i = 0; i < HTW; i++) ep[i] = rowPrev(hist[(q * 2 + 0) * 16 + j], w + NW * i);
        embed_and_send(q, 1u, hist[(q * 2 + 1) * 16 + j]);
    }

    // clock probe (Params::clk, see wavenet_wg): the head of chain 0 records shader and wall clock at both ends of its sample loop
    const bool probe = p.clk != nullptr && chainIdx == 0 && tid == 0;
    if (probe) {
        p.clk[0] = __builtin_amdgcn_s_memtime();
        p.clk[1] = __builtin_amdgcn_s_memrealtime();
    }
    const int tEnd = p.initSample + p.count;
    for (int t = p.initSample; t < tEnd; t++)
    for (int q = 0; q < nq; q++) {
        const unsigned tag = (unsigned)(t - p.initSample) + 1u;
        const bool dumpNow = DUMP && p.dump && (t == tEnd - 1);
        const int tile = cp.tile0 + q * cp.chains + chainIdx;
        const int b = tile * 16 + j;
        const bool uvalid = b < p.batch;
        const int ub = uvalid ? b : p.batch - 1;
        const unsigned long long* const skin = mstage + (size_t)q * (CC::XG + CC::SG) + CC::XG;
        int sb = tile * 16 + su;
        const bool sbValid = sb < p.batch;
        sb = sbValid ? sb : p.batch - 1;
        const float selv = p.useRng ? philox_selector(p.rngKey0, p.rngKey1, (unsigned)t, (unsigned)sb)
                                    : p.sel[(size_t)t * p.maxBatch + sb];
        const int yCur = hist[(q * 2 + 1) * 16 + j];

        // ---- skip sums of all layers arrive; + biases, ReLU -> B fragments -----------------------------
#pragma unroll
        for (int i = 0; i < HTW; i++) ep[i] = rowPrev(yCur, w + NW * i);   // for the sample after this one
        floatx4 sk[STW];
        WN_CT_DECL
        WN_CT(0)
        if (!recv_tiles_fast<STW, NW>(skin, w, lane, tag, sk, status, 0x300u, cp.timeoutTicks)) return;
        WN_CT(1)
#pragma unroll
        for (int i = 0; i < STW; i++) {
            floatx4 v = sk[i] + *(const floatx4*)(fsb + (w + NW * i) * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = __builtin_fmaxf(v[r], 0.f);
            lds_put_tile<F16>(skbuf, w + NW * i, lane, v);
            if (dumpNow && uvalid)   // the oracle applies the ReLU to the last layer's skipOut in place
                *(floatx4*)(p.skipOut + ((size_t)(L - 1) * p.maxBatch + ub) * S + (w + NW * i) * 16 + g * 4) = v;
        }
        wg_barrier();
        floatx4 zs[1][ATW];
        {
            frag sbf[1][KF_S];
            lds_get_frags<F16, KF_S>(skbuf, lane, sbf[0]);
#pragma unroll
            for (int i = 0; i < ATW; i++) zs[0][i] = *(const floatx4*)(headBias + (w + NW * i) * 16 + g * 4);
            if constexpr (HS == 0) gemm_pinned<F16, ATW, KF_S>(hw, 0, zs[0], sbf[0]);
            else {
                gemm<F16, PF, HSP, 1, ATW, KF_S>(ws, C::O_ZS, whead, whead, laneOff, zs, sbf);
#pragma unroll
                for (int i = 0; i < C::PAD1; i++) (void)take<F16, PF, HSP>(ws, C::FW_ZS + i, whead, whead, laneOff);
            }
        }
#pragma unroll
        for (int i = 0; i < ATW; i++) {
#pragma unroll
            for (int r = 0; r < 4; r++) zs[0][i][r] = __builtin_fmaxf(zs[0][i][r], 0.f);
            lds_put_tile<F16>(zsbuf, w + NW * i, lane, zs[0][i]);
            if (dumpNow && uvalid) *(floatx4*)(p.zs + (size_t)ub * A + (w + NW * i) * 16 + g * 4) = zs[0][i];
        }
        wg_barrier();
        {
            floatx4 za[1][ATW];
#pragma unroll
            for (int i = 0; i < ATW; i++) za[0][i] = *(const floatx4*)(headBias + A + (w + NW * i) * 16 + g * 4);
            if constexpr (C::ZA_B_FROM_LDS) {
                static_assert(HR == 0, "the LDS-streamed head is for the large, non-resident heads");
                gemm_ldsb<F16, PF, HSP, 1, ATW, KF_A>(ws, C::O_ZA, whead, whead, laneOff, za, zsbuf, lane);
            } else {
                frag zb[1][KF_A];
                lds_get_frags<F16, KF_A>(zsbuf, lane, zb[0]);
                if constexpr (HR >= C::FW_ZA) gemm_pinned<F16, ATW, KF_A>(hw, C::FW_ZS - HS, za[0], zb[0]);
                else gemm<F16, PF, HSP, 1, ATW, KF_A>(ws, C::O_ZA, whead, whead, laneOff, za, zb);
            }
            if constexpr (C::ALIAS_LG) wg_barrier();   // every wave is done with the zs image
#pragma unroll
            for (int i = 0; i < ATW; i++) {
                *(floatx4*)(lgbuf + j * C::LROW + (w + NW * i) * 16 + g * 4) = za[0][i];
                if (dumpNow && uvalid) *(floatx4*)(p.za + (size_t)ub * A + (w + NW * i) * 16 + g * 4) = za[0][i];
            }
        }
        if constexpr (HS == C::FHW) {
#pragma unroll
            for (int i = 0; i < C::PAD2; i++) (void)take<F16, PF, HSP>(ws, C::O_ZA + C::FW_ZA + i, whead, whead, laneOff);
        }
        wg_barrier();
        WN_CT(2)

        // ---- softmax + pick ------------------------------------------------------------------------
        int pickKeep;
        {
            float e[C::RPL];
            float total;
            const int pick = softmax_pick<A, C::LPU, C::RPL>(lgbuf + su * C::LROW + sq * C::RPL, sq, lane, selv, e, total);
            if (sq == 0) ybuf[su] = pick;
            pickKeep = pick;
            if (dumpNow && sbValid) {
                const float inv = 1.0f / total;
#pragma unroll
                for (int i = 0; i < C::RPL / 4; i++)
                    *(floatx4*)(p.p + (size_t)sb * A + sq * C::RPL + i * 4) =
                        floatx4{e[i * 4] * inv, e[i * 4 + 1] * inv, e[i * 4 + 2] * inv, e[i * 4 + 3] * inv};
            }
        }
        wg_barrier();
        WN_CT(3)
        const int yNew = ybuf[j];
        // (every thread writes the values of its own column j and reads them back at the tile's next visit: no barrier needed)
        hist[(q * 2 + 0) * 16 + j] = yCur;
        hist[(q * 2 + 1) * 16 + j] = yNew;
        if (t + 1 < tEnd) embed_and_send(q, tag + 1u, yNew);
        if (sq == 0 && sbValid) p.yOut[(size_t)sb * p.numSamples + t] = pickKeep;      // (behind the hand-off: see chain_layers)
        WN_CT(4)
        WN_CT_FLUSH(stage, t - p.initSample)
        // ybuf / lgbuf / skbuf are next written after the next unit's barriers
    }
    if (probe) {
        p.clk[2] = __builtin_amdgcn_s_memtime();
        p.clk[3] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    if (w == 0 && g == 0) {
        for (int q = 0; q < nq; q++) {
            const int b0 = (cp.tile0 + q * cp.chains + chainIdx) * 16 + j;
            if (b0 < p.batch) {
                p.yInPrev[b0] = hist[(q * 2 + 0) * 16 + j];
                p.yInCur[b0] = hist[(q * 2 + 1) * 16 + j];
            }
        }
    }
}

// ---- behind a chain launch (nvWavenetInfer::launchChain) ---------------------------------------------------------
// If the launch gave up (*status != 0: some spin ran into its bound because not every workgroup of the launch became
// resident in time), put the dilation rings and the sample history of its tiles back to what they were when it started,
// so that the gated wavenet_wg launch behind this kernel generates the same samples from the same state.
static __global__ void chain_restore_kernel(const unsigned* status, uintx4* ring, const uintx4* ringShadow, size_t n16,
                                            int* yInPrev, int* yInCur, const int* prevShadow, const int* curShadow, int nb) {
    if (__builtin_nontemporal_load(status) == 0u) return;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < n16; i += step) ring[i] = ringShadow[i];
    for (size_t i = i0; i < (size_t)nb; i += step) {
        yInPrev[i] = prevShadow[i];
        yInCur[i] = curShadow[i];
    }
}
// ... and once that launch is done: status[1] := the code, status[2] += 1, status[0] := 0 (the next launch starts clean)
static __global__ void chain_settle_kernel(unsigned* status) {
    const unsigned s = status[0];
    if (s != 0u) {
        status[1] = s;
        status[2] += 1u;
        status[0] = 0u;
    }
}

// One workgroup per (tile, stage).  Workgroup b is observed to run on XCD b % 8: the stages of a chain
// take workgroups of one residue class so that a chain's granules stay in one L2 (speed only).
template <bool F16, int R, int S, int A, bool DUMP, bool HOIST = false>
__global__ __launch_bounds__((Cfg<F16, R, S, A, 1>::THREADS), 1) void wavenet_chain(const Params p, const ChainParams cp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if constexpr (!CCfg<F16, R, S, A>::SUPPORTED) return;   // not even one layer fits a CU: never launched
    const int bidx = blockIdx.x;
    const int xcd = bidx & 7, q = bidx >> 3;
    const int stage = q % cp.stages;
    const int chainIdx = (q / cp.stages) * 8 + xcd;
    if (chainIdx >= cp.chains) return;
    if (stage == cp.stages - 1) chain_head<F16, R, S, A, DUMP>(p, cp, lds, chainIdx);
    else chain_layers<F16, R, S, A, DUMP, HOIST>(p, cp, lds, chainIdx, stage);
}

}  // namespace wn
