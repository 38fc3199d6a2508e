// wn_kernels.hpp -- gfx950 (MI355X / CDNA4) device code for autoregressive WaveNet inference.
//
// Replaces the device side of the reference (all four kernel organisations of
// /root/reference/nv_wavenet_singleblock.cuh, nv_wavenet_dualblock.cuh, nv_wavenet_persistent.cuh
// and the shared per-layer functions nv_wavenet.cuh:87-207, matrix_math.cuh, softmax.cuh) with a
// design that is native to CDNA4 rather than a translation of them.
//
//   * The batch is the MFMA N dimension.  Every mat-vec of the reference (one thread per output
//     row, K weights in that thread's registers, nv_wavenet.cuh:131-157, matrix_math.cuh:80-157)
//     becomes a [M x K] x [K x 16] MFMA GEMM over a tile of 16 utterances (BT = 1 .. 4 tiles per
//     workgroup share one pass over the weights).
//   * One workgroup = NW (4) wavefronts, one per SIMD, all working on the SAME utterance tile:
//     the M (output-row) dimension of every GEMM is split across the waves, so each wave streams
//     only its quarter of the weights.  Measured on MI355X (scripts/ubench/stream.hip): one wave
//     sustains ~30 GB/s of L2->VGPR traffic whatever the prefetch depth (the VGPR return path
//     of a SIMD), four waves on four SIMDs sustain 4x that -- so the weight stream, which is the
//     bound of this path, has to be spread over all SIMDs of the CU.
//   * Weights are pre-swizzled on upload into MFMA A-fragment order, per wave, in consumption
//     order: each wave reads ONE contiguous stream [layer 0 .. layer L-1][head] with 1-KiB
//     coalesced global_load_dwordx4, PF fragments ahead, straight into VGPRs (no LDS staging:
//     a fragment is used by exactly one wave).
//   * The MFMA result tile (lane (g,j): rows 4g..4g+3 of utterance j), converted to T_data, IS
//     the B-operand fragment of the next MFMA once the K order is permuted accordingly (done in
//     the weight packing), so activations are exchanged between the waves as ready-made
//     B fragments through LDS: one ds_write_b64 per produced tile, ds_read_b128 per consumed
//     fragment, conflict-free, one s_barrier per exchange (raw s_barrier + lgkmcnt only, so the
//     in-flight weight prefetch is never drained).
//   * Biases live in LDS for the whole launch and initialise the MFMA accumulators.
//   * The dilated history x_l[t-d_l] is a ring of exactly d_l slots per layer in global memory
//     (B-fragment order, 1-KiB coalesced rows): sum(d_l)*R*16 elements per tile instead of the
//     reference's (maxDilation+1)*(L+1) planes (nv_wavenet.cuh:334-335).  Round 6: during a launch the slots of the layers with the
//     shortest dilations -- as many as the LDS holds behind the tables, the whole ring for models with a short maxDilation -- live
//     in LDS (the LR instantiations of wavenet_wg: ring_lds_copy / tap_image; the reference stages x[t-d] through shared memory,
//     nv_wavenet.cuh:96-127), loaded from and spilled to their places in that global ring at the launch's ends.
//   * VMEM returns in order per wave.  Everything the weight stream could delay is kept off that
//     path (dilation schedule in scalar arithmetic, biases / embeddings in LDS), and the slow HBM
//     loads (conditioning, dilated tap) that could delay the weight stream are requested two layers
//     ahead, into register sets alternating by layer parity, at the point of the layer where the
//     longest take-free stretch begins; every refill of the weight ring is pinned to its take.
//   * softmax + inverse-CDF pick: logits go through LDS once; 16 lanes per utterance, reductions
//     with DPP row operations inside a 16-lane row (softmax_pick).
//   * Resident operands and the weight prefetch ring live in the accumulator file and are read in place by the
//     MFMAs (agpr_pin); the conditioning comes either packed in fragment order or, RAW kernels, straight from the
//     caller's fp32 tensor.
//   * This header is the single-workgroup organisation and the device primitives; wn_chain.hpp (multi-CU chain,
//     resident weights) builds on it.
//
// Layouts private to the engine (produced by the pack kernels at the bottom):
//   fragment of an M x K weight matrix: 16 rows x (16*TPF) k-values, 64 lanes x 16 B:
//       lane l=(g<<4|i), element e  <-  W[tile*16 + i][ (kf*TPF + (e>>2))*16 + g*4 + (e&3) ]
//   activation tile t of a vector v (MFMA D layout): lane (g,j) reg r = v[t*16 + g*4 + r] of utt j
//   wave w owns tiles t = w, w+NW, w+2NW, ... of every GEMM output.
//
// fp32 (T_data=float): v_mfma_f32_16x16x4_f32, exact fp32 FMA chains.  fp16 (T_data=half):
// v_mfma_f32_16x16x32_f16 with fp32 accumulation (the reference accumulates in fp16,
// matrix_math.cuh:119-157), fp32 transcendentals / softmax like the reference
// (nv_wavenet_util.cuh:78-86, softmax.cuh:43-47).
#pragma once

#include <hip/hip_runtime.h>
#include <utility>
#include <stdint.h>

#include <type_traits>

#include "wn_experiments.hpp"

namespace wn {

#define WN_DEV __device__ __forceinline__

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));

template <bool F16> struct Prec;
template <> struct Prec<true> {
    using elem = _Float16;
    using frag = half8;
    using quad = half4;                   // 4 consecutive elements
    static constexpr int EPL = 8;         // elements per lane per fragment (16 B)
    static constexpr int TPF = 2;         // 16-wide K tiles per fragment
};
template <> struct Prec<false> {
    using elem = float;
    using frag = floatx4;
    using quad = floatx4;
    static constexpr int EPL = 4;
    static constexpr int TPF = 1;
};

constexpr int kMaxLayers = 128;

constexpr int pick_pf(int fl, int pfmax) {
    int best = 1;
    for (int d = 1; d <= pfmax && d <= fl; d++)
        if (fl % d == 0) best = d;
    return best;
}
constexpr int align16(int x) { return (x + 15) & ~15; }

// KFC > 0: the conditioning is COMPUTED in the kernel from the upsampled features (round 5; role of the model's `cond_layers` 1x1
// convolution, pytorch/wavenet.py:190-202): every layer's part of a wave's stream then also carries Wcond[l] for this wave's gate
// tiles, KFC k-fragments wide (the features of an utterance and sample, zero-padded to KFC * 16 * TPF channels).
constexpr int kCondChannelsMax = 80;      // n_cond_channels of the reference's model (pytorch/config.json); fewer are zero-padded
template <bool F16> constexpr int feat_kfc() { return (kCondChannelsMax + 16 * Prec<F16>::TPF - 1) / (16 * Prec<F16>::TPF); }

template <bool F16, int R, int S, int A, int BT, int KFC = 0>
struct Cfg {
    using P = Prec<F16>;
    static constexpr int TPF = P::TPF, EPL = P::EPL;
    static constexpr int RT = R / 16, ST = S / 16, AT = A / 16;
    static constexpr int NW = RT >= 4 ? 4 : RT;            // wavefronts per workgroup
    static_assert(R % (16 * TPF) == 0 && S % (16 * TPF) == 0 && A % (16 * TPF) == 0, "R,S,A vs MFMA K step");
    static_assert(RT % NW == 0 && ST % NW == 0 && AT % NW == 0, "tiles must split evenly over the waves");
    static constexpr int THREADS = NW * 64;
    static constexpr int HTW = RT / NW, STW = ST / NW, ATW = AT / NW;   // tiles per wave
    static constexpr int KF_R = RT / TPF, KF_S = ST / TPF, KF_A = AT / TPF;
    // fragments of one layer per wave, LOGICAL order: prev | cur | res | skip | cond
    static constexpr int FW_GATE = 2 * HTW * KF_R, FW_RES = HTW * KF_R, FW_SKIP = STW * KF_R, FW_COND = 2 * HTW * KFC;
    static constexpr int O_PREV = 0, O_CUR = FW_GATE, O_RES = 2 * FW_GATE, O_SKIP = O_RES + FW_RES, O_COND = O_SKIP + FW_SKIP;
    static constexpr int FLW = O_COND + FW_COND;
    // PHYSICAL order of a wave's stream = the order wavenet_wg consumes it in:
    //   cur(0) cond(1) res(0) prev(1) | cur(1) skip(0) cond(2) res(1) prev(2) | ... | cur(L-1) skip(L-2) cond(0) res(L-1) prev(0) |
    //   skip(L-1) | head
    // (the skip GEMM of a layer and the conditioning GEMM of the layer after run under the next layer's gate arithmetic, the
    // dilated-tap GEMM of a layer in the exchange window at the end of the layer before; cond(0) prev(0) at the end belong to the
    // NEXT sample; cond(.) is empty unless KFC > 0).  The layers still take L*FLW fragments, and the part of layer l >= 1 starts
    // FW_SKIP fragments before l*FLW.
    // streamPos: logical fragment i of layer l -> position in the wave's stream.
    __host__ __device__ static constexpr size_t streamPos(int l, int i, int L) {
        return i < O_CUR    ? (size_t)((l == 0 ? L : l) - 1) * FLW + FW_GATE + FW_COND + FW_RES + (i - O_PREV)
               : i < O_RES  ? (l == 0 ? (size_t)(i - O_CUR) : (size_t)l * FLW - FW_SKIP + (i - O_CUR))
               : i < O_SKIP ? (size_t)l * FLW + FW_GATE + FW_COND + (i - O_RES)
               : i < O_COND ? (l == L - 1 ? (size_t)L * FLW - FW_SKIP + (i - O_SKIP)
                                          : (size_t)(l + 1) * FLW - FW_SKIP + FW_GATE + (i - O_SKIP))
                            : (size_t)((l == 0 ? L : l) - 1) * FLW + FW_GATE + (i - O_COND);
    }
    // the same places as seen from the kernel: relative to the start of the part of layer l-1 (= (l-1)*FLW, a whole
    // number of ring turns), so that position % PF is the ring slot
    static constexpr int P_CUR0 = FLW, P_CUR = FLW - FW_SKIP, P_SKIP = P_CUR + FW_GATE, P_COND = FLW + FW_GATE,
                         P_RES = P_COND + FW_COND, P_PREV = P_RES + FW_RES;
    // per-wave fragment stream of the head: zs | za
    static constexpr int FW_ZS = ATW * KF_S, FW_ZA = ATW * KF_A;
    static constexpr int FHW = FW_ZS + FW_ZA;
    // depth of the per-wave weight prefetch ring.  Measured on MI355X (C3 fp16, us per sample at
    // batch 16 / 4096): depth 3: 26.2 / 38.8, 6: 24.5 / 33.3, 9: 22.7 / 29.9, 18: 27.4 / 35.6 with the ring in
    // VGPRs (round 1).  With the ring in the accumulator file (agpr_pin: loads land in AGPRs, the MFMAs read
    // them in place) depth 9: 19.6 / 24.2 / 33.4 and depth 18 (a whole layer): 21.0 / 26.2 / 37.7 at batch
    // 16 / 4096 / 8192: a ring that does not divide the streamed head (32 fragments) is rotated once per sample,
    // and every such rotation -- like every register copy the compiler places on a loop edge -- drains the queue.
    static constexpr int PF = pick_pf(FLW, KFC > 0 ? WN_PFMAX_FEAT : WN_PFMAX);
    static_assert(FLW % PF == 0 && PF <= FW_ZS, "prefetch ring must divide the layer stream");
    // The head's weights (FHW fragments per wave) stay RESIDENT in registers for the whole launch
    // when they fit the otherwise idle accumulator half of the register file (4 regs/fragment):
    // the head then needs no weight stream at all (it is purely stream-bound otherwise).
    // HR = number of resident fragments (the tail of the head: all of it, or the A x A matrix only
    // when registers are scarcer, e.g. two tiles per workgroup); HS = head fragments still streamed.
    // Register budget of the resident head.  Measured (C3 fp16): keeping Wzs resident as well (256
    // registers) buys nothing over Wza alone (128) and costs spills once the layer loop is unrolled.
    static constexpr int HR = !F16 ? 0
                              : BT == 1 ? (FW_ZA * 4 <= WN_HEADREGS ? FW_ZA : 0)
                              : BT == 2 ? (FW_ZA * 4 <= WN_HEADREGS2 ? FW_ZA : 0)
                              : BT == 3 ? (FW_ZA * 4 <= WN_HEADREGS3 ? FW_ZA : 0)
                                        : (FW_ZA * 4 <= WN_HEADREGS4 ? FW_ZA : 0);
    static constexpr int HS = FHW - HR;
    static constexpr bool HEADRES = HS == 0;              // the stream cycles over the layers only
    // The head's part of a wave's stream is laid out  zs | PAD1 | za | PAD2  with zero fragments that bring each matrix
    // to a whole number of ring turns: however much of the head is streamed (nothing, zs, or zs and za), the ring comes
    // back in phase for the next sample by itself.  (Without them the ring had to be rotated through registers once
    // per sample, which needs every load in flight to have landed: a full drain of the prefetch queue.)
    static constexpr int PAD1 = (PF - FW_ZS % PF) % PF, PAD2 = (PF - FW_ZA % PF) % PF;
    static constexpr int O_ZS = 0, O_ZA = FW_ZS + PAD1;   // physical fragment offsets inside the head part
    static constexpr int FHWP = O_ZA + FW_ZA + PAD2;      // physical length of the head part
    static constexpr int HSP = HS == 0 ? 0 : (HS == FW_ZS ? FW_ZS + PAD1 : FHWP);   // physical length of what is streamed
    static_assert(HS == 0 || HS == FW_ZS || HS == FHW, "the streamed part of the head is nothing, zs, or all of it");
    // logical head fragment q (zs then za) -> physical offset
    __host__ __device__ static constexpr int headFrag(int q) { return q < FW_ZS ? q : q + PAD1; }
    static constexpr int FRAG_ELEMS = 64 * EPL;
    static constexpr int BIAS_L = 3 * R + S;               // fp32 biases per layer: Bh | Bres | Bskip
    static constexpr int COND_FR = 2 * HTW / TPF;          // conditioning fragments per (sample,layer,tile,wave)
    static constexpr int LPU = 4 * NW;                     // softmax lanes per utterance
    static constexpr int RPL = A / LPU;                    // logits per softmax lane
    static_assert(RPL % 4 == 0, "A too small for the softmax lane split");
    // ---- LDS layout (bytes); the fp32 bias table (runtime size) follows at LDS_FIXED ----
    static constexpr int XBUF = BT * KF_R * 1024;          // x as B fragments
    static constexpr int HBUF = XBUF;
    static constexpr int SKBUF = BT * KF_S * 1024;
    static constexpr int ZSBUF = BT * KF_A * 1024;
    static constexpr int LROW = A + 4;                     // padded logits row (floats)
    // the logits image holds LGT tiles at a time: four tiles per workgroup CAN pick their samples in two passes of two tiles (32.5 KiB
    // less than one pass of four: the current tap's embedding table then fits in LDS beside the fourth tile's exchange images)
    // (WN_SOFTMAX_2PASS, measured at 13 824 utterances, round 6: 41.6 / 42.0 us per sample against 41.3 / 41.9 in one pass with the
    //  embedding gathers going to L2 -- the two extra barriers cost what the LDS gathers save: off)
    static constexpr int LGT = (BT >= 4 && WN_SOFTMAX_2PASS) ? 2 : BT;
    static_assert(BT % LGT == 0, "the softmax passes take whole groups of tiles");
    static constexpr int LGBUF = LGT * 16 * LROW * 4;
    static constexpr int YBUF = align16(BT * 16 * 4);
    static constexpr int XPBUF = XBUF;                     // dilated tap x[t-d] as B fragments (shared by the waves)
    static constexpr int XPW = (KF_R + NW - 1) / NW;       // ring fragments owned (stored AND re-loaded) by one wave
    // Large heads (A = 1024 in fp32): the fp32 logits take the place of the zs fragment image (one
    // extra barrier between the last zs read and the first logit write), and the A x A GEMM reads its
    // B fragments from LDS as it goes instead of holding all KF_A of them in registers.
    // (also with three tiles per workgroup: the 24 KiB it frees let the current tap's embedding table into LDS)
    static constexpr bool ALIAS_LG = ZSBUF + LGBUF > 100 * 1024 || BT >= 3;
    static constexpr bool ZA_B_FROM_LDS = KF_A * BT * 4 > WN_ZA_B_REGS;
    static constexpr int OFF_X = 0, OFF_H = OFF_X + XBUF, OFF_SK = OFF_H + HBUF, OFF_ZS = OFF_SK + SKBUF;
    static constexpr int OFF_LG = ALIAS_LG ? OFF_ZS : OFF_ZS + ZSBUF;
    static constexpr int OFF_Y = ALIAS_LG ? OFF_ZS + (ZSBUF > LGBUF ? ZSBUF : LGBUF) : OFF_LG + LGBUF;
    static constexpr int OFF_XP = OFF_Y + YBUF;
    static constexpr int LDS_FIXED = OFF_XP + XPBUF;
    // + optionally both embedding tables (T_data) behind the bias table
    // Bias table in LDS.  Kernels that can dump hold the model's table as it is, [L][Bh | Bres | Bskip] + the head's, the skip rows turned
    // into running sums (the per-layer skipOut dump needs every one of them).  Dump-free kernels (round 6) only ever add the LAST running
    // sum -- at the head --, so they hold [L][Bh | Bres], ONE row of S skip-bias sums, then the head's: S * (L - 1) floats less (C3: 19 KiB).
    static constexpr int BIAS_LN = 3 * R;                  // per-layer stride of the dump-free table
    __host__ __device__ static constexpr int biasFloats(int L, bool dump) { return (dump ? L * BIAS_L : L * BIAS_LN + S) + 2 * A; }
    static size_t ldsBytes(int L, int embTables, bool dump = true) {      // embTables: 0, 1 (current tap only) or 2
        return (size_t)LDS_FIXED + (size_t)biasFloats(L, dump) * sizeof(float) + (size_t)embTables * A * R * sizeof(typename P::elem);
    }
    // one ring slot of a workgroup in LDS: x of one (layer, sample) as B fragments, all BT tiles (the layout of the tap image XPBUF)
    static constexpr int RING_SLOT = BT * KF_R * 1024;
    // slots of the layers with dilation <= D (the schedule of nv_wavenet.cuh:99,110-111)
    __host__ __device__ static constexpr int ldsRingSlots(int L, int maxDilation, int D) {
        int n = 0, d = 1;
        for (int l = 0; l < L; l++) {
            if (d <= D) n += d;
            d <<= 1;
            if (d > maxDilation) d = 1;
        }
        return n;
    }
    // per-wave stream in memory: [L][FLW] layers | [FHW] head
    __host__ __device__ static size_t headOffsetFrags(int L) { return (size_t)L * FLW; }
    __host__ __device__ static size_t waveStreamFrags(int L) { return (size_t)L * FLW + FHWP; }
};

// Dilation d_l and first ring slot of layer l.  wavenet_wg reads them from a table that travels in the kernel arguments
// (Params::dil, filled by the host): indexing the argument segment with a uniform layer number is a scalar load -- it does
// not queue behind the weight prefetch like a vector load would -- and replaces ~15 scalar ALU instructions per layer of
// schedule arithmetic (round 3).  The chain computes its few entries once per launch with dil_next.
// lds: first slot of the layer in the LDS part of the ring (wavenet_wg, round 6: the layers with d <= Params::ldsRingD keep their d
// slots in LDS for the length of a launch; counted over those layers only)
struct Dil {
    int d, off, lds;
};
WN_DEV Dil dil_first() { return Dil{1, 0, 0}; }
WN_DEV Dil dil_next(Dil s, int maxDilation, bool wrapToFirst, int ldsD = 0) {
    Dil n;
    n.off = s.off + s.d;
    n.lds = s.lds + (s.d <= ldsD ? s.d : 0);
    n.d = s.d << 1;
    if (n.d > maxDilation) n.d = 1;
    if (wrapToFirst) n = dil_first();
    return n;
}

// Everything the kernel needs, passed by value (role of nv_wavenet_params, nv_wavenet.cuh:40-85).
struct Params {
    const void* wblob;       // NW per-wave streams: [L][FLW] layer fragments | [FHW] head fragments
    const float* bias;       // [L][BIAS_L] then Bzs[A], Bza[A]
    const void* embPrev;     // [A][R] T_data
    const void* embCur;      // [A][R] T_data
    const void* cond;        // [sample][L][tile][wave][COND_FR] fragments
    const void* condRaw;     // or (RAW kernels) the caller's [sample][L][maxBatch][2R] tensor, read in place: fp32 (RAW = 1)
                             // or T_data = fp16 (RAW = 2, fp16 engine; the reference keeps m_Lh in T_data, nv_wavenet.cuh:326)
    int condRawKind;         // 0: packed `cond`; 1: condRaw is fp32; 2: condRaw is fp16; 3: `feat` (what the RAW template argument says,
                             // for the kernels that choose at run time)
    const void* feat;        // RAW = 3: upsampled features as B fragments, [sample][tile][KFC] fragments (T_data; condSamples samples);
                             // wblob / bias then are the stream WITH the conditioning weights and the gate biases + bcond
    const unsigned* gate;    // when non-NULL: the launch does nothing unless *gate != 0 (fallback behind a wavenet_chain launch)
    const float* sel;        // [N][maxBatch] uniform draws
    void* ring;              // [tile][ringSlots][KF_R] fragments
    int maxDilation;         // dilation doubles per layer and restarts at 1 past this (nv_wavenet.cuh:110-111)
    int* yInPrev;            // [maxBatch]
    int* yInCur;             // [maxBatch]
    int* yOut;               // [batch][numSamples]
    float* xtOut;            // [L][maxBatch][R]   (dump)
    float* skipOut;          // [L][maxBatch][S]   (dump)
    float* zs;               // [maxBatch][A]      (dump)
    float* za;               // [maxBatch][A]      (dump)
    float* p;                // [maxBatch][A]      (dump)
    int numLayers;
    int batch;               // utterances to generate (<= maxBatch)
    int maxBatch;            // batch stride of sel / dumps
    int numSamples;          // row stride of yOut
    int condSamples;         // samples held in cond / sel (maxSamples); cond has one padding sample more
    int initSample;
    int count;               // samples generated by this launch
    int ringSlots;           // sum of dilations
    int ldsRingD;            // wavenet_wg: layers with dilation <= this keep their ring slots in LDS during the launch (0: none), loaded from
                             // / spilled to their places in `ring` at the launch's start / end (the state between launches lives there)
    int tiles;               // ceil(maxBatch/16): tile stride of cond / ring
    int tileBase;            // first tile of this launch (workgroup b serves tiles tileBase + b*BT ...)
    int tanhEmbed;
    int dump;
    int embLds;              // embedding tables held in LDS: 0 none, 1 current tap, 2 both
    int useRng;              // selectors drawn in-kernel (Philox4x32-10) instead of read from `sel`
    unsigned rngKey0, rngKey1;
    unsigned long long* clk; // when non-NULL: workgroup 0 leaves {shader clock, wall clock} at its start and {..} at its end here
                             // (4 words): ticks of s_memtime over ticks of the constant-rate s_memrealtime = the clock the launch
                             // actually ran at (the chip clocks to its power budget: MI355X_MICROARCH.md, DVFS)
    Dil dil[kMaxLayers + 2]; // schedule of layer l; entries L and L+1 repeat layers 0 and 1 (of the next sample)
};

// ------------------------------------------------------------------------------------------
// math helpers
// ------------------------------------------------------------------------------------------

WN_DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
WN_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// In-kernel selector of (sample t, utterance b): Philox4x32-10 (Random123) with counter {t,b,0,0},
// top 24 bits of word 0 scaled to [0,1).  Replaces the host rand() table of
// pytorch/wavenet_infer.cu:92-94; oracle/wavenet_oracle.c:nvw_philox_selectors is
// the CPU restatement the tests compare against.
WN_DEV float philox_selector(unsigned k0, unsigned k1, unsigned t, unsigned b) {
    unsigned c0 = t, c1 = b, c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        if (r) {
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0;
        c1 = l1;
        c2 = n2;
        c3 = l0;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

// sigmoid: relative error of a few ulp (no cancellation)
WN_DEV float sigmoid_f(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// tanh, fp16 engine: result is rounded to fp16 afterwards, 1 - 2/(e^2x+1) is ample.
WN_DEV float tanh_fast(float x) {
    float e = fast_exp(2.0f * x);
    return 1.0f - 2.0f * fast_rcp(e + 1.0f);
}
// tanh, fp32 engine: the formula above loses relative accuracy for small |x| (cancellation), and
// the parity bars are relative (nv_wavenet_test.cu:273-298), so use the odd Taylor series below
// 0.55 and the exponential form above it.
WN_DEV float tanh_acc(float x) {
    float a = __builtin_fabsf(x);
    float x2 = x * x;
    float pz = -0.00886323552990220f;                       // -1382/155925
    pz = __builtin_fmaf(pz, x2, 0.0218694885361552f);       //  62/2835
    pz = __builtin_fmaf(pz, x2, -0.0539682539682540f);      // -17/315
    pz = __builtin_fmaf(pz, x2, 0.133333333333333f);        //  2/15
    pz = __builtin_fmaf(pz, x2, -0.333333333333333f);       // -1/3
    float small = __builtin_fmaf(x * x2, pz, x);
    float e = fast_exp(2.0f * a);
    float big = 1.0f - 2.0f * fast_rcp(e + 1.0f);
    big = __builtin_copysignf(big, x);
    return a < 0.55f ? small : big;
}
template <bool F16> WN_DEV float tanh_t(float x) { return F16 ? tanh_fast(x) : tanh_acc(x); }

// ---- the gate  h = tanh(a) * sigmoid(b) ---------------------------------------------------------
// fp32 engine: the accurate scalar forms above (parity bars are relative, nv_wavenet_test.cu:273-298).
// fp16 engine: the gate pre-activations arrive PRE-SCALED -- tanh rows by 2 log2(e), sigmoid rows by
// -log2(e), folded into Wprev, Wcur, Bh and the conditioning when they are packed (gate_prescale below)
// -- so the gate is exp2 / rcp with no multiplications in front:
//     tanh(a) = 1 - 2 / (2^a' + 1),   sigmoid(b) = 1 / (1 + 2^b').
// (Tried and slower on MI355X: the gate as one rational function with a single reciprocal, (7,6) Pade
// approximant of tanh: 1770 vs 1470 clk per layer in round 1's loader / consumer kernel -- the extra FMAs cost more than
// the transcendentals they save; a lone wave issues a v_exp_f32 every 8.9 clk, a v_fma_f32 every 5.8.)
template <bool F16> __host__ __device__ constexpr float gate_prescale(bool sigmoidRow) {
    return !F16 ? 1.0f : sigmoidRow ? -1.44269504088896340736f : 2.88539008177792681472f;
}
template <bool F16> WN_DEV float gate1(float a, float b);
template <> WN_DEV float gate1<false>(float a, float b) { return tanh_acc(a) * sigmoid_f(b); }
WN_DEV float gate_finish(float ra, float rb) { return (1.0f - 2.0f * ra) * rb; }
template <> WN_DEV float gate1<true>(float a, float b) {
    return gate_finish(fast_rcp(__builtin_amdgcn_exp2f(a) + 1.0f), fast_rcp(1.0f + __builtin_amdgcn_exp2f(b)));
}
// The same gate for a pair of values in five stages (exp2 a | exp2 b | rcp | rcp | product), so that wavenet_wg can
// issue one MFMA of an independent GEMM between two stages: a lone wave issues about two VALU instructions in the
// time one MFMA executes, and the gate is where a layer's VALU time is.  Same operations per value as gate1
// (bit-identical results).  fp32 engine: stage 4 is the whole accurate gate.
typedef float floatx2 __attribute__((ext_vector_type(2)));
template <bool F16, int ST>
WN_DEV void gate_stage(float a0, float a1, float b0, float b1, floatx2& ea, floatx2& eb, floatx2& ra, floatx2& rb,
                       floatx2& h) {
    if constexpr (!F16) {
        if constexpr (ST == 4) h = floatx2{gate1<false>(a0, b0), gate1<false>(a1, b1)};
    } else {
        if constexpr (ST == 0) ea = floatx2{__builtin_amdgcn_exp2f(a0), __builtin_amdgcn_exp2f(a1)};
        if constexpr (ST == 1) eb = floatx2{__builtin_amdgcn_exp2f(b0), __builtin_amdgcn_exp2f(b1)};
        // scalar adds: a packed-f32 VALU instruction issued beside MFMAs costs more than the two plain ones it replaces
        // (MI355X_MICROARCH.md: +22..26 clk per pair)
        if constexpr (ST == 2) ra = floatx2{fast_rcp(ea[0] + 1.0f), fast_rcp(ea[1] + 1.0f)};
        if constexpr (ST == 3) rb = floatx2{fast_rcp(eb[0] + 1.0f), fast_rcp(eb[1] + 1.0f)};
        if constexpr (ST == 4) h = floatx2{gate_finish(ra[0], rb[0]), gate_finish(ra[1], rb[1])};
    }
}

// compile-time loops: f(std::integral_constant<int, I>{}) for I in [0, N) / [A, B)
template <int A, typename F, int... I> WN_DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, A + I>{}), ...);
}
template <int N, typename F> WN_DEV void static_for(F&& f) { static_for_impl<0>(f, std::make_integer_sequence<int, N>{}); }
template <int A, int B, typename F> WN_DEV void static_for_range(F&& f) {
    static_for_impl<A>(f, std::make_integer_sequence<int, (B > A ? B - A : 0)>{});
}
template <bool F16> WN_DEV floatx4 gate4(floatx4 a, floatx4 b) {
    floatx4 h;
#pragma unroll
    for (int r = 0; r < 4; r++) h[r] = gate1<F16>(a[r], b[r]);
    return h;
}

WN_DEV floatx4 mma(half8 a, half8 b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
WN_DEV floatx4 mma(floatx4 a, floatx4 b, floatx4 c) {
#pragma unroll
    for (int s = 0; s < 4; s++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
    return c;
}

WN_DEV floatx4 quad_to_f32(half4 q) { return floatx4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]}; }
WN_DEV floatx4 quad_to_f32(floatx4 q) { return q; }

// Workgroup barrier that does NOT wait for outstanding global loads (the weight prefetch ring
// stays in flight across it): only this wave's LDS traffic is drained. __syncthreads() would emit
// s_waitcnt vmcnt(0) as well.
WN_DEV void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- LDS exchange of activations as B fragments -------------------------------------------
// tile t of a vector, held in MFMA D layout (fp32), goes to its place in the fragment image
template <bool F16> WN_DEV void lds_put_tile(char* buf, int tile, int lane, floatx4 v);
template <> WN_DEV void lds_put_tile<true>(char* buf, int tile, int lane, floatx4 v) {
    half4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *(half4*)(buf + (((tile >> 1) * 64 + lane) << 4) + ((tile & 1) << 3)) = h;
}
template <> WN_DEV void lds_put_tile<false>(char* buf, int tile, int lane, floatx4 v) {
    *(floatx4*)(buf + ((tile * 64 + lane) << 4)) = v;
}
template <bool F16, int KF>
WN_DEV void lds_get_frags(const char* buf, int lane, typename Prec<F16>::frag (&b)[KF]) {
#pragma unroll
    for (int k = 0; k < KF; k++) b[k] = *(const typename Prec<F16>::frag*)(buf + ((k * 64 + lane) << 4));
}

// ---- values pinned in accumulator registers ---------------------------------------------------------------
// An empty asm statement whose output is an "a"-class register tied to its input: from there on the value IS
// an accumulator-file (AGPR) value, and the MFMA builtins take it as their A operand in place
// (v_mfma ... a[n:n+3], v[..], v[..]) with the compiler doing the hazard bookkeeping as for any operand.
// Applied to a LOADED value right where it is consumed, the load itself is allocated an AGPR destination
// (global_load_dwordx4 a[n:n+3], ...) and the wait for it stays a counted s_waitcnt vmcnt(depth-1) in front of
// the MFMA: the weight prefetch ring then costs no architectural VGPRs at all and can be a whole layer deep.
// (Left alone, the compiler keeps such values in the VGPR class, spills them to AGPRs under pressure and
// copies each fragment back with 4 v_accvgpr_read + wait states in front of its MFMA.)
WN_DEV floatx4 agpr_pin(floatx4 v) {
    floatx4 o;
    asm volatile("" : "=a"(o) : "0"(v));
    return o;
}
template <bool PIN, typename FRAG> WN_DEV FRAG agpr_operand(FRAG f) {
    // fp16 builds keep the MFMA accumulators in VGPRs (-amdgpu-mfma-vgpr-form), so the AGPR file is free for
    // operands; fp32 builds (the parity mode) accumulate in AGPRs and leave the placement to the compiler
    if constexpr (PIN) return __builtin_bit_cast(FRAG, agpr_pin(__builtin_bit_cast(floatx4, f)));
    else return f;
}

// ------------------------------------------------------------------------------------------
// weight stream: PF fragments always in flight ahead of the MFMA that consumes them
// ------------------------------------------------------------------------------------------
// PIN: the ring lives in the accumulator file (agpr_pin); off where the AGPRs are needed as spill space instead
template <bool F16, int PF, bool PIN = F16> struct WStream {
    typename Prec<F16>::frag buf[PF];
};

// Consume fragment `idx` (position relative to `base`, a compile-time constant after unrolling;
// `base` sits a whole number of layers from the start of the wave's stream, so idx % PF is the ring
// slot) and refill the slot with fragment idx+PF.  The stream is contiguous across layers and into
// the head, so the refill address is linear, except at the end of the head (WRAP = its length)
// where it continues at `wrapBase` (layer 0 of the next sample).
template <bool F16, int PF, int WRAP, bool PIN>
WN_DEV typename Prec<F16>::frag take(WStream<F16, PF, PIN>& ws, int idx, const char* base, const char* wrapBase,
                                     unsigned laneOff, int rtWrapAt = 0x7fffffff, long rtWrapDelta = 0) {
    using frag = typename Prec<F16>::frag;
    frag a = agpr_operand<F16 && PIN>(ws.buf[idx % PF]);   // the ring lives in the accumulator file (see agpr_pin)
    int nidx = idx + PF;
    // base / wrapBase are wave-uniform (SGPR base), laneOff = lane*16.  rtWrapAt/rtWrapDelta: a
    // run-time (uniform) wrap point, used when the stream cycles over the layers only.
    const char* src = (WRAP > 0 && nidx >= WRAP) ? wrapBase + (size_t)(nidx - WRAP) * 1024 : base + (size_t)nidx * 1024;
    if (nidx >= rtWrapAt) src += rtWrapDelta;
    ws.buf[idx % PF] = *(const frag*)(src + laneOff);
    // The refill stays where the ring discipline puts it.  Left to the scheduler the loads get
    // clustered and re-ordered, and now and then a fragment that is needed next ends up among the
    // youngest requests (a near-drain of the queue): pinned, C3 fp16 runs 39.0 instead of 42.0 us per
    // sample with two tiles per workgroup at batch 8192.
    __builtin_amdgcn_sched_barrier(0);
    return a;
}

// acc[bt][mt] += W(tile mt) * b[bt]   for MT tiles of this wave, KF k-fragments, BT batch tiles
template <bool F16, int PF, int WRAP, int BT, int MT, int KF, bool PIN>
WN_DEV void gemm(WStream<F16, PF, PIN>& ws, int pos0, const char* cur, const char* next, unsigned laneOff,
                 floatx4 (&acc)[BT][MT], const typename Prec<F16>::frag (&b)[BT][KF], int rtWrapAt = 0x7fffffff,
                 long rtWrapDelta = 0) {
    // fragment order inside a GEMM: groups of G tile slots, k-fragment-major inside a group, so that
    // MFMAs accumulating into the same tile are G instructions apart (no dependent-MFMA stall)
    constexpr int G = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / G; mg++) {
#pragma unroll
        for (int kf = 0; kf < KF; kf++) {
#pragma unroll
            for (int mi = 0; mi < G; mi++) {
                const int mt = mg * G + mi;
                auto a = take<F16, PF, WRAP>(ws, pos0 + (mg * KF + kf) * G + mi, cur, next, laneOff, rtWrapAt, rtWrapDelta);
#pragma unroll
                for (int bt = 0; bt < BT; bt++) acc[bt][mt] = mma(a, b[bt][kf], acc[bt][mt]);
            }
        }
    }
}

// same, with the B fragments read from their LDS image as they are needed (KF too large for registers)
template <bool F16, int PF, int WRAP, int BT, int MT, int KF, bool PIN>
WN_DEV void gemm_ldsb(WStream<F16, PF, PIN>& ws, int pos0, const char* cur, const char* next, unsigned laneOff,
                      floatx4 (&acc)[BT][MT], const char* bimg, int lane) {
    using frag = typename Prec<F16>::frag;
    constexpr int G = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / G; mg++) {
#pragma unroll
        for (int kf = 0; kf < KF; kf++) {
            frag b[BT];
#pragma unroll
            for (int bt = 0; bt < BT; bt++) b[bt] = *(const frag*)(bimg + (((bt * KF + kf) * 64 + lane) << 4));
#pragma unroll
            for (int mi = 0; mi < G; mi++) {
                const int mt = mg * G + mi;
                auto a = take<F16, PF, WRAP>(ws, pos0 + (mg * KF + kf) * G + mi, cur, next, laneOff);
#pragma unroll
                for (int bt = 0; bt < BT; bt++) acc[bt][mt] = mma(a, b[bt], acc[bt][mt]);
            }
        }
    }
}

// ---- the same stream through a BUFFER resource (wavenet_wg) ----------------------------------------------------------
// A lone wave per SIMD issues one instruction about every 5.5 clk whatever its kind, and wavenet_wg is bound by that and
// by the latency of its loads, so the stream is read with as few instructions as possible: buffer_load
// with the wave's stream as the resource -- the address is an SGPR byte offset + lane*16 + a 12-bit immediate, no
// per-lane 64-bit address arithmetic, one s_add per 4 KiB of stream -- and each refill goes straight into the slot the
// MFMAs in front of it have just read (no second register for the slot).
// WN_TAKE_G > 1 takes the fragments of one k step together (pinning the YOUNGEST first makes one s_waitcnt vmcnt cover
// the group: 52 -> 32 waits per two layers).  Measured slower (C3 fp16, two tiles: 30 vs 28 us per sample): a group
// waits for its youngest fragment before its first MFMA, which shortens the 9-fragment lookahead by G-1, and at
// ~700 clk per L2 load the stream has no slack for that.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__host__ __device__ constexpr int take_g(int g, int pf) {     // group size: divides g, at most WN_TAKE_G and the ring
    int t = g < WN_TAKE_G ? g : WN_TAKE_G;
    t = t < pf ? t : pf;
    while (g % t) t--;
    return t;
}
WN_DEV rsrc_t make_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, -1, 0x00020000); }
// AUX: cache policy bits of the instruction (0 = default, 2 = non-temporal / streaming)
template <typename FRAG, int AUX = 0> WN_DEV FRAG buf_load(rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(FRAG, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX));
}
template <typename FRAG, int AUX = 0> WN_DEV void buf_store(rsrc_t rs, unsigned voff, unsigned soff, FRAG v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), rs, voff, soff, AUX);
}
// fragments idx .. idx+G-1 (positions relative to basePos, a whole number of ring turns from the start of the stream)
template <bool F16, int PF, bool PIN, int G>
WN_DEV void take_group(WStream<F16, PF, PIN>& ws, int idx, typename Prec<F16>::frag (&a)[G]) {
#pragma unroll
    for (int i = G - 1; i >= 0; i--) a[i] = agpr_operand<F16 && PIN>(ws.buf[(idx + i) % PF]);
}
// ... and their slots refilled with fragments idx+PF .. ; past WRAP (the end of the head) the stream continues at wrapPos
template <bool F16, int PF, int WRAP, bool PIN, int G>
WN_DEV void refill_group(WStream<F16, PF, PIN>& ws, rsrc_t rs, int idx, int basePos, int wrapPos, unsigned laneOff) {
    using frag = typename Prec<F16>::frag;
#pragma unroll
    for (int i = 0; i < G; i++) {
        const int nidx = idx + i + PF;
        const bool wr = WRAP > 0 && nidx >= WRAP;
        const int rel = wr ? nidx - WRAP : nidx;                         // compile-time after unrolling
        const int pos = (wr ? wrapPos : basePos) + (rel & ~3);           // uniform; never negative (see the layer-0 note in wavenet_wg)
        ws.buf[(idx + i) % PF] = buf_load<frag, WN_W_AUX>(rs, laneOff + (unsigned)(rel & 3) * 1024u, (unsigned)pos * 1024u);
    }
    __builtin_amdgcn_sched_barrier(0);
}
// acc[bt][mt] += W(tile mt) * b[bt]   (fragment order as in gemm())
template <bool F16, int PF, int WRAP, int BT, int MT, int KF, bool PIN>
WN_DEV void gemm_b(WStream<F16, PF, PIN>& ws, rsrc_t rs, int pos0, int basePos, int wrapPos, unsigned laneOff,
                   floatx4 (&acc)[BT][MT], const typename Prec<F16>::frag (&b)[BT][KF]) {
    constexpr int G = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / G; mg++) {
#pragma unroll
        for (int kf = 0; kf < KF; kf++) {
            constexpr int TG = take_g(G, PF);
#pragma unroll
            for (int m0 = 0; m0 < G; m0 += TG) {
                typename Prec<F16>::frag a[TG];
                const int idx = pos0 + (mg * KF + kf) * G + m0;
                take_group<F16, PF, PIN, TG>(ws, idx, a);
                // the refill of the take BEFORE this one sits between this take's pin and its first MFMA: the slot it writes was
                // read by MFMAs already issued, and the instruction fills the wait state the pinned operand needs in front of an
                // MFMA (an s_nop otherwise, 18 a layer; round 4: -1.8 % on one workgroup, -0.5 % at 12 288 utterances)
                // (needs a ring deeper than a take: the slot being refilled must not be the one just taken)
                constexpr bool EARLY = PF > TG;
                if (EARLY && !(mg == 0 && kf == 0 && m0 == 0)) refill_group<F16, PF, WRAP, PIN, TG>(ws, rs, idx - TG, basePos, wrapPos, laneOff);
#pragma unroll
                for (int mi = 0; mi < TG; mi++)
#pragma unroll
                    for (int bt = 0; bt < BT; bt++)
                        acc[bt][mg * G + m0 + mi] = mma(a[mi], b[bt][kf], acc[bt][mg * G + m0 + mi]);
                if (!EARLY || (mg == MT / G - 1 && kf == KF - 1 && m0 + TG >= G)) refill_group<F16, PF, WRAP, PIN, TG>(ws, rs, idx, basePos, wrapPos, laneOff);
            }
        }
    }
}
// same, with the B fragments read from their LDS image as they are needed (KF too large for registers)
template <bool F16, int PF, int WRAP, int BT, int MT, int KF, bool PIN>
WN_DEV void gemm_ldsb_b(WStream<F16, PF, PIN>& ws, rsrc_t rs, int pos0, int basePos, int wrapPos, unsigned laneOff,
                        floatx4 (&acc)[BT][MT], const char* bimg, int lane) {
    using frag = typename Prec<F16>::frag;
    constexpr int G = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / G; mg++) {
#pragma unroll
        for (int kf = 0; kf < KF; kf++) {
            frag b[BT];
#pragma unroll
            for (int bt = 0; bt < BT; bt++) b[bt] = *(const frag*)(bimg + (((bt * KF + kf) * 64 + lane) << 4));
            constexpr int TG = take_g(G, PF);
#pragma unroll
            for (int m0 = 0; m0 < G; m0 += TG) {
                frag a[TG];
                take_group<F16, PF, PIN, TG>(ws, pos0 + (mg * KF + kf) * G + m0, a);
#pragma unroll
                for (int mi = 0; mi < TG; mi++)
#pragma unroll
                    for (int bt = 0; bt < BT; bt++) acc[bt][mg * G + m0 + mi] = mma(a[mi], b[bt], acc[bt][mg * G + m0 + mi]);
                refill_group<F16, PF, WRAP, PIN, TG>(ws, rs, pos0 + (mg * KF + kf) * G + m0, basePos, wrapPos, laneOff);
            }
        }
    }
}
// takes that only keep the ring turning (zero fragments that pad a matrix to a whole number of ring turns)
template <bool F16, int PF, int WRAP, bool PIN, int N>
WN_DEV void skip_frags(WStream<F16, PF, PIN>& ws, rsrc_t rs, int idx, int basePos, int wrapPos, unsigned laneOff) {
    if constexpr (N > 0) {
        typename Prec<F16>::frag a[N];
        take_group<F16, PF, PIN, N>(ws, idx, a);
        refill_group<F16, PF, WRAP, PIN, N>(ws, rs, idx, basePos, wrapPos, laneOff);
    }
}

// same with the weight fragments loaded on the spot (launch prologue: no ring)
template <bool F16, int BT, int MT, int KF>
WN_DEV void gemm_direct(const char* src, unsigned laneOff, floatx4 (&acc)[BT][MT], const typename Prec<F16>::frag (&b)[BT][KF]) {
    using frag = typename Prec<F16>::frag;
    constexpr int G = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / G; mg++)
#pragma unroll
        for (int kf = 0; kf < KF; kf++)
#pragma unroll
            for (int mi = 0; mi < G; mi++) {
                const frag a = *(const frag*)(src + (size_t)((mg * KF + kf) * G + mi) * 1024 + laneOff);
#pragma unroll
                for (int bt = 0; bt < BT; bt++) acc[bt][mg * G + mi] = mma(a, b[bt][kf], acc[bt][mg * G + mi]);
            }
}

// same with register-resident weight fragments
template <bool F16, int BT, int MT, int KF, int NFR>
WN_DEV void gemm_res(const typename Prec<F16>::frag (&wres)[NFR], int pos0, floatx4 (&acc)[BT][MT],
                     const typename Prec<F16>::frag (&b)[BT][KF]) {
    constexpr int G = MT >= 4 ? 4 : MT;
#pragma unroll
    for (int mg = 0; mg < MT / G; mg++)
#pragma unroll
        for (int kf = 0; kf < KF; kf++)
#pragma unroll
            for (int mi = 0; mi < G; mi++)
#pragma unroll
                for (int bt = 0; bt < BT; bt++)
                    acc[bt][mg * G + mi] = mma(agpr_operand<F16>(wres[pos0 + (mg * KF + kf) * G + mi]), b[bt][kf], acc[bt][mg * G + mi]);
}

// ------------------------------------------------------------------------------------------
// softmax + inverse-CDF pick of ONE utterance by a group of LPU consecutive lanes (softmax.cuh:36-191;
// oracle matrix.cpp:166-183 + nv_wavenet_reference.cpp:106-121).  Lane `sq` of the group holds the RPL
// consecutive logits at `lrow` (row sq*RPL .. sq*RPL+RPL-1 of the A logits).  Returns, in every lane of
// the group, the first row whose cumulative un-normalised probability exceeds sel * total (the oracle's
// "sel < cumulative p"), or 128 when the scan falls off the end (softmax.cuh:154-155).  e[] receives
// this lane's exp(logit - max) values and `total` their sum over the group (for the probability dump).
// ------------------------------------------------------------------------------------------
// Cross-lane steps inside one 16-lane DPP row (LPU <= 16 lanes serve an utterance): data-parallel-primitive
// moves on the VALU operand path instead of __shfl_* (ds_bpermute: an LDS-crossbar round trip of ~100 clk
// each, 12 of them per pick = half of the 1 us this function took).
template <int CTRL> WN_DEV float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> WN_DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;              // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;   // lane i <-> 7-i (in 8), i <-> 15-i (in 16)
// all-reduce of a commutative, associative, EXACT op (max / min) over the LPU lanes of a group: butterflies
// inside the quads, then mirrors (a lane's mirror partner holds the other half's result)
template <int LPU, typename F> WN_DEV float group_reduce_f(float v, F op) {
    static_assert(LPU == 4 || LPU == 8 || LPU == 16, "softmax lanes per utterance");
    v = op(v, dpp_f<DPP_XOR1>(v));
    v = op(v, dpp_f<DPP_XOR2>(v));
    if (LPU >= 8) v = op(v, dpp_f<DPP_HALF_MIRROR>(v));
    if (LPU >= 16) v = op(v, dpp_f<DPP_MIRROR>(v));
    return v;
}
template <int LPU> WN_DEV int group_min_i(int v) {
    int o = dpp_i<DPP_XOR1>(v); v = o < v ? o : v;
    o = dpp_i<DPP_XOR2>(v); v = o < v ? o : v;
    if (LPU >= 8) { o = dpp_i<DPP_HALF_MIRROR>(v); v = o < v ? o : v; }
    if (LPU >= 16) { o = dpp_i<DPP_MIRROR>(v); v = o < v ? o : v; }
    return v;
}

template <int A, int LPU, int RPL>
WN_DEV int softmax_pick(const float* lrow, int sq, int lane, float sel, float (&e)[RPL], float& total) {
    (void)lane;
#pragma unroll
    for (int i = 0; i < RPL / 4; i++) {
        floatx4 v = *(const floatx4*)(lrow + i * 4);
#pragma unroll
        for (int r = 0; r < 4; r++) e[i * 4 + r] = v[r];
    }
    float m = e[0];
#pragma unroll
    for (int i = 1; i < RPL; i++) m = __builtin_fmaxf(m, e[i]);
    m = group_reduce_f<LPU>(m, [](float a, float b) { return __builtin_fmaxf(a, b); });
    float lsum = 0.f;
    {
        // exp(x - m) = 2^(x log2 e - m log2 e): one fma in front of the v_exp_f32 instead of a subtraction and a multiplication
        constexpr float kLog2e = 1.44269504088896340736f;
        const float mneg = -m * kLog2e;
#pragma unroll
        for (int i = 0; i < RPL; i++) {
            e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(e[i], kLog2e, mneg));
            lsum += e[i];
        }
    }
    // inclusive scan of the lane sums over the LPU lanes of this utterance (row_shr:o reads lane - o of the row)
    float incl = lsum;
    {
        float up = dpp_f<0x111>(incl);
        if (sq >= 1) incl += up;
        up = dpp_f<0x112>(incl);
        if (sq >= 2) incl += up;
        if (LPU >= 8) {
            up = dpp_f<0x114>(incl);
            if (sq >= 4) incl += up;
        }
        if (LPU >= 16) {
            up = dpp_f<0x118>(incl);
            if (sq >= 8) incl += up;
        }
    }
    // the group's total is its last lane's inclusive sum = the largest of the (non-decreasing) prefix sums
    total = group_reduce_f<LPU>(incl, [](float a, float b) { return __builtin_fmaxf(a, b); });
    const float target = sel * total;
    // first row of this lane whose cumulative sum exceeds the target
    // The running sums of non-negative terms never decrease, so the rows whose sum does NOT exceed the target form a prefix:
    // its length is the first row that does (compare + add-with-carry per row instead of two compares and a select).
    float cum = incl - lsum;   // exclusive prefix of this lane
    int first = 0;
#pragma unroll
    for (int i = 0; i < RPL; i++) {
        cum += e[i];
        first += (cum <= target) ? 1 : 0;
    }
    int pick = first < RPL ? sq * RPL + first : 0x7fffffff;
    pick = group_min_i<LPU>(pick);
    if (pick >= A) pick = 128;             // scan fell off the end (softmax.cuh:154-155)
    return pick;
}

// conditioning fragment k of a tile from its registers: the packed fragment itself, or (fp16 engine reading the
// caller's fp32 tensor in place) built from the two fp32 quads of the gate pair -- scaled by the gate's pre-scale and
// rounded to fp16 exactly like pack_cond_tiled_kernel does
// fp16(value * scale) of a pair of values into one packed register, one v_fma_mix per value: the product is formed
// in fp32 (from fp32 or from fp16 sources, converted exactly) and rounded to fp16 by the instruction.  The pack kernel and
// both in-place paths use THESE functions, so the three ways of handing over the conditioning agree bit for bit.
WN_DEV unsigned scale_pair_f32(float a, float b, float sc) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "s"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(d) : "v"(b), "s"(sc));
    return d;
}
WN_DEV unsigned scale_pair_f16(unsigned h2, float sc) {      // h2: two fp16 values
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "s"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h2), "s"(sc));
    return d;
}
// RAW = 2: the caller's tensor is fp16 already; register k holds the two raw quads of the gate pair (tanh tile | sigmoid tile)
template <bool F16, int RAW, int CR> WN_DEV typename Prec<F16>::frag cond_frag(const typename Prec<F16>::frag (&c)[CR], int k) {
    if constexpr (RAW == 1 && F16) {
        const floatx4 a = __builtin_bit_cast(floatx4, c[2 * k]), b = __builtin_bit_cast(floatx4, c[2 * k + 1]);
        const float st = gate_prescale<true>(false), ss = gate_prescale<true>(true);
        return __builtin_bit_cast(half8, uintx4{scale_pair_f32(a[0], a[1], st), scale_pair_f32(a[2], a[3], st),
                                                scale_pair_f32(b[0], b[1], ss), scale_pair_f32(b[2], b[3], ss)});
    } else if constexpr (RAW == 2 && F16) {
        const uintx4 q = __builtin_bit_cast(uintx4, c[k]);
        const float st = gate_prescale<true>(false), ss = gate_prescale<true>(true);
        return __builtin_bit_cast(half8, uintx4{scale_pair_f16(q[0], st), scale_pair_f16(q[1], st), scale_pair_f16(q[2], ss),
                                                scale_pair_f16(q[3], ss)});
    } else {
        return c[k];
    }
}

// acc += the conditioning fragment.  A fragment in B layout IS the D layout of TPF result tiles (lane (g,j), element e: row 4g + (e&3)
// of tile e>>2, utterance j), so the add is element by element in the lane that holds both.
// fp16, WN_COND_VALU (round 6): one v_fma_mix_f32 per value -- acc = fp16 value * 1.0 + acc, the conversion folded into the add -- 8 VALU
// instructions per fragment.  Rounds 1-5 added it on the matrix core through a 0/1 selection matrix (2 MFMAs per fragment: 8.6 % of a
// sample's MFMAs, each a 16x16x32 product of which one term per output is not zero): the same sum -- one exact product, one rounding --
// at 8 k multiply-adds of matrix-core energy per 64 useful additions.  Both forms are kept behind the switch for the A/B.
WN_DEV float fma_mix_lo(unsigned h2, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(acc));
    return d;
}
WN_DEV float fma_mix_hi(unsigned h2, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(acc));
    return d;
}
// a[0 .. TPF-1]: the TPF consecutive result tiles the fragment covers
template <bool F16> WN_DEV void cond_add(floatx4* a, typename Prec<F16>::frag c, const typename Prec<F16>::frag* selA) {
    if constexpr (F16) {
#if WN_COND_VALU
        (void)selA;
        const uintx4 q = __builtin_bit_cast(uintx4, c);
        a[0][0] = fma_mix_lo(q[0], a[0][0]);
        a[0][1] = fma_mix_hi(q[0], a[0][1]);
        a[0][2] = fma_mix_lo(q[1], a[0][2]);
        a[0][3] = fma_mix_hi(q[1], a[0][3]);
        a[1][0] = fma_mix_lo(q[2], a[1][0]);
        a[1][1] = fma_mix_hi(q[2], a[1][1]);
        a[1][2] = fma_mix_lo(q[3], a[1][2]);
        a[1][3] = fma_mix_hi(q[3], a[1][3]);
#else
#pragma unroll
        for (int tt = 0; tt < Prec<F16>::TPF; tt++) a[tt] = mma(selA[tt], c, a[tt]);
#endif
    } else {
        (void)selA;
#pragma unroll
        for (int e = 0; e < Prec<F16>::EPL; e++) a[e >> 2][e & 3] += (float)c[e];
    }
}

// ------------------------------------------------------------------------------------------
// the engine kernel: one workgroup generates `count` samples for BT tiles of 16 utterances
// ------------------------------------------------------------------------------------------
// EMBLDS: both embedding tables are copied to LDS at launch, so the gather that follows every
// sample pick (on the critical path) is a ds_read instead of a global load queued behind the
// in-flight weight prefetch.
// DUMP: the variant that can write the activation dump of the launch's last sample (getXtOut ...).
// Production launches (dumpActivations = false) use DUMP = false: even as a never-taken branch the
// dump costs accumulator read-outs in every layer (35.6 vs 39.0 us per sample at batch 8192).
// RAW: the conditioning is read in place from the caller's [N][L][B][2R] tensor (Params::condRaw: no packed copy
// exists), fp32 (RAW = 1) or, fp16 engine, T_data = fp16 (RAW = 2: half the bytes; the reference keeps its conditioning in
// T_data, nv_wavenet.cuh:326); each lane loads the 4 consecutive channels of its utterance per gate tile (16 / 8 bytes) and, in
// the fp16 engine, scales and rounds them exactly as pack_cond_tiled_kernel would have, so packed and in-place runs are
// bit-identical (for an fp16 tensor: identical to packing its values).
// RAW = 3: the conditioning is computed HERE from the upsampled features (Params::feat; role of the model's cond_layers,
// pytorch/wavenet.py:190-202): Lh[t][l] = Wcond[l] c[t] + bcond[l] as KFC more k steps of the gate GEMM, their weights in the
// wave's stream (Cfg<.., KFC>), their B operands -- KFC fragments per tile, 160 B per utterance instead of 2R * L values --
// loaded once per sample.  Summation order of the gate pre-activation: (Bh + bcond), Wcond c, dilated tap, current tap.
// LR: the launch keeps the ring slots of the short dilations in LDS (Params::ldsRingD; see ring_lds_copy).  A separate instantiation:
// the uniform branches it puts into every layer cost 3-5 % of a sample's cycles (measured, LABNOTES round 6), which pays when most of
// the ring traffic goes (a model whose whole ring fits) and not for two layers of twenty (C3 at three tiles per workgroup).
template <bool F16, int R, int S, int A, int BT, bool EMBLDS, bool DUMP = true, int RAW = 0, bool LR = false>
// (WN_EXP_TWO_WG, experiment build of round 6: the one-tile kernel with a register budget that lets TWO workgroups share a CU -- two waves per
//  SIMD, each with its own tile and its own weight stream; the A/B the round-5 review asked for at 8 192 utterances.  LABNOTES round 6.)
#ifdef WN_EXP_TWO_WG
__global__ __launch_bounds__((Cfg<F16, R, S, A, BT>::THREADS), (BT == 1 ? 2 : 1)) void wavenet_wg(const Params p) {
#else
__global__ __launch_bounds__((Cfg<F16, R, S, A, BT>::THREADS), 1) void wavenet_wg(const Params p) {
#endif
    constexpr bool FEAT = RAW == 3;
    constexpr int KFC = FEAT ? feat_kfc<F16>() : 0;
    using C = Cfg<F16, R, S, A, BT, KFC>;
    using P = Prec<F16>;
    using frag = typename P::frag;
    using quad = typename P::quad;
    using elem = typename P::elem;
    constexpr int PF = C::PF, FLW = C::FLW, FHW = C::FHW, NW = C::NW;
    constexpr int RT = C::RT, HTW = C::HTW, STW = C::STW, ATW = C::ATW;
    constexpr int KF_R = C::KF_R, KF_S = C::KF_S, KF_A = C::KF_A;

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const xbuf = lds + C::OFF_X;
    char* const hbuf = lds + C::OFF_H;
    char* const skbuf = lds + C::OFF_SK;
    char* const zsbuf = lds + C::OFF_ZS;
    float* const lgbuf = (float*)(lds + C::OFF_LG);
    int* const ybuf = (int*)(lds + C::OFF_Y);
    char* const xpbuf = lds + C::OFF_XP;
    float* const biasLds = (float*)(lds + C::LDS_FIXED);

    static_assert(RAW == 0 || RAW == 1 || (RAW == 2 && F16) || RAW == 3, "RAW: 0 packed, 1 fp32 in place, 2 fp16 in place (fp16 engine), 3 features");
    if (p.gate != nullptr && __builtin_nontemporal_load(p.gate) == 0u) return;   // (a fallback launch that is not needed)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int L = p.numLayers;
    const int tile0 = p.tileBase + blockIdx.x * BT;
    if (p.clk != nullptr && blockIdx.x == 0 && tid == 0) {
        p.clk[0] = __builtin_amdgcn_s_memtime();
        p.clk[1] = __builtin_amdgcn_s_memrealtime();
    }

    // utterance of this lane in its MFMA role (column j of tile bt)
    int ub[BT];
    bool uvalid[BT];
#pragma unroll
    for (int bt = 0; bt < BT; bt++) {
        const int b = (tile0 + bt) * 16 + j;
        uvalid[bt] = b < p.batch;
        ub[bt] = uvalid[bt] ? b : p.batch - 1;
    }
    // utterance of this lane in its softmax role: 16 utterances over NW*64 lanes
    const int su = tid / C::LPU, sq = tid % C::LPU;

    // ---- biases -> LDS ----------------------------------------------------------------------
    // The skip accumulator is only touched by MFMAs inside the layer loop: the per-layer skip biases
    // are turned into running sums here (slot of layer l := Bskip_0 + ... + Bskip_l, added in layer
    // order like the oracle does) and added once, at the head and in the per-layer dumps.
    // DUMP kernels keep every running sum (the skipOut dump of layer l needs sum l); dump-free kernels only the last (Cfg::biasFloats).
    constexpr int BLS = DUMP ? C::BIAS_L : C::BIAS_LN;            // per-layer stride of the table in LDS
    if constexpr (DUMP) {
        const int nb = L * C::BIAS_L + 2 * A;
        for (int i = tid; i < nb; i += C::THREADS) biasLds[i] = p.bias[i];
        __syncthreads();
        for (int s0 = tid; s0 < S; s0 += C::THREADS) {
            float run = biasLds[3 * R + s0];
            for (int l = 1; l < L; l++) {
                run += biasLds[l * C::BIAS_L + 3 * R + s0];
                biasLds[l * C::BIAS_L + 3 * R + s0] = run;
            }
        }
    } else {
        for (int i = tid; i < L * C::BIAS_LN; i += C::THREADS) biasLds[i] = p.bias[(i / C::BIAS_LN) * C::BIAS_L + i % C::BIAS_LN];
        for (int i = tid; i < 2 * A; i += C::THREADS) biasLds[L * C::BIAS_LN + S + i] = p.bias[L * C::BIAS_L + i];
        for (int s0 = tid; s0 < S; s0 += C::THREADS) {
            float run = p.bias[3 * R + s0];
            for (int l = 1; l < L; l++) run += p.bias[l * C::BIAS_L + 3 * R + s0];      // (the same additions in the same order)
            biasLds[L * C::BIAS_LN + s0] = run;
        }
    }
    const float* const skipBiasSum = DUMP ? biasLds + (L - 1) * C::BIAS_L + 3 * R : biasLds + L * C::BIAS_LN;      // Bskip_0 + ... + Bskip_{L-1}
    const float* const headBias = biasLds + (C::biasFloats(L, DUMP) - 2 * A);

    const unsigned laneOff = (unsigned)lane * 16u;
    // wave-uniform byte bases (SGPRs); per-lane part is laneOff
    const char* const wbase = (const char*)p.wblob + (size_t)w * C::waveStreamFrags(L) * 1024;
    const char* const whead = wbase + C::headOffsetFrags(L) * 1024;
    const elem* embPrev = (const elem*)p.embPrev;
    const elem* embCur = (const elem*)p.embCur;
    if constexpr (EMBLDS) {
        // p.embLds tables fit in LDS: the current tap's (its gather follows every pick, on the
        // critical path) and, if there is room, the older tap's (gathered one sample early)
        elem* const embLds = (elem*)(biasLds + C::biasFloats(L, DUMP));
        const floatx4* s0 = (const floatx4*)p.embCur;
        const floatx4* s1 = (const floatx4*)p.embPrev;
        constexpr int CH = (int)(A * R * sizeof(elem) / 16);
        for (int i = tid; i < CH; i += C::THREADS) {
            ((floatx4*)embLds)[i] = s0[i];
            if (p.embLds > 1) ((floatx4*)embLds)[CH + i] = s1[i];
        }
        embCur = embLds;
        if (p.embLds > 1) embPrev = embLds + A * R;
        __syncthreads();   // the tables are complete before the first gather below
    }

    int yPrev[BT], yCur[BT];
    floatx4 ep[BT][HTW];      // embedding row of the older tap, known one sample early
#pragma unroll
    for (int bt = 0; bt < BT; bt++) {
        yPrev[bt] = p.yInPrev[ub[bt]];
        yCur[bt] = p.yInCur[ub[bt]];
#pragma unroll
        for (int i = 0; i < HTW; i++)
            ep[bt][i] = quad_to_f32(*(const quad*)(embPrev + (size_t)yPrev[bt] * R + (w + NW * i) * 16 + g * 4));
    }

    // ---- prime the weight ring ----------------------------------------------------------------
    // (two tiles per workgroup reading the conditioning in place need the accumulator file as spill space: ring in VGPRs)
    constexpr bool ws_pin = F16 && !(RAW == 1 && BT == 2);
    WStream<F16, PF, ws_pin> ws;
    const rsrc_t rsW = make_rsrc(wbase);      // the wave's weight stream as a buffer: fragment positions become SGPR offsets
#pragma unroll
    for (int i = 0; i < PF; i++) ws.buf[i] = buf_load<frag, WN_W_AUX>(rsW, laneOff + (unsigned)(i & 3) * 1024u, (unsigned)(i & ~3) * 1024u);

    // ---- prefetch of the dilated input + conditioning of (sample tn, layer ln) ----------------
    // ---- resident head weights ------------------------------------------------------------------
    constexpr int HR = C::HR, HS = C::HS;
    frag hw[HR ? HR : 1];
    if constexpr (HR > 0) {
#pragma unroll
        for (int i = 0; i < HR; i++) hw[i] = *(const frag*)(whead + (size_t)C::headFrag(HS + i) * 1024 + laneOff);
    }

    // ---- prefetch of the dilated input + conditioning, TWO layers ahead (HBM latency of the
    //      conditioning stream exceeds half a layer) ---------------------------------------------
    // Each wave re-loads only the ring fragments it stored itself (k % NW == w) and the waves share
    // them through LDS: 1/NW of the ring loads per wave instead of all of them.
    constexpr int XPW = C::XPW;
    // Two register sets used alternately by layer parity (no rotation copies: a copy would force
    // the load issued one layer earlier to have landed, i.e. halve the prefetch distance).
    frag xpA[BT][XPW], xpB[BT][XPW];            // layers of even / odd parity
    // conditioning registers per tile: COND_FR packed fragments, or (fp16, in-place) one fp32 quad per gate tile
    constexpr int CR = (RAW == 1 && F16) ? 2 * C::COND_FR : C::COND_FR;
    frag cdA[BT][CR], cdB[BT][CR];      // (unused when the conditioning is computed here: FEAT)
    // FEAT: the feature fragments of the sample being generated (cfCur) and of the next one (cfNext): the conditioning GEMM under
    // the gate of the last layer belongs to layer 0 of the NEXT sample, so cfNext moves into cfCur behind the conditioning GEMM
    // under the last-but-one layer's gate, and is requested again (sample t+2) behind the sample's last take -- a whole sample
    // ahead of its use
    constexpr int KFCR = FEAT ? KFC : 1;
    frag cfCur[BT][KFCR], cfNext[BT][KFCR];
    const size_t featStride = (size_t)p.tiles * KFC * 1024;                          // one sample of features
    const char* const featMine = (const char*)p.feat + (size_t)tile0 * KFC * 1024;
    auto load_feat = [&](int tn, frag (&dst)[BT][KFCR]) {
        const int tc = tn < p.condSamples ? tn : p.condSamples - 1;                  // (the read-ahead past the last sample is never used)
        const rsrc_t rsF = make_rsrc(featMine + (size_t)tc * featStride);
#pragma unroll
        for (int bt = 0; bt < BT; bt++)
#pragma unroll
            for (int k = 0; k < KFC; k++)
                dst[bt][k] = buf_load<frag, WN_FEAT_AUX>(rsF, laneOff + (unsigned)(k & 3) * 1024u, (unsigned)((bt * KFC + (k & ~3)) * 1024));
    };
    // per-(sample,layer) strides in bytes; everything here is wave-uniform (SALU)
    const size_t condStride = (size_t)p.tiles * NW * C::COND_FR * 1024;            // one (sample,layer) row
    const char* const condMine = (const char*)p.cond + ((size_t)tile0 * NW + w) * C::COND_FR * 1024;
    const size_t ringTile = (size_t)p.ringSlots * KF_R * 1024;
    char* const ringMine = (char*)p.ring + (size_t)tile0 * ringTile;
    // loads (sample tn, layer ln) into (xd, cdd); ln may run past L-1 into the next sample.
    // The conditioning buffer carries one padding sample, so (tEnd, 0..1) stays in bounds.
    // Ring and conditioning are addressed through buffer resources as well (SGPR offsets, no per-lane pointers) and
    // always with the streaming cache policy (cache-policy bit 1, `nt`): touched once per sample, they must not evict
    // the weights every CU of an XCD re-reads from its L2.  (A run-time choice of the policy for small batches costs
    // more in selects than the cached accesses save: 21.7 vs 21.0 us at 16 utterances.)  The conditioning rows (sample, layer) are requested in exactly the order they lie
    // in memory, one per layer, so their resource just advances by one row per call.
    const rsrc_t rsRing = make_rsrc(ringMine);
    const unsigned ringTileB = (unsigned)ringTile;
    // the part of the ring that is in LDS for the length of the launch (see ring_lds_copy below): layers with dilation <= ldsD
    const int ldsD = LR ? p.ldsRingD : 0;
    char* const ringLds = (char*)(biasLds + C::biasFloats(L, DUMP)) + (size_t)(EMBLDS ? p.embLds : 0) * A * R * sizeof(elem);
    const char* condNext = condMine + (size_t)p.initSample * L * condStride;
    // in-place conditioning: one row = [maxBatch][2R] source elements; per-lane part of the address (utterance, channel quad)
    constexpr unsigned RAWE = RAW == 2 ? 2u : 4u;                                   // bytes per source element
    // cache policy of the in-place reads: a (sample, layer) row of an utterance is 2R elements, of which a wave takes two
    // quads per gate tile -- the four waves of the workgroup share every cache line of it
    // (default policy rather than streaming: 44.7 instead of 46.8 us per sample at 12 288 utterances from an fp16 tensor)
    // cache-policy bits of the ring loads / ring stores / packed-conditioning loads (2 = nt, streaming; experiments: 0, 1 = sc0, 16 = sc1)
    const size_t rawRow = (size_t)p.maxBatch * (2 * R) * RAWE;
    unsigned rawOff[BT];
#pragma unroll
    for (int bt = 0; bt < BT; bt++) rawOff[bt] = ((unsigned)ub[bt] * (unsigned)(2 * R) + (unsigned)g * 4u) * RAWE;
    auto prefetch = [&](int tn, int ln, Dil dl, frag (&xd)[BT][XPW], frag (&cdd)[BT][CR]) {
        if (ln >= L) { ln -= L; tn += 1; }
        const unsigned slot = (unsigned)(dl.off + (tn & (dl.d - 1)));
        const unsigned rp0 = slot * (unsigned)(KF_R * 1024);
        const rsrc_t rsCond = make_rsrc(condNext);
        condNext += condStride;
        if (dl.d > ldsD) {                   // (a layer whose slots are in LDS has nothing to request)
#pragma unroll
            for (int i = 0; i < XPW; i++) {
                const int k = w + NW * i;
                if (k < KF_R) {                  // (one uniform branch, all tiles inside)
#pragma unroll
                    for (int bt = 0; bt < BT; bt++)
                        xd[bt][i] = buf_load<frag, WN_RING_LD_AUX>(rsRing, laneOff, rp0 + (unsigned)bt * ringTileB + (unsigned)k * 1024u);
                }
            }
        }
#pragma unroll
        for (int bt = 0; bt < BT; bt++) {
            if constexpr (FEAT) {
                // nothing to load per layer
            } else if constexpr (RAW != 0) {
                // gate slot it = 0 .. 2*HTW-1 -> tile w + NW*(it>>1) (+RT for the sigmoid half): 4 channels of this lane's utterance
                const int tc = tn < p.condSamples ? tn : p.condSamples - 1;      // (the read-ahead past the last sample is never used)
                const rsrc_t rsRaw = make_rsrc((const char*)p.condRaw + ((size_t)tc * L + ln) * rawRow);
                auto slotOff = [&](int it) { return (unsigned)((w + NW * (it >> 1) + (it & 1) * RT) * 16) * RAWE; };
                if constexpr (RAW == 1) {
#pragma unroll
                    for (int it = 0; it < 2 * HTW; it++) cdd[bt][it] = buf_load<frag, WN_RAW_AUX>(rsRaw, rawOff[bt], slotOff(it));
                } else {
#pragma unroll
                    for (int k = 0; k < HTW; k++) {      // (fp16: COND_FR = HTW; one register = the tanh and the sigmoid quad)
                        const uintx2 qa = __builtin_amdgcn_raw_buffer_load_b64(rsRaw, rawOff[bt], slotOff(2 * k), WN_RAW_AUX);
                        const uintx2 qb = __builtin_amdgcn_raw_buffer_load_b64(rsRaw, rawOff[bt], slotOff(2 * k + 1), WN_RAW_AUX);
                        cdd[bt][k] = __builtin_bit_cast(frag, uintx4{qa[0], qa[1], qb[0], qb[1]});
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < C::COND_FR; k++)
                    cdd[bt][k] = buf_load<frag, WN_COND_AUX>(rsCond, laneOff + (unsigned)(k & 3) * 1024u,
                                                   (unsigned)((bt * NW * C::COND_FR + (k & ~3)) * 1024));
            }
        }
    };
    // ---- the short dilations' ring slots in LDS (round 6; north_star: "ring buffer staged in LDS with coalesced HBM spill") -------
    // Layers with d <= ldsD keep their d slots here for the whole launch: their tap x_l[t-d] is read by every wave straight from the
    // slot (no HBM load, no publication through the tap image), and x_l[t] replaces it from registers behind the h barrier of the
    // layer (every wave has consumed the tap by then).  The state between launches stays in the HBM ring, same places as ever: the
    // slots are loaded from it here and spilled back at the end of the launch, 1-KiB rows either way.  (Reference role:
    // nv_wavenet.cuh:96-127 stages x[t-d] through shared memory out of a global ring, :334-335.)
    auto ring_lds_copy = [&](auto TOLDS) {
        constexpr bool toLds = decltype(TOLDS)::value;
        if (ldsD <= 0) return;
        Dil dl = dil_first();
        for (int l = 0; l < L; l++) {
            if (dl.d <= ldsD) {
                // rows of 1 KiB: (slot, tile, fragment); wave w takes rows w, w + NW, ...
                const int rows = dl.d * BT * KF_R;
                for (int r = w; r < rows; r += NW) {
                    const int sl = r / (BT * KF_R), bt = (r / KF_R) % BT, k = r % KF_R;
                    const unsigned hb = (unsigned)(dl.off + sl) * (unsigned)(KF_R * 1024) + (unsigned)bt * ringTileB + (unsigned)k * 1024u;
                    char* const lp = ringLds + (size_t)(dl.lds + sl) * C::RING_SLOT + ((bt * KF_R + k) * 64 + lane) * 16;
                    if constexpr (toLds) *(frag*)lp = buf_load<frag, WN_RING_LD_AUX>(rsRing, laneOff, hb);
                    else buf_store<frag, WN_RING_ST_AUX>(rsRing, laneOff, hb, *(const frag*)lp);
                }
            }
            dl = dil_next(dl, p.maxDilation, false, ldsD);
        }
    };
    ring_lds_copy(std::true_type{});
    prefetch(p.initSample, 0, p.dil[0], xpA, cdA);
    prefetch(p.initSample, 1, p.dil[1], xpB, cdB);
    // publish this wave's fragments of the NEXT layer's dilated tap to LDS.  Before the start (t < d) the tap is zero (reference
    // :287): the ring slot read then has not been written in this utterance, and the engine clears the rings when a new utterance
    // is handed over (nvWavenetInfer::resetHistory / clearRings), so the load itself brings the zeros -- no branch here.
    // (dN: schedule entry of the layer the tap belongs to -- its slots in LDS need no publication)
    auto publish_xp = [&](const frag (&xpN)[BT][XPW], const Dil dN) {
        if (dN.d <= ldsD) return;
#pragma unroll
        for (int i = 0; i < XPW; i++) {
            const int k = w + NW * i;
            if (k < KF_R) {
#pragma unroll
                for (int bt = 0; bt < BT; bt++) *(frag*)(xpbuf + ((bt * KF_R + k) * 64 + lane) * 16) = xpN[bt][i];
            }
        }
    };
    // where the waves read the tap of (layer dN, sample tn) as B fragments: its ring slot in LDS, or the tap image
    auto tap_image = [&](const Dil dN, int tn) -> const char* {
        return dN.d <= ldsD ? ringLds + (size_t)(dN.lds + (tn & (dN.d - 1))) * C::RING_SLOT : xpbuf;
    };
    publish_xp(xpA, p.dil[0]);   // layer 0 (d = 1) of the first sample

    __syncthreads();   // bias table visible

#ifdef WN_TIMING
    // experiment build only: per-phase shader-clock sums of wave 0, written to p.p[0..15]
    unsigned long long tacc[12] = {0};
    unsigned long long tmark = 0;
#define WN_TMARK(i)                                                        \
    {                                                                      \
        unsigned long long _n = __builtin_amdgcn_s_memtime();              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 \
        tacc[i] += _n - tmark;                                             \
        tmark = _n;                                                        \
    }
#else
#define WN_TMARK(i)
#endif
    // selA[tt]: A operand that copies the rows of tile tt of a B-layout fragment into a result tile:
    // lane (g,i), element e is 1 iff e>>2 == tt and 4g + (e&3) == i   (see the fragment layout above)
    frag selA[P::TPF];
#pragma unroll
    for (int tt = 0; tt < P::TPF; tt++)
#pragma unroll
        for (int e = 0; e < P::EPL; e++)
            selA[tt][e] = (elem)(((e >> 2) == tt && g * 4 + (e & 3) == j) ? 1.0f : 0.0f);
    // acc: gate pre-activation of the next layer to run (see the layer schedule below).  For layer 0 of the first
    // sample it is formed here: bias + conditioning + dilated tap, the tap's weights read straight from their place
    // at the end of the layer stream.
    floatx4 acc[BT][2 * HTW];
    {
        frag xp[BT][KF_R];
#pragma unroll
        for (int bt = 0; bt < BT; bt++) {
            lds_get_frags<F16, KF_R>(tap_image(p.dil[0], p.initSample) + bt * KF_R * 1024, lane, xp[bt]);
#pragma unroll
            for (int i = 0; i < HTW; i++) {
                acc[bt][2 * i] = *(const floatx4*)(biasLds + (w + NW * i) * 16 + g * 4);
                acc[bt][2 * i + 1] = *(const floatx4*)(biasLds + (w + NW * i + RT) * 16 + g * 4);
            }
        }
        if constexpr (FEAT) {
            load_feat(p.initSample, cfNext);
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int k = 0; k < KFC; k++) cfCur[bt][k] = agpr_operand<F16>(cfNext[bt][k]);
            gemm_direct<F16, BT, 2 * HTW, KFC>(wbase + C::streamPos(0, C::O_COND, L) * 1024, laneOff, acc, cfCur);
            load_feat(p.initSample + 1, cfNext);
            // ... and waited for HERE (an empty asm that reads them): a load still pending at the loop entry makes the compiler drain
            // the whole queue (vmcnt(0): the weight ring with it) at the move behind layer L-2 of EVERY sample; from the loop's own
            // requests -- a whole sample of loads ahead of their reader -- it knows they have landed
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int k = 0; k < KFC; k++) asm volatile("" ::"v"(cfNext[bt][k]));
        } else {
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int k = 0; k < C::COND_FR; k++) cond_add<F16>(&acc[bt][k * P::TPF], cond_frag<F16, RAW>(cdA[bt], k), selA);
        }
        gemm_direct<F16, BT, 2 * HTW, KF_R>(wbase + C::streamPos(0, C::O_PREV, L) * 1024, laneOff, acc, xp);
    }
    const int tEnd = p.initSample + p.count;
    for (int t = p.initSample; t < tEnd; t++) {
        const bool dumpNow = DUMP && p.dump && (t == tEnd - 1);

        // selector of the utterance this lane serves in the softmax: from the uploaded table (requested here, a whole sample
        // before its use) or drawn in-kernel (further down, next to the head GEMMs)
        float selv[BT];
        if (!p.useRng) {
#pragma unroll
            for (int bt = 0; bt < BT; bt++) {
                int sb = (tile0 + bt) * 16 + su;
                sb = sb < p.batch ? sb : p.batch - 1;
                selv[bt] = p.sel[(size_t)t * p.maxBatch + sb];
            }
        }

        // ---- embedding (nv_wavenet_reference.cpp:42-56): each wave makes its own x tiles ------
        WN_TMARK(11)
        floatx4 x[BT][HTW];
#pragma unroll
        for (int bt = 0; bt < BT; bt++) {
#pragma unroll
            for (int i = 0; i < HTW; i++) {
                const int tile = w + NW * i;
                floatx4 ec = quad_to_f32(*(const quad*)(embCur + (size_t)yCur[bt] * R + tile * 16 + g * 4));
                floatx4 v = ep[bt][i] + ec;
                if (p.tanhEmbed) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = tanh_t<F16>(v[r]);
                }
                x[bt][i] = v;
                lds_put_tile<F16>(xbuf + bt * KF_R * 1024, tile, lane, v);
            }
#pragma unroll
            for (int i = 0; i < HTW; i++)
                ep[bt][i] = quad_to_f32(*(const quad*)(embPrev + (size_t)yCur[bt] * R + (w + NW * i) * 16 + g * 4));
        }
        wg_barrier();
        WN_TMARK(0)

        floatx4 skip[BT][STW];
#pragma unroll
        for (int bt = 0; bt < BT; bt++)
#pragma unroll
            for (int i = 0; i < STW; i++) skip[bt][i] = floatx4{0.f, 0.f, 0.f, 0.f};

        // ---- L dilated layers (nv_wavenet_reference.cpp:58-92) -------------------------------
        // Schedule of one layer for a wave.  Critical path: x exchange -> cur GEMM -> gate -> h exchange -> res GEMM
        // -> x exchange.  Everything else runs where that path leaves the matrix core idle:
        //  * the skip GEMM of layer l-1 is issued MFMA by MFMA between the stages of the gate arithmetic of layer l
        //    (a lone wave issues ~2 VALU instructions while one MFMA executes);
        //  * bias + conditioning + dilated-tap GEMM of layer l+1 (which do not depend on x_{l+1}) are formed around
        //    the x exchange that ends layer l: conditioning MFMAs while the x stores drain, the tap GEMM while the
        //    x fragments come back from LDS.  acc therefore always holds the pre-activation of the NEXT layer to run.
        // The weight stream is laid out in exactly this order (Cfg::streamPos).
        frag hb[BT][KF_R];
        frag xb[BT][KF_R];
#pragma unroll
        for (int bt = 0; bt < BT; bt++) lds_get_frags<F16, KF_R>(xbuf + bt * KF_R * 1024, lane, xb[bt]);
        // (xpC, cdC): register set of this layer's parity: both were consumed during the previous layer (tap
        // published, conditioning added), the prefetch for layer l+2 refills them; (xpN, cdN): the other set,
        // holding tap and conditioning of layer l+1.  dN: dilation of layer l+1, tN: the sample it belongs to.
        auto layer = [&](auto withSkip, const int l, const Dil dl, const Dil dN, const Dil dl2, frag (&xpC)[BT][XPW],
                         frag (&cdC)[BT][CR], const frag (&xpN)[BT][XPW], const frag (&cdN)[BT][CR]) {
            constexpr bool SKIP = decltype(withSkip)::value;
            // fragment positions (Cfg::P_*) count from the start of the part of layer l-1; layer 0 (the instance without a
            // skip GEMM) has no such part: its positions count from the start of the stream, so that no refill position
            // is ever formed from a negative base (with a ring shallower than FLW % 4 the 4-fragment-aligned part of the
            // first refills of layer 0 came out negative: R = 32 / S = 256 in fp16, whose layer stream of 13 fragments admits
            // only a one-deep ring, read garbage there -- found by the O(1)-recipe parity test of round 3)
            constexpr int PB = SKIP ? 0 : FLW;
            const int wl = SKIP ? (l - 1) * FLW : 0;
            const float* bl = biasLds + l * BLS;
            const int lN = l + 1 < L ? l + 1 : 0;
            const float* blN = biasLds + lN * BLS;
            const int d = dl.d;

            // current tap on top of bias + conditioning + dilated tap (xb: x as B fragments, requested behind the
            // x barrier)
            WN_TMARK(1)
            gemm_b<F16, PF, 0, BT, 2 * HTW, KF_R>(ws, rsW, SKIP ? C::P_CUR : C::P_CUR0 - PB, wl, 0, laneOff, acc, xb);
            // x_l[t] replaces x_l[t-d] in the ring (same slot): in HBM right here; a layer whose slots are in LDS keeps this wave's
            // fragments in registers and stores them behind the h barrier (other waves may still be reading the tap from that slot)
            const bool ringInLds = d <= ldsD;
            frag xkeep[BT][XPW];
            if (!ringInLds) {
                const unsigned rp = (unsigned)(dl.off + (t & (d - 1))) * (unsigned)(KF_R * 1024);
#pragma unroll
                for (int k = 0; k < KF_R; k++)
                    if (k % NW == w) {       // (one uniform branch per fragment index, all tiles inside)
#pragma unroll
                        for (int bt = 0; bt < BT; bt++)
                            buf_store<frag, WN_RING_ST_AUX>(rsRing, laneOff + (unsigned)(k & 3) * 1024u,
                                               rp + (unsigned)bt * ringTileB + (unsigned)(k & ~3) * 1024u, xb[bt][k]);
                    }
            } else {
#pragma unroll
                for (int k = 0; k < KF_R; k++)
                    if (k % NW == w) {
#pragma unroll
                        for (int bt = 0; bt < BT; bt++) xkeep[bt][k / NW] = xb[bt][k];
                    }
            }
            if constexpr (!SKIP) prefetch(t, l + 2, dl2, xpC, cdC);      // (layer 0 has no skip GEMM under its gate)
            __builtin_amdgcn_sched_barrier(0);
            WN_TMARK(2)

            // gate -> h tiles of this wave -> LDS, in stages (pairs of values: exp2 a | exp2 b | rcp | rcp | product)
            // with the MFMAs of the previous layer's skip GEMM in between:  skip <- Wskip h + skip
            // FEAT: ... and behind them the conditioning GEMM of the NEXT layer, accN <- (Bh + bcond) + Wcond c, likewise MFMA by MFMA
            // under the gate: nothing of it is on the dependent chain there (at the end of the layer, where the packed
            // conditioning is added, 6 k-steps per tile sit between the x stores and the x barrier: 540 clk per layer at three
            // tiles, measured)
            floatx4 accN[FEAT ? BT : 1][2 * HTW];
            if constexpr (FEAT) {
#pragma unroll
                for (int bt = 0; bt < BT; bt++)
#pragma unroll
                    for (int i = 0; i < HTW; i++) {
                        accN[bt][2 * i] = *(const floatx4*)(blN + (w + NW * i) * 16 + g * 4);
                        accN[bt][2 * i + 1] = *(const floatx4*)(blN + (w + NW * i + RT) * 16 + g * 4);
                    }
            }
            if constexpr (FEAT) {
                constexpr int NS = BT * HTW * 2 * 5;                          // gate stages
                constexpr int NFS = SKIP ? STW * KF_R : 0, NFC = C::FW_COND;   // fragments: skip, then cond
                constexpr int NM = (NFS + NFC) * BT;                           // MFMA slots
                constexpr int G0S = STW >= 4 ? 4 : STW, G0C = 2 * HTW >= 4 ? 4 : 2 * HTW;     // (fragment order inside a GEMM: see gemm_b)
                constexpr bool EARLY = PF > 1;      // (see gemm_b; a one-deep ring refills its only slot right behind its MFMAs)
                floatx2 ea, eb, ra, rb, hp;
                floatx4 hv;
                auto stage = [&](auto SI) {
                    constexpr int s = decltype(SI)::value, pr = s / 5, st = s % 5;
                    constexpr int bt = pr / (2 * HTW), i = (pr / 2) % HTW, r = (pr & 1) * 2;
                    gate_stage<F16, st>(acc[bt][2 * i][r], acc[bt][2 * i][r + 1], acc[bt][2 * i + 1][r], acc[bt][2 * i + 1][r + 1],
                                        ea, eb, ra, rb, hp);
                    if constexpr (st == 4) {
                        hv[r] = hp[0];
                        hv[r + 1] = hp[1];
                        if constexpr (r == 2) lds_put_tile<F16>(hbuf + bt * KF_R * 1024, w + NW * i, lane, hv);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                static_for<NFS + NFC>([&](auto FI) {
                    constexpr int fi = decltype(FI)::value;
                    constexpr bool isSkip = fi < NFS;
                    constexpr int q = isSkip ? fi : fi - NFS, G0 = isSkip ? G0S : G0C, KFq = isSkip ? KF_R : KFC;
                    constexpr int kf = (q / G0) % KFq, mt = (q / (G0 * KFq)) * G0 + q % G0;
                    constexpr int pos = isSkip ? C::P_SKIP + q : C::P_COND - PB + q;
                    constexpr int posPrev = (fi - 1 < NFS) ? C::P_SKIP + fi - 1 : C::P_COND - PB + fi - 1 - NFS;
                    frag a[1];
                    take_group<F16, PF, ws_pin, 1>(ws, pos, a);
                    if constexpr (EARLY && fi > 0) refill_group<F16, PF, 0, ws_pin, 1>(ws, rsW, posPrev, wl, 0, laneOff);
                    static_for<BT>([&](auto BI) {
                        constexpr int bt = decltype(BI)::value, m = fi * BT + bt;
                        if constexpr (isSkip) skip[bt][mt] = mma(a[0], hb[bt][kf], skip[bt][mt]);
                        else accN[bt][mt] = mma(a[0], cfCur[bt][kf], accN[bt][mt]);      // (cfCur: accumulator-file values, pinned where they are made)
                        __builtin_amdgcn_sched_barrier(0);
                        static_for_range<m * NS / NM, (m + 1) * NS / NM>(stage);
                    });
                    if constexpr (!EARLY || fi == NFS + NFC - 1) refill_group<F16, PF, 0, ws_pin, 1>(ws, rsW, pos, wl, 0, laneOff);
                    // dilated tap of layer l+2 (HBM) into the register set this layer has finished with (see the packed path below)
                    if constexpr (SKIP && fi == (NFS + NFC) * WN_REQ_AT_FEAT / 8) {
                        prefetch(t, l + 2, dl2, xpC, cdC);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if (l == L - 2) {
                    asm volatile("");                  // (keeps this a branch)
                    // pinned HERE, once per sample (agpr_pin ties its result to its operand's registers: applied at every use of a
                    // value that lives on, it costs a copy of the fragment per use -- 150 instructions per layer, measured)
#pragma unroll
                    for (int bt = 0; bt < BT; bt++)
#pragma unroll
                        for (int k = 0; k < KFC; k++) cfCur[bt][k] = agpr_operand<F16>(cfNext[bt][k]);
                }
                if constexpr (SKIP) {
                    if (dumpNow) {
                        const float* bp = biasLds + (l - 1) * C::BIAS_L + 3 * R;   // running bias sum
#pragma unroll
                        for (int bt = 0; bt < BT; bt++) {
                            if (!uvalid[bt]) continue;
#pragma unroll
                            for (int i = 0; i < STW; i++)
                                *(floatx4*)(p.skipOut + ((size_t)(l - 1) * p.maxBatch + ub[bt]) * S + (w + NW * i) * 16 + g * 4) =
                                    skip[bt][i] + *(const floatx4*)(bp + (w + NW * i) * 16 + g * 4);
                        }
                    }
                }
            } else {
                constexpr int NS = BT * HTW * 2 * 5;                  // gate stages
                constexpr int NM = SKIP ? STW * KF_R * BT : 0;        // MFMA slots
                floatx2 ea, eb, ra, rb, hp;
                floatx4 hv;
                auto stage = [&](auto SI) {
                    constexpr int s = decltype(SI)::value, pr = s / 5, st = s % 5;
                    constexpr int bt = pr / (2 * HTW), i = (pr / 2) % HTW, r = (pr & 1) * 2;
                    gate_stage<F16, st>(acc[bt][2 * i][r], acc[bt][2 * i][r + 1], acc[bt][2 * i + 1][r], acc[bt][2 * i + 1][r + 1],
                                        ea, eb, ra, rb, hp);
                    if constexpr (st == 4) {
                        hv[r] = hp[0];
                        hv[r + 1] = hp[1];
                        if constexpr (r == 2) lds_put_tile<F16>(hbuf + bt * KF_R * 1024, w + NW * i, lane, hv);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                if constexpr (SKIP) {
                    constexpr int G0 = STW >= 4 ? 4 : STW, G = take_g(G0, PF);
                    static_for<STW * KF_R / G>([&](auto GI) {           // groups of takes inside a k step (see gemm_b)
                        constexpr int gi = decltype(GI)::value, q = gi * G;      // q: first fragment of the group
                        constexpr int kf = (q / G0) % KF_R, mg = q / (G0 * KF_R), mi0 = q % G0;
                        frag a[G];
                        take_group<F16, PF, ws_pin, G>(ws, C::P_SKIP + gi * G, a);
                        constexpr bool EARLY = PF > G;      // (see gemm_b)
                        if constexpr (EARLY && gi > 0) refill_group<F16, PF, 0, ws_pin, G>(ws, rsW, C::P_SKIP + (gi - 1) * G, wl, 0, laneOff);
                        static_for<G * BT>([&](auto MI) {
                            constexpr int mi = decltype(MI)::value / BT, bt = decltype(MI)::value % BT;
                            constexpr int mt = mg * G0 + mi0 + mi, m = (q + mi) * BT + bt;
                            skip[bt][mt] = mma(a[mi], hb[bt][kf], skip[bt][mt]);
                            __builtin_amdgcn_sched_barrier(0);
                            static_for_range<m * NS / NM, (m + 1) * NS / NM>(stage);
                        });
                        if constexpr (!EARLY || gi == STW * KF_R / G - 1) refill_group<F16, PF, 0, ws_pin, G>(ws, rsW, C::P_SKIP + gi * G, wl, 0, laneOff);
                        // Dilated tap and conditioning of layer l+2 (HBM) into the register set this layer has finished
                        // with, three quarters into the skip GEMM.  VMEM returns in order per wave: the weight fragments
                        // requested behind these loads wait for them, and the first of those is taken PF takes later --
                        // from here that is behind the rest of the gate, both exchanges and the residual GEMM, the longest
                        // such stretch of a layer (issued right behind the current-tap GEMM instead: 34.2 instead of
                        // 31.7 us per sample at 8192 utterances, 41.9 instead of 39.7 at 12 288).
                        if constexpr (gi == (STW * KF_R / G) * WN_REQ_AT / 8) {
                            prefetch(t, l + 2, dl2, xpC, cdC);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    if (dumpNow) {
                        const float* bp = biasLds + (l - 1) * C::BIAS_L + 3 * R;   // running bias sum
#pragma unroll
                        for (int bt = 0; bt < BT; bt++) {
                            if (!uvalid[bt]) continue;
#pragma unroll
                            for (int i = 0; i < STW; i++)
                                *(floatx4*)(p.skipOut + ((size_t)(l - 1) * p.maxBatch + ub[bt]) * S + (w + NW * i) * 16 + g * 4) =
                                    skip[bt][i] + *(const floatx4*)(bp + (w + NW * i) * 16 + g * 4);
                        }
                    }
                } else {
                    static_for<NS>(stage);
                }
            }
            WN_TMARK(3)
            wg_barrier();   // h complete
            WN_TMARK(4)
#pragma unroll
            for (int bt = 0; bt < BT; bt++) lds_get_frags<F16, KF_R>(hbuf + bt * KF_R * 1024, lane, hb[bt]);
            if (ringInLds) {
                char* const sl = ringLds + (size_t)(dl.lds + (t & (d - 1))) * C::RING_SLOT;
#pragma unroll
                for (int i = 0; i < XPW; i++) {
                    const int k = w + NW * i;
                    if (k < KF_R) {
#pragma unroll
                        for (int bt = 0; bt < BT; bt++) *(frag*)(sl + ((bt * KF_R + k) * 64 + lane) * 16) = xkeep[bt][i];
                    }
                }
            }

            // residual accumulators start at Bres + x
            floatx4 xa[BT][HTW];
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < HTW; i++)
                    xa[bt][i] = *(const floatx4*)(bl + 2 * R + (w + NW * i) * 16 + g * 4) + x[bt][i];
            // the next layer's accumulators start at its gate bias (FEAT: + its conditioning, formed under the gate above)
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < HTW; i++) {
                    if constexpr (FEAT) {
                        acc[bt][2 * i] = accN[bt][2 * i];
                        acc[bt][2 * i + 1] = accN[bt][2 * i + 1];
                    } else {
                        acc[bt][2 * i] = *(const floatx4*)(blN + (w + NW * i) * 16 + g * 4);
                        acc[bt][2 * i + 1] = *(const floatx4*)(blN + (w + NW * i + RT) * 16 + g * 4);
                    }
                }

            // residual: x <- Wres h + Bres + x  (this wave's tiles) -> LDS
            gemm_b<F16, PF, 0, BT, HTW, KF_R>(ws, rsW, C::P_RES - PB, wl, 0, laneOff, xa, hb);
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < HTW; i++) {
                    x[bt][i] = xa[bt][i];
                    lds_put_tile<F16>(xbuf + bt * KF_R * 1024, w + NW * i, lane, xa[bt][i]);
                }
            // dilated tap of layer l+1 (layer 0 of the next sample after the last layer), requested one and a half layers
            // ago: shared through LDS with the x exchange (the readers of the previous tap passed the h barrier)
            publish_xp(xpN, dN);
            __builtin_amdgcn_sched_barrier(0);
            WN_TMARK(5)
            // + conditioning of the next layer while the x stores drain (summation order of the gate pre-activation in
            // every organisation of the engine: bias, conditioning, dilated tap, current tap).  fp16: a fragment in B
            // layout already, added by the matrix core through a 0/1 selection matrix (2 MFMAs instead of 8
            // conversions + 8 adds per fragment)
            if constexpr (FEAT) {
                // (computed under the gate above)
            } else {
#pragma unroll
                for (int bt = 0; bt < BT; bt++)
#pragma unroll
                    for (int k = 0; k < C::COND_FR; k++) cond_add<F16>(&acc[bt][k * P::TPF], cond_frag<F16, RAW>(cdN[bt], k), selA);
            }
            if (dumpNow) {
#pragma unroll
                for (int bt = 0; bt < BT; bt++) {
                    if (!uvalid[bt]) continue;
#pragma unroll
                    for (int i = 0; i < HTW; i++)
                        *(floatx4*)(p.xtOut + ((size_t)l * p.maxBatch + ub[bt]) * R + (w + NW * i) * 16 + g * 4) = x[bt][i];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            WN_TMARK(6)
            wg_barrier();   // x complete
            WN_TMARK(7)
            // the next layer's tap and x fragments are requested; the dilated-tap GEMM runs while the latter come back
            frag xp[BT][KF_R];
            const char* const tapImg = tap_image(dN, l + 1 < L ? t : t + 1);
#pragma unroll
            for (int bt = 0; bt < BT; bt++) lds_get_frags<F16, KF_R>(tapImg + bt * KF_R * 1024, lane, xp[bt]);
#pragma unroll
            for (int bt = 0; bt < BT; bt++) lds_get_frags<F16, KF_R>(xbuf + bt * KF_R * 1024, lane, xb[bt]);
            __builtin_amdgcn_sched_barrier(0);
            gemm_b<F16, PF, 0, BT, 2 * HTW, KF_R>(ws, rsW, C::P_PREV - PB, wl, 0, laneOff, acc, xp);
        };
        {
            // schedule entries of layers l, l+1, l+2 (the latter two may be layers 0, 1 of the next sample: table entries L, L+1)
            // the three schedule entries travel with the loop (scalar arithmetic: the scalar LOADS of the table in the kernel arguments
            // return out of order, so every LDS wait near one became an lgkmcnt(0); round 4)
            Dil da = dil_first(), db = dil_next(da, p.maxDilation, L == 1, ldsD), dc = dil_next(db, p.maxDilation, L == 2, ldsD);
            auto adv = [&](int l) { da = db; db = dc; dc = dil_next(dc, p.maxDilation, l + 3 == L, ldsD); };
            layer(std::false_type{}, 0, da, db, dc, xpA, cdA, xpB, cdB);
            adv(0);
            int l = 1;
            for (; l + 1 < L; l += 2) {
                layer(std::true_type{}, l, da, db, dc, xpB, cdB, xpA, cdA);
                adv(l);
                layer(std::true_type{}, l + 1, da, db, dc, xpA, cdA, xpB, cdB);
                adv(l + 1);
            }
            if (l < L) layer(std::true_type{}, l, da, db, dc, xpB, cdB, xpA, cdA);
            if (L & 1) {
                // odd layer count: layer 0 of the next sample was prefetched into the odd set
#pragma unroll
                for (int bt = 0; bt < BT; bt++) {
#pragma unroll
                    for (int i = 0; i < XPW; i++) {
                        const frag tmp = xpA[bt][i];
                        xpA[bt][i] = xpB[bt][i];
                        xpB[bt][i] = tmp;
                    }
#pragma unroll
                    for (int k = 0; k < CR; k++) {
                        const frag tmp = cdA[bt][k];
                        cdA[bt][k] = cdB[bt][k];
                        cdB[bt][k] = tmp;
                    }
                }
            }
        }
        // skip GEMM of the last layer
        gemm_b<F16, PF, 0, BT, STW, KF_R>(ws, rsW, C::P_CUR, (L - 1) * FLW, 0, laneOff, skip, hb);

        // ---- output head (nv_wavenet_reference.cpp:94-104) -----------------------------------
#pragma unroll
        for (int bt = 0; bt < BT; bt++)
#pragma unroll
            for (int i = 0; i < STW; i++) {
                floatx4 v = skip[bt][i] + *(const floatx4*)(skipBiasSum + (w + NW * i) * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = __builtin_fmaxf(v[r], 0.f);
                lds_put_tile<F16>(skbuf + bt * KF_S * 1024, w + NW * i, lane, v);
                if (dumpNow && uvalid[bt])   // the oracle applies the ReLU to the last layer's skipOut in place
                    *(floatx4*)(p.skipOut + ((size_t)(L - 1) * p.maxBatch + ub[bt]) * S + (w + NW * i) * 16 + g * 4) = v;
            }
        wg_barrier();
        floatx4 zs[BT][ATW];
        if constexpr (HS != 0 && KF_S * BT * 4 > WN_ZS_B_REGS) {
            // (the B fragments of the A x S GEMM read from their LDS image as they are needed: four tiles' 32 fragments do not fit the
            //  register file beside the accumulators)
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < ATW; i++) zs[bt][i] = *(const floatx4*)(headBias + (w + NW * i) * 16 + g * 4);
            gemm_ldsb_b<F16, PF, C::HSP, BT, ATW, KF_S>(ws, rsW, C::O_ZS, L * FLW, 0, laneOff, zs, skbuf, lane);
            skip_frags<F16, PF, C::HSP, ws_pin, C::PAD1>(ws, rsW, C::FW_ZS, L * FLW, 0, laneOff);   // (zero fragments)
        } else {
            frag sb[BT][KF_S];
#pragma unroll
            for (int bt = 0; bt < BT; bt++) {
                lds_get_frags<F16, KF_S>(skbuf + bt * KF_S * 1024, lane, sb[bt]);
#pragma unroll
                for (int i = 0; i < ATW; i++) zs[bt][i] = *(const floatx4*)(headBias + (w + NW * i) * 16 + g * 4);
            }
            if constexpr (HS == 0) gemm_res<F16, BT, ATW, KF_S>(hw, 0, zs, sb);
            else {
                gemm_b<F16, PF, C::HSP, BT, ATW, KF_S>(ws, rsW, C::O_ZS, L * FLW, 0, laneOff, zs, sb);
                skip_frags<F16, PF, C::HSP, ws_pin, C::PAD1>(ws, rsW, C::FW_ZS, L * FLW, 0, laneOff);   // (zero fragments)
            }
        }
#pragma unroll
        for (int bt = 0; bt < BT; bt++)
#pragma unroll
            for (int i = 0; i < ATW; i++) {
#pragma unroll
                for (int r = 0; r < 4; r++) zs[bt][i][r] = __builtin_fmaxf(zs[bt][i][r], 0.f);
                lds_put_tile<F16>(zsbuf + bt * KF_A * 1024, w + NW * i, lane, zs[bt][i]);
                if (dumpNow && uvalid[bt])
                    *(floatx4*)(p.zs + (size_t)ub[bt] * A + (w + NW * i) * 16 + g * 4) = zs[bt][i];
            }
        wg_barrier();
        floatx4 za[BT][ATW];
        {
#pragma unroll
            for (int bt = 0; bt < BT; bt++)
#pragma unroll
                for (int i = 0; i < ATW; i++)
                    za[bt][i] = *(const floatx4*)(headBias + A + (w + NW * i) * 16 + g * 4);
            auto draw_selectors = [&]() {
                // In-kernel selectors (Philox4x32-10, counter {sample, utterance}).  A wave instruction costs the same whether
                // its lanes compute one value or sixteen different ones: with 16 softmax lanes per utterance (one DPP row),
                // lane q < BT of every row draws the selector of tile q and the row takes it over with a row broadcast -- one
                // Philox evaluation per sample instead of one per tile.  Placed here, its arithmetic fills the wait for the zs
                // fragments instead of the head of the sample.
                if (!p.useRng) return;
                if constexpr (C::LPU == 16 && BT > 1) {
                    const int q = sq < BT ? sq : 0;
                    int sb = (tile0 + q) * 16 + su;
                    sb = sb < p.batch ? sb : p.batch - 1;
                    const float mine = philox_selector(p.rngKey0, p.rngKey1, (unsigned)t, (unsigned)sb);
                    selv[0] = dpp_f<0x150>(mine);                       // row_newbcast:0
                    if constexpr (BT > 1) selv[1] = dpp_f<0x151>(mine);
                    if constexpr (BT > 2) selv[2] = dpp_f<0x152>(mine);
                    if constexpr (BT > 3) selv[3] = dpp_f<0x153>(mine);
                } else {
#pragma unroll
                    for (int bt = 0; bt < BT; bt++) {
                        int sb = (tile0 + bt) * 16 + su;
                        sb = sb < p.batch ? sb : p.batch - 1;
                        selv[bt] = philox_selector(p.rngKey0, p.rngKey1, (unsigned)t, (unsigned)sb);
                    }
                }
            };
            if constexpr (C::ZA_B_FROM_LDS) {
                static_assert(HR == 0, "the LDS-streamed head is for the large, non-resident heads");
                draw_selectors();
                gemm_ldsb_b<F16, PF, C::HSP, BT, ATW, KF_A>(ws, rsW, C::O_ZA, L * FLW, 0, laneOff, za, zsbuf, lane);
            } else {
                frag zb[BT][KF_A];
#pragma unroll
                for (int bt = 0; bt < BT; bt++) lds_get_frags<F16, KF_A>(zsbuf + bt * KF_A * 1024, lane, zb[bt]);
                draw_selectors();
                if constexpr (HR >= C::FW_ZA) gemm_res<F16, BT, ATW, KF_A>(hw, C::FW_ZS - HS, za, zb);
                else gemm_b<F16, PF, C::HSP, BT, ATW, KF_A>(ws, rsW, C::O_ZA, L * FLW, 0, laneOff, za, zb);
            }
        }
        if constexpr (C::ALIAS_LG) wg_barrier();   // every wave is done with the zs image
        // logits -> LDS [utt][row] (row stride padded by 4 floats: conflict-free b128 writes), LGT tiles per softmax pass
        constexpr int LGT = C::LGT;
        auto put_logits = [&](auto PASS) {
            constexpr int pass = decltype(PASS)::value;
#pragma unroll
            for (int bl = 0; bl < LGT; bl++)
#pragma unroll
                for (int i = 0; i < ATW; i++) {
                    const int bt = pass * LGT + bl;
                    *(floatx4*)(lgbuf + (bl * 16 + j) * C::LROW + (w + NW * i) * 16 + g * 4) = za[bt][i];
                    if (dumpNow && uvalid[bt])
                        *(floatx4*)(p.za + (size_t)ub[bt] * A + (w + NW * i) * 16 + g * 4) = za[bt][i];
                }
        };
        put_logits(std::integral_constant<int, 0>{});
        WN_TMARK(8)
        if constexpr (HS == FHW) {
            skip_frags<F16, PF, C::HSP, ws_pin, C::PAD2>(ws, rsW, C::O_ZA + C::FW_ZA, L * FLW, 0, laneOff);   // (zero fragments)
        }
        // FEAT: the features of sample t+2 (HBM) are requested HERE, behind the last take of the sample: loads return in order, and in
        // front of the head's weight refills they would hold those up for an HBM round trip; their reader is the move behind layer
        // L-2 of the next sample (the last reader of cfNext was that move of this sample)
        if constexpr (FEAT) load_feat(t + 2, cfNext);
        wg_barrier();
        WN_TMARK(9)

        // ---- softmax + inverse-CDF pick: LPU lanes per utterance, RPL rows per lane (softmax_pick) ----
        static_for<BT / LGT>([&](auto PASS) {
            constexpr int pass = decltype(PASS)::value;
            if constexpr (pass > 0) {
                wg_barrier();          // the previous pass's logits have been read by everyone
                put_logits(PASS);
                wg_barrier();
            }
#pragma unroll
            for (int bl = 0; bl < LGT; bl++) {
                const int bt = pass * LGT + bl;
                float e[C::RPL];
                float total;
                const float* lrow = lgbuf + (bl * 16 + su) * C::LROW + sq * C::RPL;
                const int pick = softmax_pick<A, C::LPU, C::RPL>(lrow, sq, lane, selv[bt], e, total);
                const int sb = (tile0 + bt) * 16 + su;
                if (sq == 0) {
                    ybuf[bt * 16 + su] = pick;
                    if (sb < p.batch) p.yOut[(size_t)sb * p.numSamples + t] = pick;
                }
                if (dumpNow && sb < p.batch) {
                    const float inv = 1.0f / total;
#pragma unroll
                    for (int i = 0; i < C::RPL / 4; i++)
                        *(floatx4*)(p.p + (size_t)sb * A + sq * C::RPL + i * 4) =
                            floatx4{e[i * 4] * inv, e[i * 4 + 1] * inv, e[i * 4 + 2] * inv, e[i * 4 + 3] * inv};
                }
            }
        });
        wg_barrier();
#pragma unroll
        for (int bt = 0; bt < BT; bt++) {
            yPrev[bt] = yCur[bt];
            yCur[bt] = ybuf[bt * 16 + j];
        }
        // ybuf / lgbuf are next written after several barriers of the next sample
        WN_TMARK(10)
    }
#ifdef WN_TIMING
    if (tid == 0 && blockIdx.x == 0)
        for (int i = 0; i < 12; i++) p.p[i] = (float)tacc[i];
#endif

    ring_lds_copy(std::false_type{});      // (behind the last sample's closing barrier: every deferred slot store has landed)
    if (w == 0 && g == 0) {
#pragma unroll
        for (int bt = 0; bt < BT; bt++)
            if (uvalid[bt]) {
                p.yInPrev[ub[bt]] = yPrev[bt];
                p.yInCur[ub[bt]] = yCur[bt];
            }
    }
    if (p.clk != nullptr && blockIdx.x == 0 && tid == 0) {
        p.clk[2] = __builtin_amdgcn_s_memtime();
        p.clk[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// mu-law expansion of the generated indices to int16 PCM (pytorch/utils.py:62-70 +
// inference.py:58-60), columns [first, first+count) of the [batch][numSamples] buffers.  The value
// depends on the index alone: a table of A entries computed on the host in float64 like numpy.
static __global__ void mulaw_pcm_kernel(const int* __restrict__ yOut, short* __restrict__ pcm,
                                        const short* __restrict__ table, int batch, int numSamples, int first,
                                        int count) {
    const size_t n = (size_t)batch * count;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t at = (i / count) * numSamples + first + (i % count);
        pcm[at] = table[yOut[at]];
    }
}

// ------------------------------------------------------------------------------------------
// pack kernels (setup only; role of nv_wavenet_conversions.cuh + matrix_math.cuh:55-64)
// ------------------------------------------------------------------------------------------

// fp32 col-major M x K -> per-wave fragment streams.  Wave w gets the tiles t = w + NW*i in
// order (gateRT=0), or for the gated 2R x R matrices the pairs (t, t+RT) (gateRT=RT>0).
// dst + w*waveStride is the start of this matrix inside wave w's stream; idx = element of the packed matrix.
template <bool F16>
WN_DEV void pack_weight_elem(typename Prec<F16>::elem* __restrict__ dst, const float* __restrict__ src, int M, int K,
                             int NW, size_t waveStride, int gateRT, size_t idx) {
    constexpr int EPL = Prec<F16>::EPL, TPF = Prec<F16>::TPF;
    const int KF = K / (16 * TPF);
    const int tilesPerWave = M / 16 / NW;
    const size_t perWave = (size_t)tilesPerWave * KF * 64 * EPL;
    const int w = idx / perWave;
    size_t r = idx % perWave;
    const int e = r % EPL; r /= EPL;
    const int lane = r % 64; r /= 64;
    // slot order inside a matrix: groups of G slots, k-fragment-major inside a group (see gemm())
    const int G = tilesPerWave >= 4 ? 4 : tilesPerWave;
    const int mi = r % G;
    const int kf = (r / G) % KF;
    const int it = (r / (G * KF)) * G + mi;        // tile slot inside the wave's list
    int tile;
    if (gateRT > 0) tile = w + NW * (it >> 1) + (it & 1) * gateRT;
    else tile = w + NW * it;
    const int i = lane & 15, g = lane >> 4;
    const int m = tile * 16 + i;
    const int k = (kf * TPF + (e >> 2)) * 16 + g * 4 + (e & 3);
    float v = src[(size_t)m + (size_t)k * M];
    if (gateRT > 0) v *= gate_prescale<F16>(m >= M / 2);    // gated 2R x R matrix: see gate1()
    dst[(size_t)w * waveStride + (idx % perWave)] = (typename Prec<F16>::elem)v;
}
template <bool F16>
__global__ void pack_weight_kernel(typename Prec<F16>::elem* __restrict__ dst, const float* __restrict__ src, int M,
                                   int K, int NW, size_t waveStride, int gateRT) {
    const size_t n = (size_t)M * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x)
        pack_weight_elem<F16>(dst, src, M, K, NW, waveStride, gateRT, idx);
}

// One layer in one launch (setLayerWeights): the four matrices into the per-wave streams and the three
// bias vectors into the fp32 table (gate biases pre-scaled like the gate matrices).
struct LayerSrc {
    const float *Wprev, *Wcur, *Bh, *Wres, *Bres, *Wskip, *Bskip;
};
template <bool F16>
__global__ void pack_layer_kernel(typename Prec<F16>::elem* __restrict__ layerFrags, float* __restrict__ biasL, LayerSrc s,
                                  int R, int S, int NW, size_t waveStride, int oPrev, int oCur, int oRes, int oSkip) {
    constexpr int FE = 64 * Prec<F16>::EPL;
    const size_t nGate = (size_t)2 * R * R, nRes = (size_t)R * R, nSkip = (size_t)S * R;
    const size_t nW = 2 * nGate + nRes + nSkip, n = nW + 3 * R + S;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < nGate) pack_weight_elem<F16>(layerFrags + (size_t)oPrev * FE, s.Wprev, 2 * R, R, NW, waveStride, R / 16, idx);
        else if (idx < 2 * nGate) pack_weight_elem<F16>(layerFrags + (size_t)oCur * FE, s.Wcur, 2 * R, R, NW, waveStride, R / 16, idx - nGate);
        else if (idx < 2 * nGate + nRes) pack_weight_elem<F16>(layerFrags + (size_t)oRes * FE, s.Wres, R, R, NW, waveStride, 0, idx - 2 * nGate);
        else if (idx < nW) pack_weight_elem<F16>(layerFrags + (size_t)oSkip * FE, s.Wskip, S, R, NW, waveStride, 0, idx - 2 * nGate - nRes);
        else {
            const int i = (int)(idx - nW);
            if (i < 2 * R) biasL[i] = s.Bh[i] * gate_prescale<F16>(i >= R);
            else if (i < 3 * R) biasL[i] = s.Bres[i - 2 * R];
            else biasL[i] = s.Bskip[i - 3 * R];
        }
    }
}

// fp32 -> T_data, same layout
template <bool F16>
__global__ void convert_kernel(typename Prec<F16>::elem* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (typename Prec<F16>::elem)src[i];
}

// conditioning: fp32 [rows = samples*L][maxBatch][2R] -> T_data fragments, gate rows pre-scaled.
//   [rows][tiles][wave][COND_FR][lane][EPL];
//       fragment c, element e of wave w: gate slot it = c*TPF + (e>>2) -> tile = w + NW*(it>>1) + (it&1)*RT
// One workgroup per (row, tile of 16 utterances): the 16 x 2R fp32 source block is contiguous (8 KB at
// R = 64) and is read with coalesced 16-byte loads into LDS; the 2R*16 destination elements are
// contiguous too and are written as one 16-byte piece per thread.  (The first version gathered one
// scalar per thread straight from global memory: 2.2-4.5x read amplification, 19 ms for 256 samples x
// 8192 utterances; this one moves source + destination bytes once.)
template <bool F16, int R>
__global__ __launch_bounds__(256) void pack_cond_tiled_kernel(typename Prec<F16>::elem* __restrict__ dst,
                                                              const float* __restrict__ src, size_t rows, int maxBatch,
                                                              int tiles) {
    using elem = typename Prec<F16>::elem;
    using frag = typename Prec<F16>::frag;
    constexpr int EPL = Prec<F16>::EPL, TPF = Prec<F16>::TPF;
    constexpr int R2 = 2 * R, RT = R / 16, NW = RT >= 4 ? 4 : RT;
    constexpr int COND_FR = 2 * (RT / NW) / TPF;
    constexpr int LROWF = R2 + 4;                  // padded LDS row (floats)
    __shared__ __attribute__((aligned(16))) float blk[16 * LROWF];
    const size_t nblk = rows * (size_t)tiles;
    for (size_t bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
        const size_t row = bi / tiles;
        const int b0 = (int)(bi % tiles) * 16;
        const float* s = src + (row * maxBatch + b0) * R2;
        for (int i = threadIdx.x; i < 16 * R2 / 4; i += 256) {
            const int jj = i / (R2 / 4), c4 = i % (R2 / 4);
            floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
            if (b0 + jj < maxBatch) v = __builtin_nontemporal_load((const floatx4*)(s + (size_t)jj * R2 + c4 * 4));
            *(floatx4*)(blk + jj * LROWF + c4 * 4) = v;
        }
        __syncthreads();
        elem* d = dst + bi * (size_t)(16 * R2);
        for (int pi = threadIdx.x; pi < 16 * R2 / EPL; pi += 256) {
            const int lane = pi & 63, fr = pi >> 6;
            const int j = lane & 15, g = lane >> 4;
            frag o;
#pragma unroll
            for (int q = 0; q < EPL / 4; q++) {
                const int w = fr / COND_FR, c = fr % COND_FR;
                const int it = c * TPF + q;
                const int tile16 = w + NW * (it >> 1) + (it & 1) * RT;
                const int ch = tile16 * 16 + g * 4;
                const floatx4 v = *(const floatx4*)(blk + j * LROWF + ch);
                const float sc = gate_prescale<F16>(ch >= R);
                if constexpr (F16) {
                    const unsigned lo = scale_pair_f32(v[0], v[1], sc), hi = scale_pair_f32(v[2], v[3], sc);
                    const half4 h = __builtin_bit_cast(half4, uintx2{lo, hi});
#pragma unroll
                    for (int r = 0; r < 4; r++) o[q * 4 + r] = h[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) o[q * 4 + r] = (elem)(v[r] * sc);
                }
            }
            __builtin_nontemporal_store(o, (frag*)(d + (size_t)pi * EPL));
        }
        __syncthreads();
    }
}

// ---- in-kernel conditioning (RAW = 3): setup kernels -----------------------------------------------------------------------
// The model's conditioning convolution weight, [L][2R][nCond] (= cond_layers.weight [2R*L][nCond][1], pytorch/wavenet.py:73-74),
// -> per layer col-major [KC][2R] fp32, zero-padded to KC channels: what pack_weight_kernel takes
static __global__ void cond_weight_arrange_kernel(float* __restrict__ dst, const float* __restrict__ src, int L, int R2, int nCond, int KC) {
    const size_t n = (size_t)L * R2 * nCond;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nCond), r = (int)((i / nCond) % R2), l = (int)(i / ((size_t)nCond * R2));
        dst[((size_t)l * KC + c) * R2 + r] = src[i];
    }
}
// fragments of the plain stream -> their places in the stream that also carries the conditioning weights (map: {from, to} per fragment
// of ONE wave's stream; the same for every wave)
static __global__ void restream_kernel(uintx4* __restrict__ dst, const uintx4* __restrict__ src, const int2* __restrict__ map, int nmap,
                                       size_t srcWaveFrags, size_t dstWaveFrags, int NW) {
    const size_t n = (size_t)NW * nmap * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), m = (int)((i >> 6) % nmap), w = (int)((i >> 6) / nmap);
        dst[((size_t)w * dstWaveFrags + map[m].y) * 64 + lane] = src[((size_t)w * srcWaveFrags + map[m].x) * 64 + lane];
    }
}
// bias table of that stream: the plain one with bcond (pre-scaled like Bh) added to the gate biases
template <bool F16>
__global__ void feat_bias_kernel(float* __restrict__ dst, const float* __restrict__ bias, const float* __restrict__ bcond, int L, int R,
                                 int biasL, int total) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float v = bias[i];
        const int l = i / biasL, k = i % biasL;
        if (l < L && k < 2 * R) v += bcond[l * 2 * R + k] * gate_prescale<F16>(k >= R);
        dst[i] = v;
    }
}
// Upsampled features of any strided [utterance][channel][sample] tensor (fp32 or fp16 elements) -> B fragments
//   dst [sample][tiles][KFC][64 lanes][EPL]: fragment kf, lane (g, j), element e = channel (kf*TPF + (e>>2))*16 + 4g + (e&3) of
//   utterance tile*16 + j; channels >= nCond and utterances >= batch are zero.
// One workgroup per (tile, TB samples): gathered into an LDS image of the fragments, written as 16-byte pieces.
template <bool F16>
__global__ __launch_bounds__(256) void pack_features_kernel(typename Prec<F16>::elem* __restrict__ dst, const void* __restrict__ src, int srcBits,
                                                            long long bS, long long cS, long long tS, int nCond, int batch, int count,
                                                            int tiles, int tilesUsed) {
    using elem = typename Prec<F16>::elem;
    constexpr int KFC = feat_kfc<F16>(), TPF = Prec<F16>::TPF, EPL = Prec<F16>::EPL, KC = KFC * 16 * TPF, TB = 8;
    __shared__ __attribute__((aligned(16))) elem img[TB * KFC * 64 * EPL];
    const int tblocks = (count + TB - 1) / TB;
    const size_t nblk = (size_t)tilesUsed * tblocks;
    for (size_t bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
        const int tile = (int)(bi % tilesUsed), t0 = (int)(bi / tilesUsed) * TB;
        for (int i = threadIdx.x; i < 16 * KC * TB; i += 256) {
            int j, c, tt;
            if (tS == 1) { tt = i % TB; c = (i / TB) % KC; j = i / (TB * KC); }      // (the contiguous axis of the source innermost)
            else { c = i % KC; j = (i / KC) % 16; tt = i / (KC * 16); }
            const int b = tile * 16 + j, t = t0 + tt;
            float v = 0.f;
            if (c < nCond && b < batch && t < count) {
                const long long at = b * bS + c * cS + t * tS;
                v = srcBits == 16 ? (float)((const _Float16*)src)[at] : ((const float*)src)[at];
            }
            const int kf = c / (16 * TPF), tk = (c / 16) % TPF, g = (c % 16) / 4, r = c % 4;
            img[((tt * KFC + kf) * 64 + g * 16 + j) * EPL + tk * 4 + r] = (elem)v;
        }
        __syncthreads();
        for (int pi = threadIdx.x; pi < TB * KFC * 64; pi += 256) {
            const int tt = pi / (KFC * 64), rem = pi % (KFC * 64);
            if (t0 + tt < count)
                *(uintx4*)(dst + (((size_t)(t0 + tt) * tiles + tile) * KFC * 64 + rem) * EPL) = *(const uintx4*)(img + (size_t)pi * EPL);
        }
        __syncthreads();
    }
}

// ---- the upsampling half of the model's conditioning path on the engine's own kernels (round 5) ------------------------------
// WaveNet.get_cond_input (pytorch/wavenet.py:190-202) = ConvTranspose1d(n_cond, n_cond, window, stride) + trimming of its tail + the
// 1x1 `cond_layers`; the latter is computed by wavenet_wg<.., RAW=3>, this is the former, straight into the feature fragments that
// kernel reads:   c[t][co] = b[co] + sum_{j < m} sum_ci mel[ci][f - j] W[ci][co][j*stride + r],   t = f*stride + r, m = window / stride
// (frames f - j < 0 contribute nothing; the trimmed output has frames * stride samples).  Per phase r this is a GEMM
// [n_cond x m*n_cond] x [m*n_cond x columns] whose B operand for (frame f, tile) are the MEL fragments of frames f .. f-m+1 -- mel in
// the same fragment order as the features (pack_features_kernel with frames for samples) -- and whose result tiles (MFMA D layout)
// ARE feature fragments once converted to T_data.  A workgroup takes a phase: its operand A_r (RTU x m*KFC fragments) sits in LDS,
// its four waves share the columns.
constexpr int kUpRowTiles = (kCondChannelsMax + 15) / 16;
// phases (PB) and columns (CB) a wave of upsample_features_kernel takes per pass, waves per workgroup.  One phase, four columns, four waves
// ships: 0.30 ms per chunk of 256 samples x 12 288 utterances (0.33 before the next tap's mel fragments were requested under the current
// tap's MFMAs) -- 0.60 GB of feature fragments written to HBM (its floor: ~0.15 ms) and 2.4 GB of mel fragments read through L2.  Two
// phases' operands side by side in LDS (120 KiB at four taps, one workgroup per CU) halve that L2 traffic and were measured slower both
// with four waves per workgroup (0.43 ms) and with eight (0.36-0.37 ms, same GPU call); so were eight-wave workgroups of two columns
// (0.355 ms at two waves per SIMD, 0.370 forced to four: 26 spilled registers).
template <bool F16> constexpr int up_phases() { return WN_UP_PHASES(F16); }
template <bool F16> constexpr int up_cols() { return WN_UP_COLS(F16); }
template <bool F16> constexpr int up_waves() { return WN_UP_WAVES(F16); }
// table of the A operands: [stride][kUpRowTiles][m * KFC][64 lanes][EPL] from the ConvTranspose1d weight [n_cond][n_cond][window]
template <bool F16>
__global__ void pack_upsample_kernel(typename Prec<F16>::elem* __restrict__ dst, const float* __restrict__ upW, int nCond, int window, int stride) {
    constexpr int KFC = feat_kfc<F16>(), TPF = Prec<F16>::TPF, EPL = Prec<F16>::EPL;
    const int m = window / stride;
    const size_t n = (size_t)stride * kUpRowTiles * m * KFC * 64 * EPL;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        size_t q = idx;
        const int e = (int)(q % EPL); q /= EPL;
        const int lane = (int)(q % 64); q /= 64;
        const int kfu = (int)(q % (m * KFC)); q /= (size_t)(m * KFC);
        const int tr = (int)(q % kUpRowTiles);
        const int r = (int)(q / kUpRowTiles);
        const int j = kfu / KFC, kf = kfu % KFC, i = lane & 15, g = lane >> 4;
        const int co = tr * 16 + i, ci = (kf * TPF + (e >> 2)) * 16 + 4 * g + (e & 3);
        const float v = (co < nCond && ci < nCond) ? upW[((size_t)ci * nCond + co) * window + j * stride + r] : 0.f;
        dst[idx] = (typename Prec<F16>::elem)v;
    }
}
template <bool F16>
__global__ __launch_bounds__(64 * up_waves<F16>()) WN_UP_ATTR void upsample_features_kernel(typename Prec<F16>::elem* __restrict__ feat, const typename Prec<F16>::elem* __restrict__ melfrag,
                                                                const typename Prec<F16>::elem* __restrict__ tab, const float* __restrict__ bias, int m,
                                                                int stride, int tiles, int tilesUsed, int firstSample, int count) {
    using P = Prec<F16>;
    using frag = typename P::frag;
    constexpr int KFC = feat_kfc<F16>(), EPL = P::EPL, RTU = kUpRowTiles;
    // A workgroup takes PB phases (their operands side by side in LDS: PB x RTU x m*KFC KiB) and a wave CB columns at a time: a mel
    // fragment loaded from L2 feeds all PB phases, an operand fragment read from LDS feeds CB columns.  (One phase and one column per
    // pass: bound by the LDS, 0.40 ms per chunk of 256 samples x 12 288 utterances; one phase, four columns: 0.33 ms, what ships.)
    constexpr int CB = up_cols<F16>(), PB = up_phases<F16>(), NWU = up_waves<F16>(), NTH = 64 * NWU;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4;
    const int nA = RTU * m * KFC;                                     // fragments of one phase's operand
    const int npair = (stride + PB - 1) / PB;
    for (int pp = blockIdx.x; pp < npair; pp += gridDim.x) {
        const int r0 = pp * PB;
        __syncthreads();                                              // (the previous pair's readers are done)
        for (int ph = 0; ph < PB; ph++) {
            if (r0 + ph >= stride) break;
            const uintx4* src = (const uintx4*)(tab + (size_t)(r0 + ph) * nA * 64 * EPL);
            for (int i = tid; i < nA * 64; i += NTH) ((uintx4*)lds)[(size_t)ph * nA * 64 + i] = src[i];
        }
        __syncthreads();
        // frames f with a sample f*stride + r, r in the pair, inside [firstSample, firstSample + count)
        int fLo = (firstSample - (r0 + PB - 1) + stride - 1) / stride;
        if (fLo < 0) fLo = 0;
        const int last = firstSample + count - 1 - r0;
        if (last < 0) continue;
        const int fHi = last / stride;
        const int ncol = (fHi - fLo + 1) * tilesUsed;
        const int ngrp = (ncol + CB - 1) / CB;
        for (int grp = blockIdx.y * NWU + w; grp < ngrp; grp += gridDim.y * NWU) {
            int fcol[CB], tcol[CB];
#pragma unroll
            for (int c = 0; c < CB; c++) {
                const int col = grp * CB + c < ncol ? grp * CB + c : ncol - 1;      // (a partial group repeats its last column)
                fcol[c] = fLo + col / tilesUsed;
                tcol[c] = col % tilesUsed;
            }
            floatx4 acc[PB][CB][RTU];
#pragma unroll
            for (int ph = 0; ph < PB; ph++)
#pragma unroll
                for (int c = 0; c < CB; c++)
#pragma unroll
                    for (int tr = 0; tr < RTU; tr++) acc[ph][c][tr] = *(const floatx4*)(bias + tr * 16 + g * 4);
            // tap j's mel fragments (L2) are requested while tap j-1's MFMAs run: two register sets, the tap loop unrolled by two
            auto load_b = [&](frag (&b)[CB][KFC], const int j) {
#pragma unroll
                for (int c = 0; c < CB; c++) {
                    const int fj = fcol[c] - j;
                    const char* mf = (const char*)(melfrag + ((size_t)(fj < 0 ? 0 : fj) * tiles + tcol[c]) * KFC * 64 * EPL);
#pragma unroll
                    for (int kf = 0; kf < KFC; kf++) {
                        b[c][kf] = *(const frag*)(mf + ((size_t)kf * 64 + lane) * 16);
                        if (fj < 0) {                                   // (frames before the first contribute nothing)
#pragma unroll
                            for (int q = 0; q < EPL; q++) b[c][kf][q] = (typename P::elem)0.f;
                        }
                    }
                }
            };
            auto taps = [&](const frag (&b)[CB][KFC], const int j) {
#pragma unroll
                for (int ph = 0; ph < PB; ph++)
#pragma unroll
                    for (int kf = 0; kf < KFC; kf++)
#pragma unroll
                        for (int tr = 0; tr < RTU; tr++) {
                            const frag a = *(const frag*)(lds + ((size_t)ph * nA + (size_t)((tr * m + j) * KFC + kf)) * 1024 + (size_t)lane * 16);
#pragma unroll
                            for (int c = 0; c < CB; c++) acc[ph][c][tr] = mma(a, b[c][kf], acc[ph][c][tr]);
                        }
            };
            frag b0[CB][KFC], b1[CB][KFC];
            load_b(b0, 0);
            for (int j = 0; j < m; j += 2) {
                if (j + 1 < m) load_b(b1, j + 1);
                taps(b0, j);
                if (j + 1 < m) {
                    if (j + 2 < m) load_b(b0, j + 2);
                    taps(b1, j + 1);
                }
            }
#pragma unroll
            for (int ph = 0; ph < PB; ph++)
#pragma unroll
                for (int c = 0; c < CB; c++) {
                    const int t = fcol[c] * stride + r0 + ph;
                    if (grp * CB + c >= ncol || r0 + ph >= stride || t < firstSample || t >= firstSample + count) continue;
                    typename P::elem* out = feat + ((size_t)t * tiles + tcol[c]) * KFC * 64 * EPL;
#pragma unroll
                    for (int kf = 0; kf < KFC; kf++) {
                        frag o;
                        if constexpr (F16) {
                            const floatx4 lo = acc[ph][c][2 * kf], hi = (2 * kf + 1 < RTU) ? acc[ph][c][2 * kf + 1 < RTU ? 2 * kf + 1 : 0] : floatx4{0.f, 0.f, 0.f, 0.f};
                            o = half8{(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3], (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2],
                                      (_Float16)hi[3]};
                        } else {
                            o = acc[ph][c][kf < RTU ? kf : 0];
                        }
                        *(frag*)((char*)out + ((size_t)kf * 64 + lane) * 16) = o;
                    }
                }
        }
    }
}

static __global__ void silence_kernel(int* yInPrev, int* yInCur, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        yInPrev[i] = 128;   // mu-law silence, nv_wavenet.cuh:213-218
        yInCur[i] = 128;
    }
}

}  // namespace wn
