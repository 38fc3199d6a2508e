// wn_stream.hpp -- the THROUGHPUT kernel of the engine (large batches): loader / consumer waves.
//
// wn_kernels.hpp's wavenet_wg splits one 16-utterance tile over the 4 SIMDs of a CU, which is the
// lowest-latency organisation but leaves every SIMD waiting on two LDS exchanges per layer.  When
// there are at least as many tiles as SIMDs, the better organisation is the opposite one:
//
//   * one workgroup = 8 wavefronts on one CU: 4 CONSUMER waves (one per SIMD), each generating
//     its OWN tile of 16 utterances with the whole network in its registers -- the MFMA result
//     tile is the next MFMA's B operand (see wn_kernels.hpp), so a consumer needs no LDS exchange,
//     no barrier for its activations, and the softmax is a 4-lane shuffle;
//   * 4 LOADER waves (the second wave of each SIMD) stream the weights ONCE per sample for all
//     four tiles: global -> LDS with LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) into a
//     ring of NS chunks of CH fragments; every consumer reads every fragment from LDS
//     (ds_read_b128, 1 KiB per wave-instruction, conflict-free) straight into the MFMA;
//   * one bare s_barrier per chunk is the only synchronisation.  After barrier #c the consumers may
//     read chunks <= c+1 (they read RA fragments ahead of the MFMA, across the chunk boundary, so
//     the LDS latency never shows), and the loaders may overwrite the slot of chunk c-2.  Loaders
//     wait with counted vmcnt so that their newest chunk stays in flight across the barrier;
//     consumers need no wait at all at the barrier (the slot being recycled was fully consumed a
//     whole chunk earlier).  This needs a ring of >= 5 chunks;
//   * a single-wave VMEM stream is capped at ~30 GB/s (scripts/ubench/stream.hip) while the
//     LDS-DMA ring delivers 1.73 MB in 11.7 us to all four consumers (scripts/ubench/ldsring.hip).
//
// Weight stream (one, shared): [layer 0 .. L-1][head], fragments in consumption order
//   layer l: prev (2R x R) | cur (2R x R) | skip of layer l-1 (S x R) | res (R x R), padded
//   head   : skip of layer L-1 | zs (A x S) | za (A x A, rows permuted: lane g owns rows g*A/4..)
// Conditioning: [sample][layer][tile][fragment] in MFMA D-tile order; ring: [tile][slot][fragment].
#pragma once

#include <algorithm>

#include "wn_kernels.hpp"

namespace wn {

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int pick_chunk(int flp, int mx) {
    int best = 4;
    for (int d = 4; d <= mx && d <= flp; d += 4)
        if (flp % d == 0) best = d;
    return best;
}

template <bool F16, int R, int S, int A>
struct SCfg {
    using P = Prec<F16>;
    static constexpr int TPF = P::TPF, EPL = P::EPL;
    static constexpr int RT = R / 16, R2T = 2 * R / 16, ST = S / 16, AT = A / 16;
    static constexpr int KF_R = RT / TPF, KF_S = ST / TPF, KF_A = AT / TPF;
    static constexpr int F_GATE = R2T * KF_R, F_RES = RT * KF_R, F_SKIP = ST * KF_R;
    // body of layer l: prev(l) | cur(l) | skip(l-1) | res(l).  The skip GEMM of a layer is consumed one
    // body later, interleaved with the gate's VALU work (it depends on neither), so the matrix pipe
    // runs under the transcendental math.  Body 0's skip slot is zeros; skip(L-1) opens the head body.
    static constexpr int O_PREV = 0, O_CUR = F_GATE, O_SKIP = 2 * F_GATE, O_RES = O_SKIP + F_SKIP;
    static constexpr int FL = O_RES + F_RES;             // fragments per layer
    static constexpr int FLP = (FL + 7) / 8 * 8;         // padded: splits into chunks of 4k, multiple of the read-ahead
    static constexpr int CH = pick_chunk(FLP, 32);       // fragments per LDS ring chunk
    static constexpr int Q = CH / 4;                     // fragments per loader wave per chunk
    static constexpr int NCH_L = FLP / CH;               // chunks per layer
    static constexpr int F_ZS = AT * KF_S, F_ZA = AT * KF_A;
    static constexpr int H_SKIP = 0, H_ZS = F_SKIP, H_ZA = F_SKIP + F_ZS, FH = H_ZA + F_ZA;   // head body
    static constexpr int FHP = (FH + CH - 1) / CH * CH;
    static constexpr int NCH_H = FHP / CH;
    static constexpr int FRAG_ELEMS = 64 * EPL;
    static constexpr int BIAS_L = 3 * R + S;
    // LDS read-ahead depth of a consumer (fragments in flight LDS -> VGPR): divides both bodies so
    // that the register ring index is static in the rolled layer loop
    static constexpr int RA = cgcd(cgcd(cgcd(FL, FH), CH), 8);       // 1, 2, 4 or 8
    static_assert(CH % 4 == 0 && (FLP - FL) % RA == 0 && (FHP - FH) % RA == 0 && CH % RA == 0, "stream padding vs read-ahead");
    static constexpr int MIN_NS = 5;                     // ring slots needed by the barrier protocol
    static constexpr int COND_FR = R2T / TPF;            // conditioning fragments per (sample,layer,tile)
    static constexpr int ZA_REGS = A / 4;
    static_assert(A % 64 == 0, "A must be a multiple of 64");
    __host__ __device__ static size_t streamFrags(int L) { return (size_t)L * FLP + FHP; }
    static size_t ldsBytes(int L, int NS) { return (size_t)NS * CH * 1024 + ((size_t)L * BIAS_L + 2 * A) * sizeof(float); }
};

// D tiles (fp32) -> B-operand fragments (in-register, no LDS)
template <int KT> WN_DEV void to_bfrags(const floatx4 (&t)[KT], half8 (&b)[KT / 2]) {
#pragma unroll
    for (int k = 0; k < KT / 2; k++) {
        half8 r;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            r[e] = (_Float16)t[2 * k][e];
            r[4 + e] = (_Float16)t[2 * k + 1][e];
        }
        b[k] = r;
    }
}
template <int KT> WN_DEV void to_bfrags(const floatx4 (&t)[KT], floatx4 (&b)[KT]) {
#pragma unroll
    for (int k = 0; k < KT; k++) b[k] = t[k];
}

// DUMP: see wn::wavenet_wg -- production launches use the variant without any activation-dump code
// (57.5 instead of 62.3 us per sample at batch 16384).
template <bool F16, int R, int S, int A, bool DUMP = true>
// one workgroup per CU (its LDS ring fills the CU): the whole 256-register budget per wave
__global__ __launch_bounds__(512, 1) void wavenet_stream(const Params p, const int NS) {
    using C = SCfg<F16, R, S, A>;
    using P = Prec<F16>;
    using frag = typename P::frag;
    using quad = typename P::quad;
    using elem = typename P::elem;
    constexpr int CH = C::CH, Q = C::Q, FLP = C::FLP;
    constexpr int RT = C::RT, R2T = C::R2T, ST = C::ST, AT = C::AT;
    constexpr int KF_R = C::KF_R, KF_S = C::KF_S, KF_A = C::KF_A;

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const ringLds = lds;                                   // NS * CH KiB
    float* const biasLds = (float*)(lds + (size_t)NS * CH * 1024);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.numLayers;
    const unsigned laneOff = (unsigned)lane * 16u;

    // ---- biases -> LDS; skip biases become running sums (see wn_kernels.hpp) -------------------
    {
        const int nb = L * C::BIAS_L + 2 * A;
        for (int i = tid; i < nb; i += 512) biasLds[i] = p.bias[i];
        __syncthreads();
        for (int s0 = tid; s0 < S; s0 += 512) {
            float run = biasLds[3 * R + s0];
            for (int l = 1; l < L; l++) {
                run += biasLds[l * C::BIAS_L + 3 * R + s0];
                biasLds[l * C::BIAS_L + 3 * R + s0] = run;
            }
        }
        __syncthreads();
    }

    const int nchSample = L * C::NCH_L + C::NCH_H;               // chunks per sample
    const long totalChunks = (long)p.count * nchSample;

    // =============================================================================================
    // LOADER waves
    // =============================================================================================
    if (wv >= 4) {
        const int lw = wv - 4;
        const char* const src0 = (const char*)p.wblob + (size_t)lw * Q * 1024 + laneOff;
        char* const dst0 = ringLds + (size_t)lw * Q * 1024;
        int cs = 0;      // chunk index inside the sample stream of the next chunk to issue
        int slot = 0;    // its ring slot
        auto issue = [&]() {
            const char* src = src0 + (size_t)cs * CH * 1024;
            char* dst = dst0 + (size_t)slot * CH * 1024;
#pragma unroll
            for (int q = 0; q < Q; q++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
            cs = (cs + 1 == nchSample) ? 0 : cs + 1;
            slot = (slot + 1 == NS) ? 0 : slot + 1;
        };
        long issued = 0;
        for (; issued < NS - 2 && issued < totalChunks; issued++) issue();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");                  // barrier #0: chunks 0..NS-3 landed
        for (long c = 0; c + 1 < totalChunks; c++) {             // step c ends with barrier #c+1
            if (issued < totalChunks) {
                issue();                                         // chunk c+NS-2 -> slot of chunk c-2 (consumed)
                issued++;
                // every older chunk (<= c+NS-3, so chunk c+2) has landed; this step's Q loads stay
                // in flight across the barrier
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_barrier" ::: "memory");
        }
        return;
    }

    // =============================================================================================
    // CONSUMER waves: one tile of 16 utterances each
    // =============================================================================================
    const int g = lane >> 4, j = lane & 15;
    const int tile = blockIdx.x * 4 + wv;
    const int b = tile * 16 + j;
    const bool valid = b < p.batch;
    const int bc = valid ? b : p.batch - 1;
    const bool nt = p.ntStream != 0;
    const float* const headBias = biasLds + L * C::BIAS_L;

    const elem* const embPrev = (const elem*)p.embPrev;
    const elem* const embCur = (const elem*)p.embCur;
    const size_t condStride = (size_t)p.tiles * C::COND_FR * 1024;        // one (sample,layer) row
    const char* const condMine = (const char*)p.cond + (size_t)tile * C::COND_FR * 1024;
    char* const ringMine = (char*)p.ring + (size_t)tile * p.ringSlots * KF_R * 1024;

    // LDS ring = circular buffer of NS*CH fragments indexed by the stream position.  `rd` is the ring
    // index of the next fragment to read ahead; ab[] holds the RA fragments ahead of the MFMA.
    constexpr int RA = C::RA;
    const int ringFrags = NS * CH;
    int rd = 0;
    frag ab[RA];
    bool firstChunk = true;   // barrier #0 is taken explicitly before the read-ahead ring is primed
    const char* rdPtr = ringLds + laneOff;
    auto rd_advance = [&](int n) {
        rd += n;
        if (rd >= ringFrags) rd -= ringFrags;
    };
    // next_frag(f): the A fragment at position f of the current body (BODY consumed / BODYP stored
    // fragments; read-ahead skips the padding when it crosses into the next body)
    auto next_frag = [&](auto bodyTag, auto bodypTag, const int f) -> frag {
        constexpr int BODY = decltype(bodyTag)::value, BODYP = decltype(bodypTag)::value;
        if (f % CH == 0) {                                         // consumption enters a new chunk
            if (!firstChunk) asm volatile("s_barrier" ::: "memory");
            firstChunk = false;
        }
        const frag a = ab[f % RA];
        if (f % RA == 0) {
            // the next RA read-ahead fragments are contiguous in the ring (everything is kept a
            // multiple of RA): one address per group, immediates inside it
            if (f + RA == BODY) rd_advance(BODYP - BODY);          // read-ahead leaves this body
            rdPtr = ringLds + (size_t)rd * 1024 + laneOff;
            rd_advance(RA);
        }
        ab[f % RA] = *(const frag*)(rdPtr + (f % RA) * 1024);
        return a;
    };
    // acc[mt] += W(tile mt) * b.  Fragment order inside a GEMM: groups of G tiles, k-fragment-major
    // inside a group, so that the MFMAs accumulating into one tile are G instructions apart.
    auto gemm_s = [&](auto mtTag, auto kfTag, auto bodyTag, auto bodypTag, int pos0, floatx4* acc, const frag* bfr) {
        constexpr int MT = decltype(mtTag)::value, KF = decltype(kfTag)::value;
        constexpr int G = MT >= 4 ? 4 : MT;
#pragma unroll
        for (int mg = 0; mg < MT / G; mg++)
#pragma unroll
            for (int kf = 0; kf < KF; kf++)
#pragma unroll
                for (int mi = 0; mi < G; mi++) {
                    const frag a = next_frag(bodyTag, bodypTag, pos0 + (mg * KF + kf) * G + mi);
                    acc[mg * G + mi] = mma(a, bfr[kf], acc[mg * G + mi]);
                }
    };
    using IC_FL = std::integral_constant<int, C::FL>;
    using IC_FLP = std::integral_constant<int, C::FLP>;
    using IC_FH = std::integral_constant<int, C::FH>;
    using IC_FHP = std::integral_constant<int, C::FHP>;
    using IC_R2T = std::integral_constant<int, R2T>;
    using IC_RT = std::integral_constant<int, RT>;
    using IC_ST = std::integral_constant<int, ST>;
    using IC_AT = std::integral_constant<int, AT>;
    using IC_KFR = std::integral_constant<int, KF_R>;
    using IC_KFS = std::integral_constant<int, KF_S>;
    using IC_KFA = std::integral_constant<int, KF_A>;

    int yPrev = p.yInPrev[bc];
    int yCur = p.yInCur[bc];

    // prefetch of the dilated input + conditioning, one layer ahead, issued at the top of a layer
    frag xpN[KF_R], cdN[C::COND_FR];
    auto prefetch = [&](int tn, int ln, Dil dl) {
        if (ln >= L) { ln -= L; tn += 1; }
        const unsigned sl = (unsigned)(dl.off + (tn & (dl.d - 1)));
        const char* rp = ringMine + (size_t)sl * (KF_R * 1024);
        const char* cp = condMine + ((size_t)tn * L + ln) * condStride;     // padded by one sample
#pragma unroll
        for (int k = 0; k < KF_R; k++) xpN[k] = ld_stream((const frag*)(rp + k * 1024 + laneOff), nt);
#pragma unroll
        for (int k = 0; k < C::COND_FR; k++) cdN[k] = ld_stream((const frag*)(cp + k * 1024 + laneOff), nt);
    };
    prefetch(p.initSample, 0, dil_first());

    // barrier #0 (chunks 0 and 1 have landed), then fill the read-ahead ring; from here on the
    // per-chunk barriers inside gemm_s are #1, #2, ...
    asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < RA; i++) {
        ab[i] = *(const frag*)(ringLds + (size_t)rd * 1024 + laneOff);
        rd_advance(1);
    }
#ifdef WN_TIMING
    unsigned long long tacc[12] = {0};
    unsigned long long tmark = __builtin_amdgcn_s_memtime();
#define WN_SMARK(i)                                                        \
    {                                                                      \
        unsigned long long _n = __builtin_amdgcn_s_memtime();              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 \
        tacc[i] += _n - tmark;                                             \
        tmark = _n;                                                        \
    }
#else
#define WN_SMARK(i)
#endif
    const int tEnd = p.initSample + p.count;
    for (int t = p.initSample; t < tEnd; t++) {
        const bool dumpNow = DUMP && p.dump && (t == tEnd - 1);
        const float selv = p.useRng ? philox_selector(p.rngKey0, p.rngKey1, (unsigned)t, (unsigned)bc)
                                    : p.sel[(size_t)t * p.maxBatch + bc];
        WN_SMARK(11)

        // ---- embedding (nv_wavenet_reference.cpp:42-56) ---------------------------------------
        floatx4 x[RT];
        {
            quad qp[RT], qc[RT];     // all gathers in flight before any of the math
#pragma unroll
            for (int tt = 0; tt < RT; tt++) {
                qp[tt] = *(const quad*)(embPrev + (size_t)yPrev * R + tt * 16 + g * 4);
                qc[tt] = *(const quad*)(embCur + (size_t)yCur * R + tt * 16 + g * 4);
            }
#pragma unroll
            for (int tt = 0; tt < RT; tt++) {
                floatx4 ev = quad_to_f32(qp[tt]) + quad_to_f32(qc[tt]);
                if (p.tanhEmbed) {
#pragma unroll
                    for (int r = 0; r < 4; r++) ev[r] = tanh_t<F16>(ev[r]);
                }
                x[tt] = ev;
            }
        }

        floatx4 skip[ST];
#pragma unroll
        for (int i = 0; i < ST; i++) skip[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        WN_SMARK(0)

        // ---- L dilated layers (nv_wavenet_reference.cpp:58-92) -------------------------------
        frag hb[KF_R];          // h of the previous layer as B fragments (zero before layer 0)
#pragma unroll
        for (int k = 0; k < KF_R; k++)
#pragma unroll
            for (int e = 0; e < P::EPL; e++) hb[k][e] = (elem)0.f;
        Dil dl = dil_first();
        for (int l = 0; l < L; l++) {
            const float* bl = biasLds + l * C::BIAS_L;
            const int d = dl.d;
            const Dil dl1 = dil_next(dl, p.maxDilation, l + 1 == L);
            const bool havePrev = t >= d;

            frag xb[KF_R], xp[KF_R], cd[C::COND_FR];
            to_bfrags<RT>(x, xb);
#pragma unroll
            for (int k = 0; k < KF_R; k++) {
                xp[k] = xpN[k];
                if (!havePrev) {
#pragma unroll
                    for (int e = 0; e < P::EPL; e++) xp[k][e] = (elem)0.f;       // reference :287
                }
            }
#pragma unroll
            for (int k = 0; k < C::COND_FR; k++) cd[k] = cdN[k];
            // x_l[t] replaces x_l[t-d] in the ring (same slot), then request layer l+1's inputs
            {
                char* rp = ringMine + (size_t)(unsigned)(dl.off + (t & (d - 1))) * (KF_R * 1024);
#pragma unroll
                for (int k = 0; k < KF_R; k++) st_stream((frag*)(rp + k * 1024 + laneOff), xb[k], nt);
            }
            prefetch(t, l + 1, dl1);

            // z = Wprev x[t-d] + Wcur x[t] + Bh + Lh
            floatx4 acc[R2T];
#pragma unroll
            for (int i = 0; i < R2T; i++) acc[i] = *(const floatx4*)(bl + i * 16 + g * 4);
            WN_SMARK(1)
            gemm_s(IC_R2T{}, IC_KFR{}, IC_FL{}, IC_FLP{}, C::O_PREV, acc, xp);
            gemm_s(IC_R2T{}, IC_KFR{}, IC_FL{}, IC_FLP{}, C::O_CUR, acc, xb);
            WN_SMARK(2)

            // gate(l)  h = tanh(z_lo + Lh) * sigmoid(z_hi + Lh)  interleaved, value by value, with the
            // skip GEMM of layer l-1 (skip <- Wskip h_{l-1} + skip): the MFMAs run under the VALU work
            floatx4 h[RT];
            {
                constexpr int NSK = C::F_SKIP, NV = RT * 4, GS = ST >= 4 ? 4 : ST;
                int vdone = 0;
#pragma unroll
                for (int sidx = 0; sidx < NSK; sidx++) {
                    const int mg = sidx / (KF_R * GS), kf = (sidx / GS) % KF_R, mi = sidx % GS;
                    const frag a = next_frag(IC_FL{}, IC_FLP{}, C::O_SKIP + sidx);
                    skip[mg * GS + mi] = mma(a, hb[kf], skip[mg * GS + mi]);
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        if (v == vdone && (v + 1) * NSK <= (sidx + 1) * NV) {
                            const int tt = v >> 2, r = v & 3;
                            const float zl = acc[tt][r] + (float)cd[tt / P::TPF][(tt % P::TPF) * 4 + r];
                            const float zh = acc[tt + RT][r] + (float)cd[(tt + RT) / P::TPF][((tt + RT) % P::TPF) * 4 + r];
                            h[tt][r] = gate1<F16>(zl, zh);
                            vdone = v + 1;
                        }
                    }
                    // pin the interleaving: without this the scheduler clusters all MFMAs first
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (dumpNow && valid && l > 0) {
                const float* bp = biasLds + (l - 1) * C::BIAS_L + 3 * R;           // running bias sum
#pragma unroll
                for (int i = 0; i < ST; i++)
                    *(floatx4*)(p.skipOut + ((size_t)(l - 1) * p.maxBatch + b) * S + i * 16 + g * 4) =
                        skip[i] + *(const floatx4*)(bp + i * 16 + g * 4);
            }
            to_bfrags<RT>(h, hb);

            // residual: x <- Wres h + Bres + x
            floatx4 xa[RT];
#pragma unroll
            for (int tt = 0; tt < RT; tt++) xa[tt] = *(const floatx4*)(bl + 2 * R + tt * 16 + g * 4) + x[tt];
            WN_SMARK(3)
            gemm_s(IC_RT{}, IC_KFR{}, IC_FL{}, IC_FLP{}, C::O_RES, xa, hb);
#pragma unroll
            for (int tt = 0; tt < RT; tt++) x[tt] = xa[tt];
            WN_SMARK(4)
            if (dumpNow && valid) {
#pragma unroll
                for (int tt = 0; tt < RT; tt++)
                    *(floatx4*)(p.xtOut + ((size_t)l * p.maxBatch + b) * R + tt * 16 + g * 4) = x[tt];
            }
            WN_SMARK(5)
            dl = dl1;
        }

        WN_SMARK(6)
        // ---- output head (nv_wavenet_reference.cpp:94-104) -----------------------------------
        frag zb[KF_A];
        gemm_s(IC_ST{}, IC_KFR{}, IC_FH{}, IC_FHP{}, C::H_SKIP, skip, hb);   // skip GEMM of the last layer
        {
            frag sb[KF_S];
            {
                const float* bs = biasLds + (L - 1) * C::BIAS_L + 3 * R;
#pragma unroll
                for (int i = 0; i < ST; i++) {
                    skip[i] += *(const floatx4*)(bs + i * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; r++) skip[i][r] = __builtin_fmaxf(skip[i][r], 0.f);
                    if (dumpNow && valid)   // the oracle applies the ReLU to the last layer's skipOut in place
                        *(floatx4*)(p.skipOut + ((size_t)(L - 1) * p.maxBatch + b) * S + i * 16 + g * 4) = skip[i];
                }
                to_bfrags<ST>(skip, sb);
            }
            floatx4 zs[AT];
#pragma unroll
            for (int i = 0; i < AT; i++) zs[i] = *(const floatx4*)(headBias + i * 16 + g * 4);
            gemm_s(IC_AT{}, IC_KFS{}, IC_FH{}, IC_FHP{}, C::H_ZS, zs, sb);
#pragma unroll
            for (int i = 0; i < AT; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) zs[i][r] = __builtin_fmaxf(zs[i][r], 0.f);
            if (dumpNow && valid) {
#pragma unroll
                for (int i = 0; i < AT; i++) *(floatx4*)(p.zs + (size_t)b * A + i * 16 + g * 4) = zs[i];
            }
            to_bfrags<AT>(zs, zb);
        }
        // logits: lane g owns rows g*A/4 .. (g+1)*A/4-1 in register order (row-permuted packing)
        floatx4 za[AT];
#pragma unroll
        for (int i = 0; i < AT; i++) za[i] = *(const floatx4*)(headBias + A + g * (A / 4) + i * 4);
        gemm_s(IC_AT{}, IC_KFA{}, IC_FH{}, IC_FHP{}, C::H_ZA, za, zb);
        if (dumpNow && valid) {
#pragma unroll
            for (int i = 0; i < AT; i++) *(floatx4*)(p.za + (size_t)b * A + g * (A / 4) + i * 4) = za[i];
        }

        WN_SMARK(7)
        // ---- softmax + inverse-CDF pick, 4 lanes per utterance (softmax.cuh:36-191; oracle
        //      matrix.cpp:166-183, nv_wavenet_reference.cpp:106-121) ---------------------------
        float m = za[0][0];
#pragma unroll
        for (int i = 0; i < AT; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) m = __builtin_fmaxf(m, za[i][r]);
        m = __builtin_fmaxf(m, __shfl_xor(m, 16));
        m = __builtin_fmaxf(m, __shfl_xor(m, 32));
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < AT; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float e = fast_exp(za[i][r] - m);
                za[i][r] = e;
                lsum += e;
            }
        const float u = __shfl_xor(lsum, 16);
        const float ps = lsum + u;
        const float v2 = __shfl_xor(ps, 32);
        const float total = ps + v2;
        const float prefix = ((g & 1) ? u : 0.f) + ((g & 2) ? v2 : 0.f);
        const float target = selv * total;
        float cum = prefix;
        int first = C::ZA_REGS;          // first row of this lane with target < cumulative sum
#pragma unroll
        for (int i = 0; i < AT; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                cum += za[i][r];
                first = (first == C::ZA_REGS && target < cum) ? i * 4 + r : first;
            }
        if (dumpNow && valid) {
            const float inv = 1.0f / total;
#pragma unroll
            for (int i = 0; i < AT; i++) *(floatx4*)(p.p + (size_t)b * A + g * (A / 4) + i * 4) = za[i] * inv;
        }
        int y = first < C::ZA_REGS ? g * C::ZA_REGS + first : 0x7fffffff;
        {
            int o = __shfl_xor(y, 16);
            y = o < y ? o : y;
            o = __shfl_xor(y, 32);
            y = o < y ? o : y;
        }
        if (y >= A) y = 128;                       // scan fell off the end (softmax.cuh:154-155)

        if (valid && g == 0) p.yOut[(size_t)b * p.numSamples + t] = y;
        yPrev = yCur;
        yCur = y;
        WN_SMARK(8)
    }
#ifdef WN_TIMING
    if (tid == 0 && blockIdx.x == 0)
        for (int i = 0; i < 12; i++) p.p[i] = (float)tacc[i];
#endif

    if (valid && g == 0) {
        p.yInPrev[b] = yPrev;
        p.yInCur[b] = yCur;
    }
}

// ---- pack kernels for the shared stream -----------------------------------------------------------
// fp32 col-major M x K -> fragments (mt-major, kf-minor); rowperm: lane-contiguous rows (logits);
// gate: a gated 2R x R matrix (rows pre-scaled for gate1())
template <bool F16>
__global__ void pack_weight_stream_kernel(typename Prec<F16>::elem* __restrict__ dst, const float* __restrict__ src,
                                          int M, int K, int rowperm, int gate) {
    constexpr int EPL = Prec<F16>::EPL, TPF = Prec<F16>::TPF;
    const int KF = K / (16 * TPF);
    const size_t n = (size_t)M * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx % EPL;
        const int lane = (idx / EPL) % 64;
        const int f = idx / (EPL * 64);
        const int MT = M / 16;
        const int G = MT >= 4 ? 4 : MT;                       // see gemm_s: groups of G tiles, kf-major
        const int mi = f % G, kf = (f / G) % KF, mt = (f / (G * KF)) * G + mi;
        const int i = lane & 15, g = lane >> 4;
        const int m = rowperm ? ((i >> 2) * (M / 4) + mt * 4 + (i & 3)) : (mt * 16 + i);
        const int k = (kf * TPF + (e >> 2)) * 16 + g * 4 + (e & 3);
        float v = src[(size_t)m + (size_t)k * M];
        if (gate) v *= gate_prescale<F16>(m >= M / 2);          // gated 2R x R matrix: see gate1()
        dst[idx] = (typename Prec<F16>::elem)v;
    }
}

}  // namespace wn
