// nv_wavenet.hpp -- host engine: nvWavenetInfer<T_weight, T_data, R, S, A> for MI355X (gfx950).
//
// Drop-in for the class of the same name in /root/reference/nv_wavenet.cuh:220-640: same template
// parameters, Implementation enum values, constructor, setEmbeddings / setLayerWeights /
// setOutWeights / setInputs, run / run_partial / run_chunks and debug getters, with the same
// argument meaning, layouts (col-major fp32 weights in, host OR device pointers, data copied),
// defaults and error convention (HIP errors print "GPUassert: ..." and exit, precondition
// violations assert, unsupported shapes make run() return false, nv_wavenet_util.cuh:34-40,
// nv_wavenet_singleblock.cuh:273-286).  Streams are hipStream_t.
//
// What is different underneath.  `Implementation` selects between device-code ORGANISATIONS of the
// same MFMA engine (all parity-tested against each other and the oracle):
//   SINGLE_BLOCK            one workgroup runs the whole network for its utterance tile(s), weights
//                           streamed from L2 every sample: wn::wavenet_wg with 1 to 4 tiles of 16
//                           utterances per workgroup by batch size
//   DUAL_BLOCK, PERSISTENT  wn::wavenet_chain: the layer stack split over a chain of CUs, each holding
//                           its layers' weights resident in registers + LDS, plus a head CU; hand-offs
//                           through L2-visible tagged granules (wn_chain.hpp); fewest CUs that hold the model
//   MANYBLOCK_NONPERSISTENT the same chain with one layer per CU
//   AUTO                    chosen from (R, S, A, L, batch, CUs): see pickOrganisation()
// An explicit Organisation (last constructor argument, beyond the reference's signature) overrides.
// batch_size_per_block is validated like the reference (nv_wavenet.cuh:559-561) but the batch tile is
// fixed by the MFMA shape (16 utterances), so it is a no-op hint.  Device buffers are laid out for this
// engine, not for the reference's kernels; the getters return the reference's layouts.
#pragma once

#include <assert.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <type_traits>
#include <vector>

#include "wn_chain.hpp"
#include "wn_kernels.hpp"

#ifndef gpuErrChk
#define gpuErrChk(ans) { wnGpuAssert((ans), __FILE__, __LINE__); }
inline void wnGpuAssert(hipError_t code, const char* file, int line, bool abort = true) {
    if (code != hipSuccess) {
        fprintf(stderr, "GPUassert: %s %s %d\n", hipGetErrorString(code), file, line);
        if (abort) exit(code);
    }
}
#endif

// kernel organisations (beyond the reference: its Implementation enum maps onto these, see above)
enum nvwOrganisation {
    NVW_ORG_AUTO = 0,     // from Implementation and the batch size
    NVW_ORG_WG = 1,       // wn::wavenet_wg, 1 to 4 tiles per workgroup by batch size
    NVW_ORG_WG1 = 2,      // wn::wavenet_wg, one tile per workgroup
    NVW_ORG_WG2 = 3,      // wn::wavenet_wg, two tiles per workgroup
    NVW_ORG_WG3 = 4,      // wn::wavenet_wg, three tiles per workgroup (fp16, R <= 64; else two)
    NVW_ORG_CHAIN = 5,    // wn::wavenet_chain, as many layers per CU as stay resident
    NVW_ORG_CHAIN1 = 6,   // wn::wavenet_chain, one layer per CU
    NVW_ORG_RETIRED7 = 7, // were: wn::wavenet_bcast (every wave its own tile, weights broadcast through an LDS ring; rounds 3-4) and its
    NVW_ORG_RETIRED8 = 8, // variants: measured, never real time anywhere, removed in round 5 (LABNOTES.md) -- refused
    NVW_ORG_RETIRED9 = 9,
    NVW_ORG_WG4 = 10,     // wn::wavenet_wg, four tiles per workgroup (round 6: fp16, R <= 64, dump-free launches with packed conditioning; else three)
    NVW_ORG_LAST = NVW_ORG_WG4
};

template <typename T_weight, typename T_data, int R = 64, int S = 128, int A = 256>
class nvWavenetInfer {
public:
    enum Implementation { AUTO = 0, SINGLE_BLOCK, DUAL_BLOCK, PERSISTENT, MANYBLOCK_NONPERSISTENT };

    static constexpr bool F16 = !std::is_same<T_data, float>::value;
    static_assert(std::is_same<T_data, float>::value || std::is_same<T_data, half>::value,
                  "T_data must be float or half");
    static_assert(std::is_same<T_weight, float>::value == std::is_same<T_data, float>::value,
                  "T_weight/T_data must be <float,float> or <half2,half>");

protected:
    using C = wn::Cfg<F16, R, S, A, 1>;   // stream / layout constants do not depend on BT
    using CC = wn::CCfg<F16, R, S, A>;     // multi-CU chain
    using elem = typename wn::Prec<F16>::elem;
    // in-kernel conditioning (setConditioningWeights + features): the stream that also carries Wcond
    static constexpr int KFC = wn::feat_kfc<F16>();                    // feature fragments per tile and sample
    static constexpr int KC = KFC * 16 * wn::Prec<F16>::TPF;           // channels the features are padded to
    using CF = wn::Cfg<F16, R, S, A, 1, KFC>;

    Implementation m_implementation;
    int m_numLayers, m_maxBatch, m_maxSamples, m_maxDilation, m_tiles, m_numCUs;
    int m_org;           // resolved organisation: NVW_ORG_WG1 / WG2 / WG3 / WG (by batch at run time) / CHAIN / CHAIN1
    bool m_supported;    // false: this shape does not fit the CU (run() returns false, like the reference's unsupported variants)
    int m_chainLpc, m_chainStages;   // layers per chain stage, stages (layer stages + head)
    bool m_tanhEmbed;
    int m_num_samples_per_chunk;
    int m_ringSlots;
    int m_ringDirtyTiles;         // leading tiles whose rings launches have written since they were last zero
    int m_ringLdsMode;            // ring slots of the short dilations in LDS during wavenet_wg launches: 0 as many as fit (default), -1 never
    int m_lastStride;    // row stride of m_yOut in the latest launch (= its num_samples)

    elem* m_wblob;      // packed weight fragments: L layers then the head
    float* m_bias;      // fp32 biases
    elem* m_embedPrev;  // [A][R]
    elem* m_embedCur;
    elem* m_cond;       // packed conditioning (allocated by the first setInputs / setConditioning / packConditioning)
    int m_condRawSamples;
    const void* m_condRaw;    // or: the caller's [N][L][maxBatch][2R] device tensor, consumed in place (setConditioningDirect)
    int m_condRawKind;        // 1: fp32, 2: fp16 (T_data of the fp16 engine)
    const void* m_condUser;   // or: the caller's device buffer ALREADY in the engine's fragment order (setConditioningPacked)
    int m_condUserSamples;    // samples that buffer holds (+ one padding sample)
    // in-kernel conditioning (round 5): Lh = Wcond c + bcond computed by wavenet_wg from the upsampled features
    elem* m_wblobF;           // the weight streams with the conditioning weights in every layer's part (Cfg<.., KFC>)
    float* m_biasF;           // bias table with bcond added to the gate biases
    float* m_condW;           // [L][KC][2R] fp32: the conditioning weight, col-major per layer, channels zero-padded
    float* m_condB;           // [L][2R]
    int2* m_restream;         // fragment map plain stream -> that stream (restream_kernel)
    int m_restreamN;
    int m_nCond;              // channels of the model's features (0: no conditioning weights handed over)
    bool m_featDirty;         // m_wblobF / m_biasF are behind m_wblob / m_bias / m_condW
    elem* m_upTab;            // upsampling (setUpsampling): the ConvTranspose1d weight as MFMA A operands, [stride][5][m * KFC] fragments
    float* m_upBias;          // its bias, [80] fp32
    int m_upWindow, m_upStride;
    elem* m_melFrag;          // the utterances' mel frames in fragment order (setMel), [m_melFrames][m_tiles][KFC] fragments
    int m_melFrames, m_melCap;
    elem* m_feat;             // features packed by the engine (packFeatures), [maxSamples][m_tiles][KFC] fragments
    const void* m_featPtr;    // the features the runs read: m_feat or the caller's buffer in that order (setConditioningFeatures)
    int m_featSamples;
    float* m_outputSelectors;
    elem* m_ring;
    int *m_yInPrev, *m_yInCur, *m_yOut;
    float *m_XtOut, *m_skipOut, *m_Zs, *m_Za, *m_p;
    unsigned long long* m_mail;   // chain mailboxes
    unsigned* m_chainStatus;      // [0] time-out code of the chain launch in flight (0 = fine), [1] code of the latest launch that
                                  // gave up, [2] launches that gave up and were re-run by wavenet_wg (wn::chain_settle_kernel)
    size_t m_mailBytes;
    elem* m_ringShadow;           // the dilation rings as they were before the chain launch in flight ...
    int* m_histShadow;            // ... and the sample history: what the fallback launch starts from
    long long m_chainTimeoutTicks;

    float* m_stage;     // device staging for fp32 uploads from host pointers
    size_t m_stageElems;

    // extensions beyond the reference: in-kernel selectors, int16 PCM output
    bool m_useRng;
    unsigned long long m_rngSeed;
    short* m_pcm;       // [maxBatch][maxSamples] int16, allocated on first setAudioOut
    short* m_mulaw;     // [A] PCM value of every sample index
    short* m_pcmUser;   // caller's buffer (host or device), filled wherever yOut is
    size_t m_pcmUserElems;   // its size in int16 values when the caller said so (0: unknown)
    unsigned long long* m_clk;   // clock probe of the latest wavenet_wg launch (wn::Params::clk), when switched on
    bool m_clkOn;

    // events of run_chunks / run_stream, made on first use and kept
    std::vector<hipEvent_t> m_poolEvents;
    hipEvent_t pooledEvent(size_t i) {
        while (m_poolEvents.size() <= i) {
            hipEvent_t ev;
            gpuErrChk(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            m_poolEvents.push_back(ev);
        }
        return m_poolEvents[i];
    }
    static bool isDevicePtr(const void* ptr) {
        hipPointerAttribute_t attr;
        hipError_t e = hipPointerGetAttributes(&attr, ptr);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
    }
    // ---- staging of fp32 host sources: every array of ONE upload call gets its own place in the
    //      staging buffer, so the pack kernels of that call can all be in flight; the public set* calls
    //      end with one stream synchronisation (the caller may free or reuse its buffers afterwards)
    size_t m_stageUsed;
    void stageBegin(size_t totalElems) {
        if (totalElems > m_stageElems) {
            gpuErrChk(hipStreamSynchronize(0));
            if (m_stage) gpuErrChk(hipFree(m_stage));
            gpuErrChk(hipMalloc(&m_stage, totalElems * sizeof(float)));
            m_stageElems = totalElems;
        }
        m_stageUsed = 0;
    }
    const float* onDevice(const float* src, size_t n) {
        if (isDevicePtr(src)) return src;
        assert(m_stageUsed + n <= m_stageElems);
        float* d = m_stage + m_stageUsed;
        m_stageUsed += (n + 3) & ~(size_t)3;
        gpuErrChk(hipMemcpyAsync(d, src, n * sizeof(float), hipMemcpyHostToDevice, 0));
        return d;
    }
    static int gridFor(size_t n) {
        size_t g = (n + 255) / 256;
        return (int)(g > 4096 ? 4096 : (g ? g : 1));
    }
    // col-major fp32 M x K -> the NW per-wave fragment streams; blockFrag = fragment offset of this
    // matrix inside each wave's stream
    void packWeight(size_t blockFrag, const float* src, int M, int K, int gateRT) {
        const float* d = onDevice(src, (size_t)M * K);
        hipLaunchKernelGGL((wn::pack_weight_kernel<F16>), dim3(gridFor((size_t)M * K)), dim3(256), 0, 0,
                           m_wblob + blockFrag * C::FRAG_ELEMS, d, M, K, C::NW,
                           C::waveStreamFrags(m_numLayers) * C::FRAG_ELEMS, gateRT);
        gpuErrChk(hipGetLastError());
    }
    void convertTo(elem* dst, const float* src, size_t n) {
        const float* d = onDevice(src, n);
        hipLaunchKernelGGL((wn::convert_kernel<F16>), dim3(gridFor(n)), dim3(256), 0, 0, dst, d, n);
        gpuErrChk(hipGetLastError());
    }
    // (dump: the kernel variant that can dump holds every layer's running skip-bias sum in LDS, the dump-free one a single row)
    template <int BT> static size_t ldsNeed(int L, int embTables, bool dump = true) { return wn::Cfg<F16, R, S, A, BT>::ldsBytes(L, embTables, dump); }
    static constexpr size_t kLdsMax = 160 * 1024;
    template <int BT> bool ldsFits(bool dump = true) const { return ldsNeed<BT>(m_numLayers, 0, dump) <= kLdsMax; }
    // how many embedding tables fit in LDS beside everything else: 2, 1 (current tap) or 0
    template <int BT> int embTables(bool dump = true) const {
        return ldsNeed<BT>(m_numLayers, 2, dump) <= kLdsMax ? 2 : ldsNeed<BT>(m_numLayers, 1, dump) <= kLdsMax ? 1 : 0;
    }
    // DUMP = false (no activation dump code at all): every conditioning path of the fp16 engine (the production path); for the fp32
    // engine -- the parity mode, and what the reference's PyTorch entry wavenet_infer() runs -- the packed-conditioning kernels and
    // the chain (round 6: that entry has no getter that could read a dump, pytorch/wavenet_infer.h:33-58); fp32 launches that read
    // the conditioning in place or compute it from features carry the dump code whether asked or not
    // dilation schedule (nv_wavenet.cuh:99,110-111) as a table in the kernel arguments: dilation and first ring slot
    void fillSchedule(wn::Params& p) const {
        int d = 1, off = 0;
        for (int l = 0; l < m_numLayers; l++) {
            p.dil[l].d = d;
            p.dil[l].off = off;
            p.dil[l].lds = 0;
            off += d;
            d <<= 1;
            if (d > m_maxDilation) d = 1;
        }
        p.dil[m_numLayers] = p.dil[0];
        p.dil[m_numLayers + 1] = p.dil[1];
    }
    // The ring slots of the short dilations in LDS (round 6): the largest dilation D (a power of two) whose layers' slots -- sum of d
    // over the layers with d <= D, BT x R/32 KiB each -- fit beside everything else; 0: none.  Sets p.ldsRingD and the LDS slot numbers
    // of the schedule table; returns the bytes to add to the launch's dynamic LDS.
    template <int BT> size_t placeLdsRing(wn::Params& p, size_t need) const {
        using CB = wn::Cfg<F16, R, S, A, BT>;
        int D = 0;
        for (int d = 1; d <= m_maxDilation && d <= WN_LDS_RING_MAXD; d <<= 1)
            if (need + (size_t)CB::ldsRingSlots(m_numLayers, m_maxDilation, d) * CB::RING_SLOT <= kLdsMax) D = d;
        p.ldsRingD = D;
        int slot = 0;
        for (int l = 0; l < m_numLayers; l++) {
            p.dil[l].lds = slot;
            if (p.dil[l].d <= D) slot += p.dil[l].d;
        }
        p.dil[m_numLayers] = p.dil[0];
        p.dil[m_numLayers + 1] = p.dil[1];
        return (size_t)slot * CB::RING_SLOT;
    }
    // ... and whether a launch uses them (the LR instantiations of wavenet_wg: the dump-free kernels).
    // m_ringLdsMode >= 0 (default): as many of the short dilations as fit; -1: never.  Measured at C3 (LABNOTES round 6, us per sample
    // with / without): one tile per workgroup, d <= 4 on chip, 21.1 / 21.7; two tiles, d <= 2 in the place of the older tap's
    // embedding table, 28.4 / 29.2; three tiles, d <= 1, 37.3 / 37.4; four tiles, d <= 1, 43.6 / 44.2.
    static constexpr bool lrBuilt(bool dump, int raw) { return !dump && (raw == 0 || F16); }      // (fp32 engines: dump-free code exists for the packed conditioning only)
    template <int BT> size_t ringPlan(wn::Params& p, size_t need, bool dump, int raw) const {
        p.ldsRingD = 0;
#ifdef WN_EXP_TWO_WG
        if (BT == 1) return 0;
#endif
        if (!lrBuilt(dump, raw) || m_ringLdsMode < 0) return 0;
        wn::Params q = p;
        const size_t bytes = placeLdsRing<BT>(q, need);
        if (q.ldsRingD <= 0) return 0;
        p = q;
        return bytes;
    }
    template <int BT, bool EMB, bool DUMP, int RAW> bool launchK(wn::Params& p, int tiles, int nEmb, hipStream_t stream) {
        using CB = wn::Cfg<F16, R, S, A, BT>;
        const int grid = (tiles + BT - 1) / BT;
        p.embLds = nEmb;
        const size_t need = ldsNeed<BT>(m_numLayers, nEmb, DUMP);
        if constexpr (lrBuilt(DUMP, RAW)) {
            const size_t ringBytes = ringPlan<BT>(p, need, DUMP, RAW);
            if (ringBytes > 0) {
                hipLaunchKernelGGL((wn::wavenet_wg<F16, R, S, A, BT, EMB, DUMP, RAW, true>), dim3(grid), dim3(CB::THREADS), need + ringBytes, stream, p);
                return hipGetLastError() == hipSuccess;
            }
        }
        p.ldsRingD = 0;
        hipLaunchKernelGGL((wn::wavenet_wg<F16, R, S, A, BT, EMB, DUMP, RAW>), dim3(grid), dim3(CB::THREADS), need, stream, p);
        return hipGetLastError() == hipSuccess;
    }
    // embedding tables of a launch in LDS: as many as fit; a launch whose two tables leave no room for a single ring slot gives up the
    // OLDER tap's table for ring slots (its gather is one sample early, off the dependent chain; C3 at two tiles: 28.4 against 29.2 us)
    template <int BT> int planEmb(bool dump, int raw) const {
        int nEmb = embTables<BT>(dump);
#ifdef WN_EXP_TWO_WG
        if (BT == 1) return 0;      // (experiment: two workgroups per CU: 80 KiB of LDS each)
#endif
        if (nEmb == 2 && m_ringLdsMode >= 0 && lrBuilt(dump, raw)) {
            wn::Params q;
            fillSchedule(q);
            if (placeLdsRing<BT>(q, ldsNeed<BT>(m_numLayers, 2, dump)) == 0) nEmb = 1;
        }
        return nEmb;
    }
    template <int BT, bool DUMP, int RAW> bool launchE(wn::Params& p, int tiles, hipStream_t stream) {
        const int nEmb = planEmb<BT>(DUMP, RAW);
        return nEmb ? launchK<BT, true, DUMP, RAW>(p, tiles, nEmb, stream) : launchK<BT, false, DUMP, RAW>(p, tiles, 0, stream);
    }
    template <int BT, bool DUMP> bool launchD(wn::Params& p, int tiles, hipStream_t stream) {
        if (p.condRawKind == 3) return launchE<BT, DUMP, 3>(p, tiles, stream);
        if constexpr (F16) {
            if (p.condRawKind == 2) return launchE<BT, DUMP, 2>(p, tiles, stream);
        }
        return p.condRawKind == 1 ? launchE<BT, DUMP, 1>(p, tiles, stream) : launchE<BT, DUMP, 0>(p, tiles, stream);
    }
    template <int BT> bool launch(wn::Params& p, int tiles, hipStream_t stream) {
        if (!p.dump) {
            if constexpr (F16) return launchD<BT, false>(p, tiles, stream);
            else if (p.condRawKind == 0) return launchE<BT, false, 0>(p, tiles, stream);
        }
        return launchD<BT, true>(p, tiles, stream);
    }
    template <int BT, bool EMB, bool DUMP, int RAW> void allowLdsK() {
        const size_t need = ldsNeed<BT>(m_numLayers, EMB ? embTables<BT>(DUMP) : 0, DUMP);
        if (need <= kLdsMax) {
            gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_wg<F16, R, S, A, BT, EMB, DUMP, RAW>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
            if constexpr (lrBuilt(DUMP, RAW))      // (the whole LDS: what the tables leave free holds ring slots, placeLdsRing)
                gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_wg<F16, R, S, A, BT, EMB, DUMP, RAW, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
        }
    }
    template <int BT, bool DUMP> void allowLdsD() {
        allowLdsK<BT, false, DUMP, 0>();
        allowLdsK<BT, true, DUMP, 0>();
        allowLdsK<BT, false, DUMP, 1>();
        allowLdsK<BT, true, DUMP, 1>();
        allowLdsK<BT, false, DUMP, 3>();
        allowLdsK<BT, true, DUMP, 3>();
        if constexpr (F16) {
            allowLdsK<BT, false, DUMP, 2>();
            allowLdsK<BT, true, DUMP, 2>();
        }
    }
    template <int BT> void allowLds() {
        allowLdsD<BT, true>();
        if constexpr (F16) allowLdsD<BT, false>();
        else {
            allowLdsK<BT, false, false, 0>();
            allowLdsK<BT, true, false, 0>();
        }
    }
    float* headBias() { return m_bias + (size_t)m_numLayers * C::BIAS_L; }

    // ---- organisation ----------------------------------------------------------------------------
    static int chainStagesFor(int L, int lpc) { return (L + lpc - 1) / lpc + 1; }
    // layers per stage: the fewest stages that keep every layer resident, balanced
    static int chainLpcMax(int L) {
        if (!CC::SUPPORTED) return 0;
        const int ns = (L + CC::LPC - 1) / CC::LPC;
        return (L + ns - 1) / ns;
    }
    // a chain of `stages` CUs serves up to TPC_MAX tiles in turn (round 5): all tiles of the batch ride resident chains
    bool chainFits(int lpc, int tiles) const {
        return lpc > 0 && chainStagesFor(m_numLayers, lpc) * ((tiles + CC::TPC_MAX - 1) / CC::TPC_MAX) <= m_numCUs;
    }
    // The single-workgroup organisation: one, two or -- fp16, R <= 64 -- three or four tiles per workgroup (split over the 4 SIMDs
    // of a CU) by batch size, see wgTiles(); beyond four tiles per CU the launch has more workgroups than CUs.
    int singleOrg(int) const { return NVW_ORG_WG; }
    // Per-sample time models (microseconds) of the organisations that can run `tiles` tiles, from the shape: weight bytes per
    // sample W, layers L, CUs.  Every constant is a measurement on MI355X, kept with its source:
    struct OrgTimes {
        // the multi-CU chain only competes at batches of a few tiles, where the GPU holds its full shader clock
        // (profiles/r04_clock_wg_sweep.json: 2.39 GHz up to 192 busy CUs, 2.0 with all 256)
        static constexpr double kClockMHz = 2390.0;
        static constexpr double kStreamBytesPerClk = 58.0;   // weight stream L2 -> registers of one CU (scripts/ubench/stream.hip)
        static constexpr double kStreamLayerUs = 0.25;       // dependent chain per layer beside the stream (one tile)
        static constexpr double kHeadUs = 4.0;               // head GEMMs + softmax + embedding, either organisation
        static constexpr double kChainHopUs = 1.1;           // one stage hand-off (hop 0.55 + entry / exit)
        static constexpr double kChainLayerUs(int r) { return r >= 128 ? 0.55 : 0.4; }   // resident-weight layer of a stage
        // several tiles per chain: a sample of a tile occupies the busiest stage (the head) this long, so a chain's sample period is
        // max(trip round the chain, tiles per chain x this)  (round 5, C4: profiles/r05_chain_c4_tiles_per_chain.json)
        static constexpr double kChainUnitUs = 5.0;
    };
    int pickOrganisation(int tiles) const {
        const int single = singleOrg(tiles);
        const int lpc = chainLpcMax(m_numLayers);
        if (!chainFits(lpc, tiles)) return single;
        const double wBytes = sizeof(elem) * ((double)m_numLayers * (5.0 * R * R + (double)S * R) + (double)A * S + (double)A * A);
        // one workgroup per (1 .. wgMax) tiles, whole rounds of them beyond one per CU
        const int wgMax = WG3 ? 3 : WG2 ? 2 : 1;
        const int rounds = (tiles + wgMax * m_numCUs - 1) / (wgMax * m_numCUs);
        const double tStream = (wBytes / (OrgTimes::kStreamBytesPerClk * OrgTimes::kClockMHz) + OrgTimes::kStreamLayerUs * m_numLayers + OrgTimes::kHeadUs) * rounds;
        const int stages = chainStagesFor(m_numLayers, lpc), perLaunch = m_numCUs / stages;
        const int tpc = (tiles + perLaunch - 1) / perLaunch;
        // several tiles per chain: measured where wavenet_wg cannot be real time (C4, R = 128); the shapes with a three-tile
        // wavenet_wg have their throughput organisation in that (C3: 1.3 utterances per CU and us against the chain's 0.7)
        if (tpc > 1 && WG3) return single;
        double tChain = OrgTimes::kChainHopUs * stages + OrgTimes::kChainLayerUs(R) * m_numLayers + OrgTimes::kHeadUs;
        if (tpc * OrgTimes::kChainUnitUs > tChain) tChain = tpc * OrgTimes::kChainUnitUs;
        // (the time model of wavenet_wg is its one-tile latency: with two or three tiles per workgroup, i.e. beyond one tile per CU, a
        //  sample takes up to twice as long -- the chain is only preferred there when even that is slower)
        const double tWg = tiles > m_numCUs ? tStream * (tiles > 2 * m_numCUs && WG3 ? 1.9 : WG2 ? 1.4 : 1.0) : tStream;
        return tChain < tWg ? NVW_ORG_CHAIN : single;
    }
    void resolveOrganisation(int requested) {
        const int tiles = (m_maxBatch + 15) / 16;
        int org = requested;
        if (org == NVW_ORG_AUTO) {
            switch (m_implementation) {
                case SINGLE_BLOCK: org = singleOrg(tiles); break;
                case DUAL_BLOCK:
                case PERSISTENT: org = chainFits(chainLpcMax(m_numLayers), tiles) ? NVW_ORG_CHAIN : singleOrg(tiles); break;
                case MANYBLOCK_NONPERSISTENT:
                    org = (CC::SUPPORTED && chainFits(1, tiles)) ? NVW_ORG_CHAIN1
                          : chainFits(chainLpcMax(m_numLayers), tiles) ? NVW_ORG_CHAIN : singleOrg(tiles);
                    break;
                default: org = pickOrganisation(tiles); break;
            }
        }
        if (org == NVW_ORG_CHAIN && !chainFits(chainLpcMax(m_numLayers), tiles)) org = singleOrg(tiles);
        if (org == NVW_ORG_CHAIN1 && !(CC::SUPPORTED && chainFits(1, tiles))) org = singleOrg(tiles);
        m_org = org;
        m_chainLpc = org == NVW_ORG_CHAIN ? chainLpcMax(m_numLayers) : org == NVW_ORG_CHAIN1 ? 1 : 0;
        m_chainStages = m_chainLpc ? chainStagesFor(m_numLayers, m_chainLpc) : 0;
    }
    bool isChain() const { return m_chainLpc > 0; }
    // launches whose chains serve several tiles use the instantiation that requests a unit's packed conditioning up front (wn_chain.hpp: HOIST)
    // (built for the shapes whose stages hold at most two layers -- R = 128: C4 --, where it costs no scratch; C2 / C3 stages hold five and spill with it)
    static constexpr bool kChainHoistBuilt = F16 && CC::SUPPORTED && CC::LP <= 2 && A <= 512;      // (the A = 1024 head spills as it is)
    static bool chainHoists(int tpc) { return kChainHoistBuilt && tpc > WN_CHAIN_HOIST_FROM; }
    // tiles per chain for a batch of `tiles` tiles: as few as put every tile on a resident chain, at most TPC_MAX
    int chainTpc(int tiles) const {
        const int perLaunch = m_numCUs / (m_chainStages > 0 ? m_chainStages : 1);
        int tpc = perLaunch > 0 ? (tiles + perLaunch - 1) / perLaunch : 1;
#ifdef WN_CHAIN_TPC_MIN      // (experiment: the mailbox layout of several tiles per chain for launches that have one)
        if (tpc < WN_CHAIN_TPC_MIN) tpc = WN_CHAIN_TPC_MIN;
#endif
        return tpc < 1 ? 1 : tpc > CC::TPC_MAX ? CC::TPC_MAX : tpc;
    }
    // tiles per workgroup of wn::wavenet_wg for a batch of `tiles` tiles
    static constexpr bool WG3 = F16 && R <= 64;   // shapes with a three-tile instantiation
    static constexpr bool WG2 = R < 128;          // ... with a two-tile one (two tiles of R >= 128 need the registers of three: 300-560 spilled,
                                                  // slower than one tile; they fit the LDS only for shallow models anyway)
    bool wg3Fits() const {
        if constexpr (WG3) return ldsFits<3>();
        return false;
    }
    // ... with a four-tile one (round 6): dump-free kernels only -- the folded skip-bias table is what makes room for the fourth tile's
    // exchange images -- and for the packed conditioning; other launches of such an engine take three tiles per workgroup
    static constexpr bool WG4 = WG3;
    bool wg4Fits() const {
        if constexpr (WG4) return ldsFits<4>(false);
        return false;
    }
    int wgTiles(int tiles) const {
#ifdef WN_EXP_TWO_WG
        if (m_org == NVW_ORG_WG && tiles <= 2 * m_numCUs) return 1;      // (experiment: up to two one-tile workgroups per CU)
#endif
        // AUTO beyond three tiles per CU: the launch runs in whole rounds of workgroups, ceil(tiles / (BT x CUs)) of them; a round of
        // four-tile workgroups takes 1.3 times a round of three-tile ones at the socket's power limit (47 against 36 us per sample,
        // LABNOTES round 6), so four tiles win where they save a round: (3, 4] and (6, 8] tiles per CU ...
        bool four = m_org == NVW_ORG_WG4;
        if (m_org == NVW_ORG_WG && tiles > WN_WG4_FROM * m_numCUs) {
            const int r4 = (tiles + 4 * m_numCUs - 1) / (4 * m_numCUs), r3 = (tiles + 3 * m_numCUs - 1) / (3 * m_numCUs);
            four = r4 * 47 <= r3 * 36;
        }
        if (four && wg4Fits() && wg3Fits()) return 4;
        const bool three = four || m_org == NVW_ORG_WG3 || (m_org == NVW_ORG_WG && tiles > 2 * m_numCUs);
        if (three && wg3Fits()) return 3;
        const bool two = three || m_org == NVW_ORG_WG2 || (m_org == NVW_ORG_WG && tiles > m_numCUs);
        if constexpr (WG2) return (two && ldsFits<2>()) ? 2 : 1;
        return 1;
    }

public:
    nvWavenetInfer(int numLayers, int maxDilation, int batchSize, int numSamples, int impl = 0,
                   bool tanhEmbed = true, int organisation = NVW_ORG_AUTO)
        : m_implementation((Implementation)impl), m_numLayers(numLayers), m_maxBatch(batchSize),
          m_maxSamples(numSamples), m_maxDilation(maxDilation), m_tanhEmbed(tanhEmbed),
          m_num_samples_per_chunk(0), m_lastStride(numSamples), m_cond(NULL), m_condRawSamples(0), m_condRaw(NULL), m_condRawKind(0), m_condUser(NULL), m_condUserSamples(0),
          m_wblobF(NULL), m_biasF(NULL), m_condW(NULL), m_condB(NULL), m_restream(NULL), m_restreamN(0), m_nCond(0), m_featDirty(true),
          m_upTab(NULL), m_upBias(NULL), m_upWindow(0), m_upStride(0), m_melFrag(NULL), m_melFrames(0), m_melCap(0),
          m_feat(NULL), m_featPtr(NULL), m_featSamples(0),
          m_mail(NULL), m_chainStatus(NULL), m_mailBytes(0), m_ringShadow(NULL), m_histShadow(NULL),
          m_chainTimeoutTicks(wn::kChainTimeoutTicks), m_stage(NULL), m_stageElems(0), m_useRng(false), m_rngSeed(0), m_pcm(NULL),
          m_mulaw(NULL), m_pcmUser(NULL), m_pcmUserElems(0), m_clk(NULL), m_clkOn(false), m_stageUsed(0) {
        assert(numLayers >= 2 && batchSize > 0 && numSamples > 0 && maxDilation > 0);
        m_ringDirtyTiles = 0;
        m_ringLdsMode = 0;
        assert(numLayers <= wn::kMaxLayers);
        {
            int dev = 0;
            hipDeviceProp_t prop;
            gpuErrChk(hipGetDevice(&dev));
            gpuErrChk(hipGetDeviceProperties(&prop, dev));
            m_numCUs = prop.multiProcessorCount;
        }
        resolveOrganisation(organisation);
        // The bias table of the whole model lives in the LDS of a wavenet_wg workgroup: a model whose
        // table does not fit cannot run there (the reference prints and returns false for shapes a
        // variant does not support, nv_wavenet_singleblock.cuh:273-286)
        m_supported = isChain() || ldsFits<1>();
        if (!m_supported)
            fprintf(stderr, "nvWavenetInfer: R=%d S=%d A=%d with %d layers needs %zu bytes of LDS (> 160 KiB): unsupported\n", R,
                    S, A, numLayers, ldsNeed<1>(numLayers, 0));

        // conditioning / ring are allocated for whole workgroups (two to four tiles per workgroup may be chosen), else
        // exactly the tiles of the batch
        {
            const int tiles = (batchSize + 15) / 16;
            int group = isChain() ? 1 : wgTiles(tiles);
            // (an engine that launches four tiles per workgroup also launches three: smaller batches, launches that dump or read the
            //  conditioning in place)
            if (group >= 3 && wg4Fits()) group = 12;
            m_tiles = (tiles + group - 1) / group * group;
        }

        // dilation schedule (nv_wavenet.cuh:99,110-111): d doubles per layer, back to 1 past maxDilation
        {
            int d = 1, slots = 0;
            for (int l = 0; l < numLayers; l++) {
                slots += d;
                d <<= 1;
                if (d > maxDilation) d = 1;
            }
            m_ringSlots = slots;
        }

        const size_t wElems = (size_t)C::NW * C::waveStreamFrags(numLayers) * C::FRAG_ELEMS;
        gpuErrChk(hipMalloc(&m_wblob, wElems * sizeof(elem)));
        gpuErrChk(hipMemset(m_wblob, 0, wElems * sizeof(elem)));
        const size_t bElems = (size_t)numLayers * C::BIAS_L + 2 * A;
        gpuErrChk(hipMalloc(&m_bias, bElems * sizeof(float)));
        gpuErrChk(hipMemset(m_bias, 0, bElems * sizeof(float)));
        gpuErrChk(hipMalloc(&m_embedPrev, (size_t)A * R * sizeof(elem)));
        gpuErrChk(hipMalloc(&m_embedCur, (size_t)A * R * sizeof(elem)));
        gpuErrChk(hipMemset(m_embedPrev, 0, (size_t)A * R * sizeof(elem)));
        gpuErrChk(hipMemset(m_embedCur, 0, (size_t)A * R * sizeof(elem)));

        // (the packed conditioning -- the largest buffer by far -- is allocated when a caller first packs some: engines
        //  that only ever read the conditioning in place never pay for it)
        gpuErrChk(hipMalloc(&m_outputSelectors, (size_t)numSamples * batchSize * sizeof(float)));
        gpuErrChk(hipMemset(m_outputSelectors, 0, (size_t)numSamples * batchSize * sizeof(float)));

        const size_t ringBytes = ringElems() * sizeof(elem);
        gpuErrChk(hipMalloc(&m_ring, ringBytes));
        gpuErrChk(hipMemset(m_ring, 0, ringBytes));

        gpuErrChk(hipMalloc(&m_yInPrev, batchSize * sizeof(int)));
        gpuErrChk(hipMalloc(&m_yInCur, batchSize * sizeof(int)));
        gpuErrChk(hipMalloc(&m_yOut, (size_t)numSamples * batchSize * sizeof(int)));
        gpuErrChk(hipMemset(m_yOut, 0, (size_t)numSamples * batchSize * sizeof(int)));

        gpuErrChk(hipMalloc(&m_XtOut, (size_t)numLayers * R * batchSize * sizeof(float)));
        gpuErrChk(hipMalloc(&m_skipOut, (size_t)numLayers * S * batchSize * sizeof(float)));
        gpuErrChk(hipMalloc(&m_Zs, (size_t)A * batchSize * sizeof(float)));
        gpuErrChk(hipMalloc(&m_Za, (size_t)A * batchSize * sizeof(float)));
        gpuErrChk(hipMalloc(&m_p, (size_t)A * batchSize * sizeof(float)));
        gpuErrChk(hipMemset(m_XtOut, 0, (size_t)numLayers * R * batchSize * sizeof(float)));
        gpuErrChk(hipMemset(m_skipOut, 0, (size_t)numLayers * S * batchSize * sizeof(float)));
        gpuErrChk(hipMemset(m_Zs, 0, (size_t)A * batchSize * sizeof(float)));
        gpuErrChk(hipMemset(m_Za, 0, (size_t)A * batchSize * sizeof(float)));
        gpuErrChk(hipMemset(m_p, 0, (size_t)A * batchSize * sizeof(float)));

        gpuErrChk(hipMalloc(&m_chainStatus, 4 * sizeof(unsigned)));
        gpuErrChk(hipMemset(m_chainStatus, 0, 4 * sizeof(unsigned)));
        if (isChain()) {
            {
                const int tiles = (batchSize + 15) / 16, perLaunch = m_numCUs / m_chainStages;
                m_mailBytes = CC::mailGranules(tiles < perLaunch ? tiles : perLaunch, m_chainStages, chainTpc(tiles)) * sizeof(unsigned long long);
            }
            gpuErrChk(hipMalloc(&m_mail, m_mailBytes));
            gpuErrChk(hipMemset(m_mail, 0, m_mailBytes));
            gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_chain<F16, R, S, A, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)CC::ldsBytes()));
            gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_chain<F16, R, S, A, false>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)CC::ldsBytes()));
            if constexpr (kChainHoistBuilt)      // (several tiles per chain: the instantiation that requests a unit's conditioning up front)
                gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_chain<F16, R, S, A, false, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)CC::ldsBytes()));
            if (chainHasFallback()) {
                gpuErrChk(hipMalloc(&m_ringShadow, ringBytes));
                gpuErrChk(hipMalloc(&m_histShadow, 2 * (size_t)batchSize * sizeof(int)));
            }
        }

        hipLaunchKernelGGL(wn::silence_kernel, dim3(1), dim3(256), 0, 0, m_yInPrev, m_yInCur, m_maxBatch);
        gpuErrChk(hipGetLastError());

        if (ldsFits<1>()) {            // (a chain engine launches wavenet_wg as its fallback)
            allowLds<1>();
            if (!isChain()) {
                if constexpr (WG2) allowLds<2>();
                if constexpr (WG3) allowLds<3>();
                if constexpr (WG4) {
                    if (wg4Fits()) {
                        allowLdsK<4, false, false, 0>();
                        allowLdsK<4, true, false, 0>();
                    }
                }
            }
        }
        gpuErrChk(hipDeviceSynchronize());
    }

    virtual ~nvWavenetInfer() {
        gpuErrChk(hipDeviceSynchronize());
        for (hipEvent_t ev : m_poolEvents) gpuErrChk(hipEventDestroy(ev));
        gpuErrChk(hipFree(m_wblob));
        gpuErrChk(hipFree(m_bias));
        gpuErrChk(hipFree(m_embedPrev));
        gpuErrChk(hipFree(m_embedCur));
        if (m_cond) gpuErrChk(hipFree(m_cond));
        if (m_wblobF) gpuErrChk(hipFree(m_wblobF));
        if (m_biasF) gpuErrChk(hipFree(m_biasF));
        if (m_condW) gpuErrChk(hipFree(m_condW));
        if (m_condB) gpuErrChk(hipFree(m_condB));
        if (m_restream) gpuErrChk(hipFree(m_restream));
        if (m_feat) gpuErrChk(hipFree(m_feat));
        if (m_upTab) gpuErrChk(hipFree(m_upTab));
        if (m_upBias) gpuErrChk(hipFree(m_upBias));
        if (m_melFrag) gpuErrChk(hipFree(m_melFrag));
        gpuErrChk(hipFree(m_outputSelectors));
        gpuErrChk(hipFree(m_ring));
        gpuErrChk(hipFree(m_yInPrev));
        gpuErrChk(hipFree(m_yInCur));
        gpuErrChk(hipFree(m_yOut));
        gpuErrChk(hipFree(m_XtOut));
        gpuErrChk(hipFree(m_skipOut));
        gpuErrChk(hipFree(m_Zs));
        gpuErrChk(hipFree(m_Za));
        gpuErrChk(hipFree(m_p));
        gpuErrChk(hipFree(m_chainStatus));
        if (m_mail) gpuErrChk(hipFree(m_mail));
        if (m_ringShadow) gpuErrChk(hipFree(m_ringShadow));
        if (m_histShadow) gpuErrChk(hipFree(m_histShadow));
        if (m_stage) gpuErrChk(hipFree(m_stage));
        if (m_pcm) gpuErrChk(hipFree(m_pcm));
        if (m_mulaw) gpuErrChk(hipFree(m_mulaw));
        if (m_clk) gpuErrChk(hipFree(m_clk));
    }

    // false: the shape does not fit this GPU's CUs in the chosen organisation; run() returns false
    bool supported() const { return m_supported; }
    // Multi-CU launches.  A wavenet_chain launch whose workgroups are not all resident in time (other work holds CUs:
    // another stream, engine or process) gives up -- every spin is bounded -- and the SAME stream then re-runs the
    // launch's samples on wavenet_wg from the state the launch started with (launchChain), so the samples delivered are
    // the right ones either way.  chainStatus(): 0, or the code of a give-up that could NOT be repaired (0x100+stage: x,
    // 0x200+stage: skip sums, 0x300: head, 0x400+: placement exchange); chainFallbacks(): launches that were re-run;
    // chainLastTimeout(): the code of the latest of them.  All three synchronise the device.
    unsigned chainStatus() { return statusWord(0); }
    unsigned chainFallbacks() { return statusWord(2); }
    unsigned chainLastTimeout() { return statusWord(1); }
    // Clock probe (measurement aid; wavenet_wg launches): workgroup 0 of every launch that follows records the shader-clock
    // counter and the constant-rate wall-clock counter at its start and end.  lastLaunchClockGHz(): shader ticks per wall
    // second of the latest launch = the clock the chip granted it under its power budget (0 when no launch was probed);
    // synchronises the device.
    void setClockProbe(bool on) {
        if (on && !m_clk) {
            gpuErrChk(hipMalloc(&m_clk, 4 * sizeof(unsigned long long)));
            gpuErrChk(hipMemset(m_clk, 0, 4 * sizeof(unsigned long long)));
        }
        m_clkOn = on;
    }
    double lastLaunchClockGHz() {
        if (!m_clk) return 0.0;
        unsigned long long c[4];
        gpuErrChk(hipDeviceSynchronize());
        gpuErrChk(hipMemcpy(c, m_clk, sizeof(c), hipMemcpyDeviceToHost));
        int dev = 0, wallKHz = 0;
        gpuErrChk(hipGetDevice(&dev));
        gpuErrChk(hipDeviceGetAttribute(&wallKHz, hipDeviceAttributeWallClockRate, dev));
        if (c[3] <= c[1] || wallKHz <= 0) return 0.0;
        return (double)(c[2] - c[0]) / (double)(c[3] - c[1]) * (double)wallKHz * 1e-6;
    }
    // The dilation ring on chip (north_star: "ring buffer staged in LDS with coalesced HBM spill"): wavenet_wg launches keep the slots
    // of the layers with the shortest dilations in the LDS their tables leave free, loaded from / spilled to the HBM ring at the
    // launch's ends.  mode >= 0 (default 0): as many layers as fit -- a model with a short maxDilation keeps its WHOLE ring on chip and
    // touches the HBM ring at the ends of a launch only; C3 (maxDilation 512) keeps d <= 4 / 2 / 1 / 1 at one / two / three / four tiles
    // per workgroup --; -1: never.  Samples are identical either way (tests/test_parity_gpu.py).
    void setRingInLds(int mode) { m_ringLdsMode = mode < 0 ? -1 : 0; }
    // bound of every hand-off spin of the chain (default 1.5 s)
    void setChainTimeoutMs(double ms) { m_chainTimeoutTicks = (long long)(ms * 1e5); }

    // ---- model upload: fp32 in, host or device pointers, data is copied ---------------------
    // embedPrev / embedCur: [A][R]   (nv_wavenet.cuh:396-399)
    virtual void setEmbeddings(float* embedPrev, float* embedCur) {
        stageBegin((size_t)2 * A * R);
        convertTo(m_embedPrev, embedPrev, (size_t)A * R);
        convertTo(m_embedCur, embedCur, (size_t)A * R);
        gpuErrChk(hipStreamSynchronize(0));
    }
    // ---- in-kernel conditioning (beyond the reference class; SURVEY.md 8f rank 1 "feeding Lh directly") -------------------------
    // The model's conditioning convolution (cond_layers of pytorch/wavenet.py:73-74,197): Wcond [L][2R][nCond] (its weight
    // [2R*L][nCond][1] as it is), bcond [L][2R], fp32, host or device, copied.  With them and the upsampled features
    // (setConditioningFeatures / packFeatures) wavenet_wg computes Lh[t][l] = Wcond[l] c[t] + bcond[l] itself: the [N][L][B][2R]
    // tensor of setInputs is never built.  false: more channels than the kernels are built for (kCondChannelsMax).
    bool setConditioningWeights(const float* Wcond, const float* bcond, int nCond) {
        if (nCond <= 0 || nCond > wn::kCondChannelsMax) return false;
        const size_t nW = (size_t)m_numLayers * 2 * R * nCond, nB = (size_t)m_numLayers * 2 * R;
        stageBegin(nW + nB + 8);
        if (!m_condW) {
            gpuErrChk(hipMalloc(&m_condW, (size_t)m_numLayers * KC * 2 * R * sizeof(float)));
            gpuErrChk(hipMalloc(&m_condB, nB * sizeof(float)));
        }
        gpuErrChk(hipMemsetAsync(m_condW, 0, (size_t)m_numLayers * KC * 2 * R * sizeof(float), 0));
        const float* dW = onDevice(Wcond, nW);
        hipLaunchKernelGGL(wn::cond_weight_arrange_kernel, dim3(gridFor(nW)), dim3(256), 0, 0, m_condW, dW, m_numLayers, 2 * R, nCond, KC);
        gpuErrChk(hipGetLastError());
        gpuErrChk(hipMemcpyAsync(m_condB, bcond, nB * sizeof(float), hipMemcpyDefault, 0));
        gpuErrChk(hipStreamSynchronize(0));
        if (nCond != m_nCond) {
            // the upsampling table, its bias and the mel frames were packed for the old channel count: hand them over again
            m_upStride = 0;
            m_upWindow = 0;
            m_melFrames = 0;
        }
        m_nCond = nCond;
        m_featDirty = true;
        return true;
    }
    int conditioningChannels() const { return m_nCond; }
    int featureFragments() const { return KFC; }
    // elements (T_data) of numSamples samples of features in fragment order: numSamples x condTiles() x KFC x 64 x EPL
    size_t featureElems(int numSamples) const { return (size_t)numSamples * m_tiles * KFC * C::FRAG_ELEMS; }
    // Features the caller has produced in fragment order (device memory, T_data; pack_features_kernel documents the order):
    // used in place, kept alive and unchanged until the runs that follow have completed.  Resets the history like setInputs.
    void setConditioningFeatures(const void* frags, int numSamples) {
        assert(numSamples > 0 && numSamples <= m_maxSamples && m_nCond > 0);
        assert(isDevicePtr(frags));
        resetHistory(0);
        gpuErrChk(hipStreamSynchronize(0));
        dropLhConditioning();
        m_featPtr = frags;
        m_featSamples = numSamples;
    }
    // Samples [firstSample, firstSample + count) of the features from a device tensor of `precision`-bit floats addressed as
    // x[b * bStride + c * cStride + (t - firstSample) * tStride] (the model's upsample output [B][nCond][T] has strides (nCond*T, T, 1);
    // channels-last [B][T][nCond] works as well), into the engine's own buffer, asynchronously on `stream`.  History untouched.
    void packFeatures(const void* x, int precision, long long bStride, long long cStride, long long tStride, int firstSample, int count,
                      hipStream_t stream = 0) {
        assert(firstSample >= 0 && count > 0 && firstSample + count <= m_maxSamples && m_nCond > 0);
        assert(precision == 32 || precision == 16);
        assert(isDevicePtr(x));
        if (!m_feat) {
            gpuErrChk(hipMalloc(&m_feat, featureElems(m_maxSamples) * sizeof(elem)));
            gpuErrChk(hipMemsetAsync(m_feat, 0, featureElems(m_maxSamples) * sizeof(elem), stream));
            gpuErrChk(hipStreamSynchronize(stream));
        }
        dropLhConditioning();
        m_featPtr = m_feat;
        m_featSamples = m_maxSamples;      // (run_partial of a chunk only reads what has been packed: the caller's ordering, as with packConditioning)
        const int tilesUsed = (m_maxBatch + 15) / 16;
        const size_t nblk = (size_t)tilesUsed * ((count + 7) / 8);
        hipLaunchKernelGGL((wn::pack_features_kernel<F16>), dim3((unsigned)(nblk > 65536 ? 65536 : nblk)), dim3(256), 0, stream,
                           m_feat + featureElems(firstSample), x, precision, bStride, cStride, tStride, m_nCond, m_maxBatch, count, m_tiles,
                           tilesUsed);
        gpuErrChk(hipGetLastError());
    }
    // the whole utterance at once (resets the history like setInputs; synchronises)
    void setFeatures(const void* x, int precision, long long bStride, long long cStride, long long tStride, int numSamples) {
        resetHistory(0);
        packFeatures(x, precision, bStride, cStride, tStride, 0, numSamples, 0);
        gpuErrChk(hipStreamSynchronize(0));
    }
    bool conditioningFromFeatures() const { return m_featPtr != NULL; }
    // ---- ... and the upsampling in front of it (the other half of WaveNet.get_cond_input, pytorch/wavenet.py:195-197) -----------
    // The model's `upsample` ConvTranspose1d: upW [nCond][nCond][window], upB [nCond], fp32, host or device, copied; window must be a
    // multiple of stride (the reference's 800 / 200), at most 5 strides.  Needs setConditioningWeights first (the channel count).
    static constexpr int kUpMaxTaps = 5;      // (the operands of a workgroup's phases sit in LDS: phases x 5 x taps x KFC KiB <= 160 KiB)
    bool setUpsampling(const float* upW, const float* upB, int window, int stride) {
        if (m_nCond <= 0 || stride <= 0 || window < stride || window % stride != 0 || window / stride > kUpMaxTaps) return false;
        const size_t nW = (size_t)m_nCond * m_nCond * window;
        stageBegin(nW + 128);
        if (m_upTab) gpuErrChk(hipFree(m_upTab));
        const size_t tabElems = (size_t)stride * wn::kUpRowTiles * (window / stride) * KFC * C::FRAG_ELEMS;
        gpuErrChk(hipMalloc(&m_upTab, tabElems * sizeof(elem)));
        if (!m_upBias) gpuErrChk(hipMalloc(&m_upBias, wn::kUpRowTiles * 16 * sizeof(float)));
        gpuErrChk(hipMemsetAsync(m_upBias, 0, wn::kUpRowTiles * 16 * sizeof(float), 0));
        const float* dW = onDevice(upW, nW);
        hipLaunchKernelGGL((wn::pack_upsample_kernel<F16>), dim3(gridFor(tabElems)), dim3(256), 0, 0, m_upTab, dW, m_nCond, window, stride);
        gpuErrChk(hipGetLastError());
        gpuErrChk(hipMemcpyAsync(m_upBias, upB, m_nCond * sizeof(float), hipMemcpyDefault, 0));
        // (per engine, hence per device: the attribute belongs to the device's copy of the kernel)
        gpuErrChk(hipFuncSetAttribute((const void*)wn::upsample_features_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        gpuErrChk(hipStreamSynchronize(0));
        if (stride != m_upStride) m_melFrames = 0;      // frames handed over for another stride were checked against that stride
        m_upWindow = window;
        m_upStride = stride;
        return true;
    }
    int upsamplingStride() const { return m_upStride; }
    int melSamples() const { return m_melFrames * m_upStride; }      // samples the frames handed over with setMel upsample to (0: none)
    bool hasFeatureBuffer() const { return m_feat != NULL; }         // the engine's own feature buffer exists (packFeatures / upsampleFeatures)
    int maxBatch() const { return m_maxBatch; }
    // debug getter: samples [firstSample, firstSample + count) of the engine's own feature buffer (fragment order, T_data) -> dst
    void getFeatures(void* dst, int firstSample, int count) {
        assert(m_feat != NULL && firstSample >= 0 && count > 0 && firstSample + count <= m_maxSamples);
        gpuErrChk(hipDeviceSynchronize());
        gpuErrChk(hipMemcpy(dst, m_feat + featureElems(firstSample), featureElems(count) * sizeof(elem), hipMemcpyDefault));
    }
    // The utterances' features before upsampling ("mel frames"): device tensor of `precision`-bit floats addressed
    // x[b * bStride + c * cStride + f * fStride], frames <= maxSamples / stride; copied (into fragment order).  Resets the history like
    // setInputs: the start of an utterance batch.
    void setMel(const void* mel, int precision, long long bStride, long long cStride, long long fStride, int frames) {
        assert(m_upStride > 0 && frames > 0 && (long long)frames * m_upStride <= m_maxSamples);
        assert(isDevicePtr(mel) && (precision == 32 || precision == 16));
        if (frames > m_melCap) {
            if (m_melFrag) gpuErrChk(hipFree(m_melFrag));
            m_melCap = m_maxSamples / m_upStride;
            gpuErrChk(hipMalloc(&m_melFrag, featureElems(m_melCap) * sizeof(elem)));
        }
        resetHistory(0);
        const int tilesUsed = (m_maxBatch + 15) / 16;
        gpuErrChk(hipMemsetAsync(m_melFrag, 0, featureElems(frames) * sizeof(elem), 0));      // (tiles beyond the batch: zero frames)
        const size_t nblk = (size_t)tilesUsed * ((frames + 7) / 8);
        hipLaunchKernelGGL((wn::pack_features_kernel<F16>), dim3((unsigned)(nblk > 65536 ? 65536 : nblk)), dim3(256), 0, 0, m_melFrag, mel, precision,
                           bStride, cStride, fStride, m_nCond, m_maxBatch, frames, m_tiles, tilesUsed);
        gpuErrChk(hipGetLastError());
        gpuErrChk(hipStreamSynchronize(0));
        m_melFrames = frames;
    }
    // Samples [firstSample, firstSample + count) of the upsampled features from the frames handed over with setMel, into the engine's
    // feature buffer, asynchronously on `stream` (one MFMA kernel: upsample_features_kernel); the runs that follow read them.
    void upsampleFeatures(int firstSample, int count, hipStream_t stream = 0) {
        assert(m_upStride > 0 && m_melFrames > 0);
        assert(firstSample >= 0 && count > 0 && firstSample + count <= m_melFrames * m_upStride);
        if (!m_feat) {
            gpuErrChk(hipMalloc(&m_feat, featureElems(m_maxSamples) * sizeof(elem)));
            gpuErrChk(hipMemsetAsync(m_feat, 0, featureElems(m_maxSamples) * sizeof(elem), stream));
            gpuErrChk(hipStreamSynchronize(stream));
        }
        dropLhConditioning();
        m_featPtr = m_feat;
        m_featSamples = m_melFrames * m_upStride;
        const int m = m_upWindow / m_upStride, tilesUsed = (m_maxBatch + 15) / 16;
        const size_t lds = (size_t)wn::up_phases<F16>() * wn::kUpRowTiles * m * KFC * 1024;      // (a pair of phases in fp16)
        // a phase per workgroup; phases with few columns (long strides, short chunks) get more workgroups per phase
        const int pairs = (m_upStride + wn::up_phases<F16>() - 1) / wn::up_phases<F16>();
        const int gx = pairs < 1024 ? pairs : 1024;
        const long long cols = (long long)((count + m_upStride - 1) / m_upStride + 1) * tilesUsed;
        int gy = (int)((cols + 255) / 256);         // (a wave takes groups of two or four columns)
        const int gyMax = (1024 + gx - 1) / gx;
        if (gy > gyMax) gy = gyMax;
        if (gy < 1) gy = 1;
        hipLaunchKernelGGL((wn::upsample_features_kernel<F16>), dim3(gx, gy), dim3(64 * wn::up_waves<F16>()), lds, stream, m_feat, m_melFrag, m_upTab, m_upBias, m, m_upStride,
                           m_tiles, tilesUsed, firstSample, count);
        gpuErrChk(hipGetLastError());
    }
    // Features in, samples out (role of pytorch/inference.py:40-62 around run_chunks, nv_wavenet.cuh:445-497): the whole utterance
    // batch chunk by chunk -- upsampling of a chunk's features, its generation launch, the copy of its samples (and PCM) on a second
    // stream, consume(yOut, first, count) on the calling thread -- from the frames handed over with setMel.  Selectors: the seed or
    // table in force.  num_samples <= frames * stride.
    template <class Callback>
    bool run_stream(int num_samples_per_chunk, Callback consume, int num_samples, int batch_size, int* yOut = NULL, hipStream_t stream = 0) {
        assert(num_samples_per_chunk > 0 && m_melFrames > 0 && num_samples <= m_melFrames * m_upStride);
        struct Piece {
            int first, count;
            hipEvent_t generated, delivered;
        };
        std::vector<Piece> pieces;
        for (int first = 0; first < num_samples; first += num_samples_per_chunk) {
            Piece pc;
            pc.first = first;
            pc.count = num_samples - first < num_samples_per_chunk ? num_samples - first : num_samples_per_chunk;
            pc.generated = pooledEvent(2 * pieces.size());
            pc.delivered = pooledEvent(2 * pieces.size() + 1);
            pieces.push_back(pc);
        }
        // (the events are the engine's, made once.  A second stream exists only where it buys something: when the samples or the PCM
        //  go to HOST memory, so that the copy of a chunk overlaps the generation of the next; device-resident outputs are copied in
        //  stream order, 20 us per chunk.  Made per call and destroyed at its end -- 1.4 ms -- because streams kept alive in the engine
        //  take hardware queues away from the caller's own streams: with them pooled, a launch on a caller's side stream queued up
        //  behind a kernel of another of its streams, tests/..::test_chain_launch_that_cannot_become_resident..)
        const bool hostOut = (yOut != NULL && !isDevicePtr(yOut)) || (m_pcmUser != NULL && !isDevicePtr(m_pcmUser));
        hipStream_t genStream = stream, outStream = stream;
        const bool ownGen = hostOut && !stream;
        if (ownGen) gpuErrChk(hipStreamCreate(&genStream));
        if (hostOut) gpuErrChk(hipStreamCreate(&outStream));
        bool ok = true;
        for (size_t k = 0; k < pieces.size(); k++) {
            const Piece& pc = pieces[k];
            // (the generation kernel holds every CU for the length of its launch: the upsampling of a chunk runs in front of it on the
            //  same stream -- 1 to 2 % of the chunk's time -- instead of beside it)
            upsampleFeatures(pc.first, pc.count, genStream);
            m_num_samples_per_chunk = pc.count;
            ok = run_partial(pc.first, num_samples, batch_size, NULL, 1, false, genStream) && ok;
            gpuErrChk(hipEventRecord(pc.generated, genStream));
            if (hostOut) gpuErrChk(hipStreamWaitEvent(outStream, pc.generated, 0));
            if (yOut) getYOut(yOut, pc.first, pc.count, outStream);
            if (m_pcmUser) getAudioOut(m_pcmUser, pc.first, pc.count, outStream);
            gpuErrChk(hipEventRecord(pc.delivered, outStream));
        }
        m_num_samples_per_chunk = 0;
        for (size_t k = 0; k < pieces.size(); k++) {
            gpuErrChk(hipEventSynchronize(pieces[k].delivered));
            consume(yOut, pieces[k].first, pieces[k].count);
        }
        if (ownGen) gpuErrChk(hipStreamDestroy(genStream));
        if (hostOut) gpuErrChk(hipStreamDestroy(outStream));
        return ok;
    }
    // col-major Wprev,Wcur 2RxR; Bh 2R; Wres RxR; Bres R; Wskip SxR; Bskip S (nv_wavenet.cuh:400-409)
    virtual void setLayerWeights(int layer, float* Wprev, float* Wcur, float* Bh, float* Wres, float* Bres,
                                 float* Wskip, float* Bskip) {
        assert(layer >= 0 && layer < m_numLayers);
        stageBegin((size_t)5 * R * R + (size_t)S * R + 3 * R + S + 32);
        float* b = m_bias + (size_t)layer * C::BIAS_L;
        // one launch packs the four matrices and the three bias vectors of the layer
        wn::LayerSrc src;
        src.Wprev = onDevice(Wprev, (size_t)2 * R * R);
        src.Wcur = onDevice(Wcur, (size_t)2 * R * R);
        src.Bh = onDevice(Bh, 2 * R);
        src.Wres = onDevice(Wres, (size_t)R * R);
        src.Bres = onDevice(Bres, R);
        src.Wskip = onDevice(Wskip, (size_t)S * R);
        src.Bskip = onDevice(Bskip, S);
        hipLaunchKernelGGL((wn::pack_layer_kernel<F16>), dim3(gridFor((size_t)5 * R * R + (size_t)S * R)), dim3(256), 0, 0,
                           m_wblob, b, src, R, S, C::NW, C::waveStreamFrags(m_numLayers) * C::FRAG_ELEMS,
                           (int)C::streamPos(layer, C::O_PREV, m_numLayers), (int)C::streamPos(layer, C::O_CUR, m_numLayers),
                           (int)C::streamPos(layer, C::O_RES, m_numLayers), (int)C::streamPos(layer, C::O_SKIP, m_numLayers));
        gpuErrChk(hipGetLastError());
        gpuErrChk(hipStreamSynchronize(0));
        m_featDirty = true;
    }
    // col-major Wzs AxS, Bzs A, Wza AxA, Bza A (nv_wavenet.cuh:410-415)
    virtual void setOutWeights(float* Wzs, float* Bzs, float* Wza, float* Bza) {
        stageBegin((size_t)A * S + (size_t)A * A + 16);
        const size_t hf = C::headOffsetFrags(m_numLayers);
        packWeight(hf + C::O_ZS, Wzs, A, S, 0);
        packWeight(hf + C::O_ZA, Wza, A, A, 0);
        gpuErrChk(hipMemcpyAsync(headBias(), Bzs, A * sizeof(float), hipMemcpyDefault, 0));
        gpuErrChk(hipMemcpyAsync(headBias() + A, Bza, A * sizeof(float), hipMemcpyDefault, 0));
        gpuErrChk(hipStreamSynchronize(0));
        m_featDirty = true;
    }

    // Lh: [maxSamples][L][maxBatch][2R] conditioning, outputSelectors: [maxSamples][maxBatch]
    // uniform draws; resets the sample history to 128 (nv_wavenet.cuh:417-422).
    void setInputs(float* Lh, float* outputSelectors) { setInputs(Lh, outputSelectors, m_maxSamples); }
    // Same with numSamples <= maxSamples rows of Lh / outputSelectors (an utterance shorter than the
    // engine's capacity: both layouts are sample-major, so a prefix is a valid input)
    void setInputs(float* Lh, float* outputSelectors, int numSamples) {
        setConditioning(Lh, numSamples);
        m_useRng = false;
        gpuErrChk(hipMemcpy(m_outputSelectors, outputSelectors, (size_t)numSamples * m_maxBatch * sizeof(float),
                            hipMemcpyDefault));
    }

    // ---- extensions beyond the reference --------------------------------
    // The conditioning half of setInputs (also resets the history to 128); pair it with
    // setSelectorSeed() and no [N][B] selector matrix is ever built or uploaded.
    void setConditioning(float* Lh) { setConditioning(Lh, m_maxSamples); }
    void setConditioning(float* Lh, int numSamples, hipStream_t stream = 0) {
        assert(numSamples > 0 && numSamples <= m_maxSamples);
        resetHistory(stream);
        packConditioning(Lh, 0, numSamples, stream);
        gpuErrChk(hipStreamSynchronize(stream));
    }
    // Packs samples [firstSample, firstSample + count) of the conditioning (Lh points at sample firstSample)
    // into the engine's fragment order, asynchronously on `stream` when Lh is device memory: lets a caller
    // stream the conditioning chunk by chunk behind run_partial() of the previous chunk.  From here on the engine
    // reads the packed copy (a tensor handed over with setConditioningDirect is let go).
    void packConditioning(float* Lh, int firstSample, int count, hipStream_t stream = 0) {
        assert(firstSample >= 0 && count > 0 && firstSample + count <= m_maxSamples);
        m_condRaw = NULL;
        m_condRawKind = 0;
        m_condUser = NULL;
        m_featPtr = NULL;
        const size_t rows = (size_t)count * m_numLayers;
        const size_t srcPerRow = (size_t)m_maxBatch * 2 * R;
        const size_t dstPerRow = (size_t)m_tiles * 16 * 2 * R;
        if (!m_cond) {
            const size_t condElems = (size_t)(m_maxSamples + 1) * m_numLayers * dstPerRow;   // + one padding sample
            gpuErrChk(hipMalloc(&m_cond, condElems * sizeof(elem)));
            // zeroed before ANY stream may pack into it (a later chunk packed on another stream must not be overtaken by this)
            gpuErrChk(hipMemsetAsync(m_cond, 0, condElems * sizeof(elem), stream));
            gpuErrChk(hipStreamSynchronize(stream));
        }
        elem* const dst0 = m_cond + (size_t)firstSample * m_numLayers * dstPerRow;
        const bool dev = isDevicePtr(Lh);
        // host sources go through the staging buffer in chunks of <= 64 Mi floats
        size_t chunkRows = dev ? rows : ((size_t)64 << 20) / srcPerRow;
        if (chunkRows < 1) chunkRows = 1;
        if (!dev) stageBegin((chunkRows < rows ? chunkRows : rows) * srcPerRow);
        for (size_t r0 = 0; r0 < rows; r0 += chunkRows) {
            const size_t nr = (rows - r0 < chunkRows) ? rows - r0 : chunkRows;
            const float* src = Lh + r0 * srcPerRow;
            if (!dev) {
                gpuErrChk(hipStreamSynchronize(stream));   // the previous chunk's kernel has read the staging buffer
                gpuErrChk(hipMemcpy(m_stage, src, nr * srcPerRow * sizeof(float), hipMemcpyHostToDevice));
                src = m_stage;
            }
            // one workgroup per (row, tile): 16 utterances x 2R channels, read and written coalesced
            const size_t nblk = nr * (size_t)m_tiles;
            const int grid = (int)(nblk > 65536 ? 65536 : nblk);
            hipLaunchKernelGGL((wn::pack_cond_tiled_kernel<F16, R>), dim3(grid), dim3(256), 0, stream, dst0 + r0 * dstPerRow, src, nr,
                               m_maxBatch, m_tiles);
            gpuErrChk(hipGetLastError());
        }
    }
    // Device-resident conditioning WITHOUT the copy (the reference's own recommendation, README.md:44; SURVEY.md 8f
    // rank 1): Lh is the caller's [numSamples][L][maxBatch][2R] tensor in device memory, fp32 (precision = 32) or -- fp16
    // engine -- T_data = fp16 (precision = 16: half the bytes; the reference keeps m_Lh in T_data, nv_wavenet.cuh:326); the
    // kernels read it in place (16 / 8 bytes per lane and gate tile) and nothing is packed.  The caller keeps it alive and
    // unchanged until the run calls that follow have completed.  Resets the sample history like setInputs.  Host pointers
    // (fp32 only) fall back to setConditioning.
    void setConditioningDirect(const void* Lh, int numSamples, int precision = 32) {
        assert(numSamples > 0 && numSamples <= m_maxSamples);
        assert(precision == 32 || (precision == 16 && F16));
        if (!isDevicePtr(Lh)) {
            assert(precision == 32);
            setConditioning((float*)Lh, numSamples);
            return;
        }
        resetHistory(0);
        gpuErrChk(hipStreamSynchronize(0));
        m_condRaw = Lh;
        m_condRawKind = precision == 16 ? 2 : 1;
        m_condRawSamples = numSamples;
        m_condUser = NULL;
        m_featPtr = NULL;
    }
    // Conditioning that the caller has PRODUCED in the engine's own fragment order (round 3): T_data
    // [numSamples + 1][L][condTiles()][wave][fragment][lane][EPL], gate rows pre-scaled -- exactly what packConditioning writes
    // (pack_cond_tiled_kernel documents the order; nv_wavenet_amd/nv_wavenet.py: cond_fragment_order / get_cond_input(layout=
    // "packed") produce it from the model's conditioning convolution by permuting and scaling that convolution's output
    // channels, i.e. for free).  The generation kernels then run their packed path on the caller's buffer: no copy, no second
    // pass, no in-place conversions -- the headline kernel as it is.  The buffer holds one padding sample past the last (the
    // prefetch reads ahead), stays alive and unchanged until the runs that follow have completed.  Resets the history.
    void setConditioningPacked(const void* frags, int numSamples) {
        assert(numSamples > 0 && numSamples <= m_maxSamples);
        assert(isDevicePtr(frags));
        resetHistory(0);
        gpuErrChk(hipStreamSynchronize(0));
        m_condRaw = NULL;
        m_condRawKind = 0;
        m_featPtr = NULL;
        m_condUser = frags;
        m_condUserSamples = numSamples;      // (run_partial refuses to generate past what the caller handed over)
    }
    // elements (T_data) a packed buffer of numSamples samples must hold: (numSamples + 1) x L x condTiles() x 16 x 2R
    size_t condPackedElems(int numSamples) const { return (size_t)(numSamples + 1) * m_numLayers * m_tiles * 16 * 2 * R; }
    // tiles of 16 utterances per (sample, layer) row of the packed conditioning (the batch rounded up to whole workgroups)
    int condTiles() const { return m_tiles; }
    bool conditioningInPlace() const { return m_condRaw != NULL; }
    // the selector half of setInputs: [numSamples][maxBatch] uniform draws (host or device), conditioning and history untouched
    void setSelectors(float* outputSelectors, int numSamples) {
        assert(numSamples > 0 && numSamples <= m_maxSamples);
        m_useRng = false;
        gpuErrChk(hipMemcpy(m_outputSelectors, outputSelectors, (size_t)numSamples * m_maxBatch * sizeof(float), hipMemcpyDefault));
    }
    // Selectors are drawn inside the kernel: Philox4x32-10, counter {sample, utterance, 0, 0}, key =
    // seed (replaces the rand() table of pytorch/wavenet_infer.cu:92-94).  A later setInputs()
    // returns to the uploaded table.
    void setSelectorSeed(unsigned long long seed) {
        m_useRng = true;
        m_rngSeed = seed;
    }
    // int16 PCM beside the indices: pcm[b][t] = int16(32768 * mu_law_decode(y[b][t], A))
    // (pytorch/utils.py:62-70, inference.py:58-60).  pcmOut: caller-owned [batch][num_samples] int16 of the run calls
    // that follow, host or device; filled by run / run_partial / run_chunks wherever yOut is; NULL disables.
    // pcmElems: the buffer's size in int16 values (0 = not stated); a stated size is checked by every run call.
    void setAudioOut(short* pcmOut, size_t pcmElems = 0) {
        m_pcmUser = pcmOut;
        m_pcmUserElems = pcmOut ? pcmElems : 0;
        if (pcmOut && !m_pcm) {
            gpuErrChk(hipMalloc(&m_pcm, (size_t)m_maxSamples * m_maxBatch * sizeof(short)));
            gpuErrChk(hipMemset(m_pcm, 0, (size_t)m_maxSamples * m_maxBatch * sizeof(short)));
            std::vector<short> table(A);
            const double mu = (double)A - 1.0;
            for (int y = 0; y < A; y++) {
                const double signal = 2.0 * ((double)y / mu) - 1.0;
                const double magnitude = (1.0 / mu) * (std::pow(1.0 + mu, std::fabs(signal)) - 1.0);
                const double v = 32768.0 * (signal > 0 ? magnitude : (signal < 0 ? -magnitude : 0.0));
                // Truncation like numpy's astype('int16') in the reference's inference.py:58-60, INCLUDING its
                // wrap of the top bin (y = A-1 -> +32768 -> -32768): kept on purpose, this output is pinned
                // bit for bit to the reference's own utils.py / inference.py (tests/golden/mulaw_pcm.npz)
                table[y] = (short)(int)v;
            }
            gpuErrChk(hipMalloc(&m_mulaw, A * sizeof(short)));
            gpuErrChk(hipMemcpy(m_mulaw, table.data(), A * sizeof(short), hipMemcpyHostToDevice));
        }
    }
    void getAudioOut(short* pcm, int offset, int size, hipStream_t stream = 0) {
        gpuErrChk(hipMemcpy2DAsync(pcm + offset, m_lastStride * sizeof(short), m_pcm + offset,
                                   m_lastStride * sizeof(short), size * sizeof(short), m_maxBatch, hipMemcpyDefault,
                                   stream));
    }

    // Which device code run(num_samples, batch_size, ..., dumpActivations) launches: kernel name with its
    // template arguments, tiles per workgroup, workgroups, dynamic LDS bytes (for benchmarks / logs).
    void kernelInfo(int batch_size, bool dumpActivations, char* buf, int n) const {
        const int tiles = (batch_size + 15) / 16;
        const bool chainLaunch = isChain() && !m_featPtr;
        const bool dump = dumpActivations || (!F16 && !chainLaunch && (m_featPtr || m_condRaw));      // (see launch())
        if (chainLaunch) {
            const int perLaunch = m_numCUs / m_chainStages, chains = tiles < perLaunch ? tiles : perLaunch;
            const int tpcNow = chainTpc((m_maxBatch + 15) / 16);
            snprintf(buf, n, "wn::wavenet_chain<%s,%d,%d,%d,DUMP=%d%s> stages=%d layers/stage=%d chains=%d tiles/chain=%d wgs=%d lds=%zu",
                     F16 ? "fp16" : "fp32", R, S, A, dump ? 1 : 0, (!dump && chainHoists(tpcNow)) ? ",HOIST=1" : "", m_chainStages, m_chainLpc, chains, tpcNow,
                     m_chainStages * chains, CC::ldsBytes());
            return;
        }
        const int raw = m_featPtr ? 3 : m_condRaw ? m_condRawKind : 0;
        const int bt = launchTiles(tiles, dump, raw);
        int nEmb = planEmb<1>(dump, raw);
        size_t lds = ldsNeed<1>(m_numLayers, nEmb, dump);
        if constexpr (WG2) {
            if (bt == 2) {
                nEmb = planEmb<2>(dump, raw);
                lds = ldsNeed<2>(m_numLayers, nEmb, dump);
            }
        }
        if constexpr (WG3) {
            if (bt == 3) {
                nEmb = planEmb<3>(dump, raw);
                lds = ldsNeed<3>(m_numLayers, nEmb, dump);
            }
            if (bt == 4) {
                nEmb = planEmb<4>(dump, raw);
                lds = ldsNeed<4>(m_numLayers, nEmb, dump);
            }
        }
        // ring slots in LDS (ringPlan): the largest dilation held there, 0 = the launch keeps the whole ring in HBM
        int ringD = 0;
        {
            wn::Params q;
            fillSchedule(q);
            size_t rb = 0;
            if (bt == 1) rb = ringPlan<1>(q, lds, dump, raw);
            if constexpr (WG2) {
                if (bt == 2) rb = ringPlan<2>(q, lds, dump, raw);
            }
            if constexpr (WG3) {
                if (bt == 3) rb = ringPlan<3>(q, lds, dump, raw);
                if (bt == 4) rb = ringPlan<4>(q, lds, dump, raw);
            }
            lds += rb;
            ringD = q.ldsRingD;
        }
        char ring[48] = "";
        if (ringD > 0) snprintf(ring, sizeof(ring), " ring_in_lds=d<=%d", ringD);
        snprintf(buf, n, "wn::wavenet_wg<%s,%d,%d,%d,BT=%d,EMBLDS=%d,DUMP=%d,RAW=%d%s> tiles/wg=%d wgs=%d lds=%zu%s",
                 F16 ? "fp16" : "fp32", R, S, A, bt, nEmb, dump ? 1 : 0, raw, ringD > 0 ? ",LR=1" : "", bt, (tiles + bt - 1) / bt, lds, ring);
    }

    // ---- debug getters: last generated sample's activations, reference layouts --------------
    void getXtOut(int layer, float* hXt) {
        gpuErrChk(hipMemcpy(hXt, m_XtOut + (size_t)layer * m_maxBatch * R, (size_t)m_maxBatch * R * sizeof(float),
                            hipMemcpyDefault));
    }
    void getSkipOut(int layer, float* hSkipOut) {
        gpuErrChk(hipMemcpy(hSkipOut, m_skipOut + (size_t)layer * m_maxBatch * S,
                            (size_t)m_maxBatch * S * sizeof(float), hipMemcpyDefault));
    }
    void getZs(float* hZs) { gpuErrChk(hipMemcpy(hZs, m_Zs, (size_t)m_maxBatch * A * sizeof(float), hipMemcpyDefault)); }
    void getZa(float* hZa) { gpuErrChk(hipMemcpy(hZa, m_Za, (size_t)m_maxBatch * A * sizeof(float), hipMemcpyDefault)); }
    void getP(float* hP) { gpuErrChk(hipMemcpy(hP, m_p, (size_t)m_maxBatch * A * sizeof(float), hipMemcpyDefault)); }
    // Columns [offset, offset + size) of every utterance's row, device -> caller (host or device),
    // asynchronously on `stream` (role of nv_wavenet.cuh:439-444): the sample buffer is [batch][stride]
    // on both sides, so this is one strided 2-D copy of `size` ints per row.
    void getYOut(int* yOut, int offset, int size, hipStream_t stream = 0) {
        const size_t rowBytes = (size_t)m_lastStride * sizeof(int);
        gpuErrChk(hipMemcpy2DAsync(yOut + offset, rowBytes, m_yOut + offset, rowBytes, (size_t)size * sizeof(int),
                                   (size_t)m_maxBatch, hipMemcpyDefault, stream));
    }

    // ---- generation --------------------------------------------------------------------------
    // Generates num_samples in pieces of num_samples_per_chunk and hands every finished piece to
    // consume(yOut, firstSample, count) on the calling thread (role of nv_wavenet.cuh:445-497).  Two
    // streams: generation of piece k+1 is enqueued right behind piece k and never waits for the host;
    // the device-to-caller copy of piece k runs on a second stream as soon as an event says piece k is
    // complete, so copies and consumers overlap the generation of later pieces.  Returns after the
    // last piece has been consumed.
    template <class Callback>
    bool run_chunks(int num_samples_per_chunk, Callback consume, int num_samples, int batch_size, int* yOut = NULL,
                    int batch_size_per_block = 1, bool dumpActivations = false, hipStream_t stream = 0) {
        (void)dumpActivations;
        assert(num_samples_per_chunk > 0);
        struct Piece {
            int first, count;
            hipEvent_t generated, delivered;
        };
        std::vector<Piece> pieces;
        for (int first = 0; first < num_samples; first += num_samples_per_chunk) {
            Piece pc;
            pc.first = first;
            pc.count = num_samples - first < num_samples_per_chunk ? num_samples - first : num_samples_per_chunk;
            pc.generated = pooledEvent(2 * pieces.size());
            pc.delivered = pooledEvent(2 * pieces.size() + 1);
            pieces.push_back(pc);
        }
        // (the events are the engine's, made once.  A second stream exists only where it buys something: when the samples or the PCM
        //  go to HOST memory, so that the copy of a chunk overlaps the generation of the next; device-resident outputs are copied in
        //  stream order, 20 us per chunk.  Made per call and destroyed at its end -- 1.4 ms -- because streams kept alive in the engine
        //  take hardware queues away from the caller's own streams: with them pooled, a launch on a caller's side stream queued up
        //  behind a kernel of another of its streams, tests/..::test_chain_launch_that_cannot_become_resident..)
        const bool hostOut = (yOut != NULL && !isDevicePtr(yOut)) || (m_pcmUser != NULL && !isDevicePtr(m_pcmUser));
        hipStream_t genStream = stream, outStream = stream;
        const bool ownGen = hostOut && !stream;
        if (ownGen) gpuErrChk(hipStreamCreate(&genStream));
        if (hostOut) gpuErrChk(hipStreamCreate(&outStream));

        bool ok = true;
        for (size_t k = 0; k < pieces.size(); k++) {
            const Piece& pc = pieces[k];
            m_num_samples_per_chunk = pc.count;
            // The reference dumps activations in every chunk (nv_wavenet.cuh:471, hard-coded true) and
            // its test reads them back afterwards; only the last chunk's dump can be observed, so only
            // the last chunk runs the dump-capable kernel variant.
            ok = run_partial(pc.first, num_samples, batch_size, NULL, batch_size_per_block, k + 1 == pieces.size(), genStream) && ok;
            gpuErrChk(hipEventRecord(pc.generated, genStream));
            if (hostOut) gpuErrChk(hipStreamWaitEvent(outStream, pc.generated, 0));
            if (yOut) getYOut(yOut, pc.first, pc.count, outStream);
            if (m_pcmUser) getAudioOut(m_pcmUser, pc.first, pc.count, outStream);
            gpuErrChk(hipEventRecord(pc.delivered, outStream));
        }
        m_num_samples_per_chunk = 0;
        for (size_t k = 0; k < pieces.size(); k++) {
            gpuErrChk(hipEventSynchronize(pieces[k].delivered));
            consume(yOut, pieces[k].first, pieces[k].count);
        }
        if (ownGen) gpuErrChk(hipStreamDestroy(genStream));
        if (hostOut) gpuErrChk(hipStreamDestroy(outStream));
        if (isChain() && chainStatus() != 0) ok = false;   // (everything has completed: the check costs nothing)
        return ok;
    }

    // Generates samples [init_sample, init_sample + chunk) continuing from device-resident state
    // (history, dilation ring); chunk = the run_chunks chunk, or num_samples (nv_wavenet.cuh:499-635).
    // Asynchronous on `stream`.  yOut (host or device, [batch][num_samples] ints) receives the
    // sample buffer when non-NULL.
    bool run_partial(int init_sample, int num_samples, int batch_size, int* yOut = NULL, int batch_size_per_block = 1,
                     bool dumpActivations = false, hipStream_t stream = 0) {
        assert(batch_size_per_block > 0 && batch_size_per_block < 5);
        assert(batch_size % batch_size_per_block == 0);
        assert(batch_size > 0 && batch_size <= m_maxBatch);
        assert(num_samples <= m_maxSamples);
        assert(m_condRaw != NULL || m_cond != NULL || m_condUser != NULL || m_featPtr != NULL);  // some conditioning has been handed over
        assert(m_featPtr == NULL || num_samples <= m_featSamples);
        assert(m_condRaw == NULL || num_samples <= m_condRawSamples);      // ... and the in-place tensor covers the run
        assert(m_condUser == NULL || num_samples <= m_condUserSamples);    // ... and so does a caller's packed buffer
        assert(m_pcmUser == NULL || m_pcmUserElems == 0 || m_pcmUserElems >= (size_t)batch_size * num_samples);
        if (m_implementation == SINGLE_BLOCK) assert(S <= 4 * R);
        if (!m_supported) return false;

        // a new utterance: the dilation rings read as zero until written (wavenet_wg takes x[t-d] = 0 for t < d from the ring
        // itself).  resetHistory() -- every way of handing over a new utterance's conditioning calls it -- has normally done this
        // already, outside the generation's critical path; this covers run() after run() on the same inputs.
        if (init_sample == 0) clearRings(stream);
        {
            // the tiles whose rings this launch writes: a wavenet_wg workgroup of BT tiles stores into the rings of ALL its tiles, the
            // padding tiles beyond the batch included (a later, larger batch must find those slots zero as well)
            int touched = (batch_size + 15) / 16;
            if (!(isChain() && !m_featPtr)) {
                int bt = wgTiles(touched);
                if (bt >= 3 && wg4Fits()) bt = 12;      // (three or four tiles per workgroup by launch: both roundings)
                touched = (touched + bt - 1) / bt * bt;
            }
            if (touched > m_tiles) touched = m_tiles;
            if (touched > m_ringDirtyTiles) m_ringDirtyTiles = touched;
        }
        wn::Params p;
        p.wblob = m_wblob;
        p.bias = m_bias;
        p.embPrev = m_embedPrev;
        p.embCur = m_embedCur;
        p.cond = m_condUser ? m_condUser : m_cond;
        p.condRaw = m_condRaw;
        p.condRawKind = m_condRaw ? m_condRawKind : 0;
        p.feat = NULL;
        if (m_featPtr) {
            // the conditioning is computed by wavenet_wg from the features: its own stream (with Wcond) and bias table (with bcond)
            if (m_featDirty) buildFeatStream(stream);
            p.wblob = m_wblobF;
            p.bias = m_biasF;
            p.feat = m_featPtr;
            p.condRawKind = 3;
        }
        p.gate = NULL;
        p.sel = m_outputSelectors;
        p.ring = m_ring;
        p.maxDilation = m_maxDilation;
        p.yInPrev = m_yInPrev;
        p.yInCur = m_yInCur;
        p.yOut = m_yOut;
        p.xtOut = m_XtOut;
        p.skipOut = m_skipOut;
        p.zs = m_Zs;
        p.za = m_Za;
        p.p = m_p;
        p.numLayers = m_numLayers;
        p.batch = batch_size;
        p.maxBatch = m_maxBatch;
        p.numSamples = num_samples;
        p.condSamples = m_featPtr ? m_featSamples : m_condRaw ? m_condRawSamples : m_maxSamples;
        p.initSample = init_sample;
        p.count = m_num_samples_per_chunk ? m_num_samples_per_chunk : num_samples;
        if (p.initSample + p.count > num_samples) p.count = num_samples - p.initSample;
        p.ringSlots = m_ringSlots;
        p.ldsRingD = 0;                 // (the wavenet_wg launchers place ring slots in the LDS their tables leave free: placeLdsRing)
        p.tiles = m_tiles;
        p.tileBase = 0;
        p.tanhEmbed = m_tanhEmbed ? 1 : 0;
        p.dump = dumpActivations ? 1 : 0;
        p.embLds = 0;
        p.useRng = m_useRng ? 1 : 0;
        p.rngKey0 = (unsigned)m_rngSeed;
        p.rngKey1 = (unsigned)(m_rngSeed >> 32);
        p.clk = m_clkOn ? m_clk : NULL;
        fillSchedule(p);
        m_lastStride = num_samples;
        if (p.count <= 0) return true;

        const int tiles = (batch_size + 15) / 16;
        // (computing the conditioning from the features is wavenet_wg's: a chain engine runs it for such a launch)
        bool result = (isChain() && !m_featPtr) ? launchChain(p, tiles, stream) : launchWg(p, tiles, stream);
        if (m_pcmUser != NULL) {
            // the indices of a finished sample are final: the expansion is a per-element map of yOut
            hipLaunchKernelGGL(wn::mulaw_pcm_kernel, dim3(gridFor((size_t)batch_size * p.count)), dim3(256), 0, stream,
                               m_yOut, m_pcm, m_mulaw, batch_size, num_samples, p.initSample, p.count);
            result = result && hipGetLastError() == hipSuccess;
        }
        if (yOut != NULL) {
            gpuErrChk(hipMemcpyAsync(yOut, m_yOut, (size_t)num_samples * batch_size * sizeof(int), hipMemcpyDefault, stream));
            if (m_pcmUser != NULL)
                gpuErrChk(hipMemcpyAsync(m_pcmUser, m_pcm, (size_t)num_samples * batch_size * sizeof(short),
                                         hipMemcpyDefault, stream));
        }
        return result;
    }

    // Samples [init_sample, init_sample + count) only, asynchronously on `stream` (what run_chunks does per
    // chunk, for callers that drive the chunks themselves, e.g. with the conditioning streamed in between)
    bool run_range(int init_sample, int count, int num_samples, int batch_size, hipStream_t stream = 0) {
        m_num_samples_per_chunk = count;
        const bool ok = run_partial(init_sample, num_samples, batch_size, NULL, 1, false, stream);
        m_num_samples_per_chunk = 0;
        return ok;
    }
    // the sample history back to 128 (what setInputs does) and a clean hand-off status, asynchronously on `stream`
    void resetHistory(hipStream_t stream = 0) {
        hipLaunchKernelGGL(wn::silence_kernel, dim3(1), dim3(256), 0, stream, m_yInPrev, m_yInCur, m_maxBatch);
        gpuErrChk(hipGetLastError());
        gpuErrChk(hipMemsetAsync(m_chainStatus, 0, sizeof(unsigned), stream));
        clearRings(stream);
    }
    // zeroes the dilation rings of the tiles that launches have written since the last clear (4 MiB per tile at C3: 1 ms at 12 288
    // utterances, once per utterance)
    void clearRings(hipStream_t stream) {
        if (m_ringDirtyTiles > 0)
            gpuErrChk(hipMemsetAsync(m_ring, 0, (size_t)m_ringDirtyTiles * m_ringSlots * R * 16 * sizeof(elem), stream));
        m_ringDirtyTiles = 0;
    }

    bool run(int num_samples, int batch_size, int* yOut = NULL, int batch_size_per_block = 1,
             bool dumpActivations = false, hipStream_t stream = 0) {
        m_num_samples_per_chunk = 0;
        return run_partial(0, num_samples, batch_size, yOut, batch_size_per_block, dumpActivations, stream);
    }

protected:
    void dropLhConditioning() {
        m_condRaw = NULL;
        m_condRawKind = 0;
        m_condUser = NULL;
    }
    // m_wblobF / m_biasF from m_wblob / m_bias / m_condW / m_condB, asynchronously on `stream`
    void buildFeatStream(hipStream_t stream) {
        assert(m_nCond > 0);
        const int L = m_numLayers;
        const size_t wElems = (size_t)C::NW * CF::waveStreamFrags(L) * C::FRAG_ELEMS;
        const int bTotal = L * C::BIAS_L + 2 * A;
        if (!m_wblobF) {
            gpuErrChk(hipMalloc(&m_wblobF, wElems * sizeof(elem)));
            gpuErrChk(hipMalloc(&m_biasF, (size_t)bTotal * sizeof(float)));
            std::vector<int2> map;
            for (int l = 0; l < L; l++)
                for (int i = 0; i < C::FLW; i++) map.push_back(make_int2((int)C::streamPos(l, i, L), (int)CF::streamPos(l, i, L)));
            for (int q = 0; q < C::FHW; q++)
                map.push_back(make_int2((int)C::headOffsetFrags(L) + C::headFrag(q), (int)CF::headOffsetFrags(L) + CF::headFrag(q)));
            m_restreamN = (int)map.size();
            gpuErrChk(hipMalloc(&m_restream, map.size() * sizeof(int2)));
            gpuErrChk(hipMemcpy(m_restream, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice));
        }
        gpuErrChk(hipMemsetAsync(m_wblobF, 0, wElems * sizeof(elem), stream));
        hipLaunchKernelGGL(wn::restream_kernel, dim3(gridFor((size_t)C::NW * m_restreamN * 64)), dim3(256), 0, stream, (wn::uintx4*)m_wblobF,
                           (const wn::uintx4*)m_wblob, (const int2*)m_restream, m_restreamN, C::waveStreamFrags(L), CF::waveStreamFrags(L), C::NW);
        gpuErrChk(hipGetLastError());
        for (int l = 0; l < L; l++) {
            hipLaunchKernelGGL((wn::pack_weight_kernel<F16>), dim3(gridFor((size_t)2 * R * KC)), dim3(256), 0, stream,
                               m_wblobF + CF::streamPos(l, CF::O_COND, L) * C::FRAG_ELEMS, m_condW + (size_t)l * KC * 2 * R, 2 * R, KC, C::NW,
                               CF::waveStreamFrags(L) * C::FRAG_ELEMS, R / 16);
            gpuErrChk(hipGetLastError());
        }
        hipLaunchKernelGGL((wn::feat_bias_kernel<F16>), dim3(gridFor(bTotal)), dim3(256), 0, stream, m_biasF, m_bias, m_condB, L, R, C::BIAS_L,
                           bTotal);
        gpuErrChk(hipGetLastError());
        m_featDirty = false;
    }
    size_t ringElems() const { return (size_t)m_tiles * m_ringSlots * R * 16; }
    unsigned statusWord(int i) {
        unsigned s = 0;
        gpuErrChk(hipDeviceSynchronize());
        gpuErrChk(hipMemcpy(&s, m_chainStatus + i, sizeof(unsigned), hipMemcpyDeviceToHost));
        return s;
    }
    // tiles per workgroup of THIS launch: the four-tile kernels exist dump-free and for the packed conditioning only (with the conditioning
    // computed in the kernel a fourth tile does not fit the register file: 2-31 spilled registers by shape, and no faster -- LABNOTES round 6)
    int launchTiles(int tiles, bool dump, int raw) const {
        const int bt = wgTiles(tiles);
        return (bt == 4 && (dump || raw != 0)) ? 3 : bt;
    }
    // wavenet_wg by batch size: one to four tiles per workgroup
    bool launchWg(wn::Params& p, int tiles, hipStream_t stream) {
        const int bt = launchTiles(tiles, p.dump != 0, p.condRawKind);
        if (bt == 4) {
            if constexpr (WG4) return launchE<4, false, 0>(p, tiles, stream);
            return false;
        }
        if (bt == 3) {
            if constexpr (WG3) return launch<3>(p, tiles, stream);
            return false;
        }
        if (bt == 2) {
            if constexpr (WG2) return launch<2>(p, tiles, stream);
            return false;
        }
        return launch<1>(p, tiles, stream);
    }
    // a chain launch that gives up can be re-run by wavenet_wg when the model's bias table fits one workgroup's LDS
    bool chainHasFallback() const { return ldsFits<1>(); }

    // The multi-CU chain: every (tile, stage) workgroup must be resident at the same time, so tiles are launched in
    // groups of at most CUs / stages chains; mailboxes are re-zeroed before every launch.  Residency cannot be
    // guaranteed when other work shares the GPU, so every chain launch is bracketed, in stream order and without any
    // host round trip: (1) rings and history of the launch's tiles are copied aside; (2) the chain runs; (3) if it
    // gave up (status word set by the first spin that timed out), chain_restore_kernel puts rings and history back
    // and a wavenet_wg launch gated on the status word generates the launch's samples instead (arithmetic and state
    // layouts are the same in both organisations, bit for bit); (4) chain_settle_kernel counts the event and clears
    // the word.  When the chain completes, (3) is two empty launches.
    bool launchChain(wn::Params& p, int tiles, hipStream_t stream) {
        wn::ChainParams cp;
        cp.mail = m_mail;
        cp.status = m_chainStatus;
        cp.stages = m_chainStages;
        cp.lpc = m_chainLpc;
        cp.timeoutTicks = m_chainTimeoutTicks;
        const int perLaunch = m_numCUs / m_chainStages;
        if (perLaunch < 1) return false;
        const bool dump = p.dump != 0;
        const bool fallback = m_ringShadow != NULL;
        const size_t ringTileElems = (size_t)m_ringSlots * R * 16;
        // tiles beyond the chains that are resident at once ride the same chains, up to TPC_MAX per chain (round 5); more than
        // that takes further launches
        const int tpc = chainTpc((m_maxBatch + 15) / 16);        // (what the mailboxes were sized for)
        cp.tpc = tpc;
        const int tilesPerLaunch = perLaunch * tpc;
        for (int t0 = 0; t0 < tiles; t0 += tilesPerLaunch) {
            const int nT = tiles - t0 < tilesPerLaunch ? tiles - t0 : tilesPerLaunch;
            cp.tile0 = t0;
            cp.ntiles = nT;
            cp.chains = nT < perLaunch ? nT : perLaunch;
            const int b0 = t0 * 16, nb = (p.batch - b0 < nT * 16 ? p.batch - b0 : nT * 16);
            if (fallback) {
                gpuErrChk(hipMemcpyAsync(m_ringShadow + t0 * ringTileElems, m_ring + t0 * ringTileElems,
                                         nT * ringTileElems * sizeof(elem), hipMemcpyDeviceToDevice, stream));
                gpuErrChk(hipMemcpyAsync(m_histShadow + b0, m_yInPrev + b0, nb * sizeof(int), hipMemcpyDeviceToDevice, stream));
                gpuErrChk(hipMemcpyAsync(m_histShadow + m_maxBatch + b0, m_yInCur + b0, nb * sizeof(int), hipMemcpyDeviceToDevice, stream));
            }
            gpuErrChk(hipMemsetAsync(m_mail, 0, CC::mailGranules(cp.chains, m_chainStages, tpc) * sizeof(unsigned long long), stream));
            const int grid = 8 * m_chainStages * ((cp.chains + 7) / 8);
            p.embLds = CC::embTables();
            p.gate = NULL;
            if (dump) {
                hipLaunchKernelGGL((wn::wavenet_chain<F16, R, S, A, true>), dim3(grid), dim3(C::THREADS), CC::ldsBytes(), stream, p, cp);
            } else {
                bool hoisted = false;
                if constexpr (kChainHoistBuilt) {
                    if (chainHoists(tpc)) {
                        hipLaunchKernelGGL((wn::wavenet_chain<F16, R, S, A, false, true>), dim3(grid), dim3(C::THREADS), CC::ldsBytes(), stream, p, cp);
                        hoisted = true;
                    }
                }
                if (!hoisted) hipLaunchKernelGGL((wn::wavenet_chain<F16, R, S, A, false>), dim3(grid), dim3(C::THREADS), CC::ldsBytes(), stream, p, cp);
            }
            if (hipGetLastError() != hipSuccess) return false;
            if (fallback) {
                const size_t n16 = nT * ringTileElems * sizeof(elem) / 16;
                hipLaunchKernelGGL(wn::chain_restore_kernel, dim3(gridFor(n16)), dim3(256), 0, stream, (const unsigned*)m_chainStatus,
                                   (wn::uintx4*)(m_ring + t0 * ringTileElems), (const wn::uintx4*)(m_ringShadow + t0 * ringTileElems), n16,
                                   m_yInPrev + b0, m_yInCur + b0, (const int*)(m_histShadow + b0), (const int*)(m_histShadow + m_maxBatch + b0), nb);
                if (hipGetLastError() != hipSuccess) return false;
                // the same samples for the same tiles on wavenet_wg, one tile per workgroup, only if the chain gave up
                wn::Params q = p;
                q.gate = m_chainStatus;
                q.batch = b0 + nb;
                const int nEmb = embTables<1>();
                q.embLds = nEmb;
                bool ok;
                if (nEmb) ok = launchGated<true>(q, t0, nT, nEmb, stream);
                else ok = launchGated<false>(q, t0, nT, 0, stream);
                if (!ok) return false;
                hipLaunchKernelGGL(wn::chain_settle_kernel, dim3(1), dim3(1), 0, stream, m_chainStatus);
                if (hipGetLastError() != hipSuccess) return false;
            }
        }
        return true;
    }
    // wavenet_wg<BT = 1> for tiles [tile0, tile0 + ntiles), gated on Params::gate
    template <bool EMB> bool launchGated(wn::Params& q, int tile0, int ntiles, int nEmb, hipStream_t stream) {
        q.tileBase = tile0;
        const int kind = q.condRawKind;
        const bool dump = q.dump != 0 || (!F16 && kind != 0);      // (see launch(): the fp32 dump-free kernel is the packed one)
        const size_t lds = ldsNeed<1>(m_numLayers, nEmb, dump);
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(ntiles), dim3(C::THREADS), lds, stream, q);
            return hipGetLastError() == hipSuccess;
        };
        if (!dump) {
            if constexpr (F16) {
                if (kind == 2) return go(wn::wavenet_wg<F16, R, S, A, 1, EMB, false, 2>);
                if (kind == 1) return go(wn::wavenet_wg<F16, R, S, A, 1, EMB, false, 1>);
            }
            return go(wn::wavenet_wg<F16, R, S, A, 1, EMB, false, 0>);
        }
        if constexpr (F16) {
            if (kind == 2) return go(wn::wavenet_wg<F16, R, S, A, 1, EMB, true, 2>);
        }
        if (kind == 1) return go(wn::wavenet_wg<F16, R, S, A, 1, EMB, true, 1>);
        return go(wn::wavenet_wg<F16, R, S, A, 1, EMB, true, 0>);
    }
};
