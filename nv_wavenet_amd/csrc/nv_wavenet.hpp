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
//                           streamed from L2 every sample: wn::wavenet_wg (1 or 2 tiles of 16 utterances
//                           per workgroup) or, beyond two tiles per CU, wn::wavenet_stream
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
#include "wn_pipe.hpp"
#include "wn_stream.hpp"

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
    NVW_ORG_WG = 1,       // wn::wavenet_wg, 1 or 2 tiles per workgroup by batch size
    NVW_ORG_WG1 = 2,      // wn::wavenet_wg, one tile per workgroup
    NVW_ORG_WG2 = 3,      // wn::wavenet_wg, two tiles per workgroup
    NVW_ORG_STREAM = 4,   // wn::wavenet_stream (loader / consumer waves, 4 tiles per workgroup)
    NVW_ORG_CHAIN = 5,    // wn::wavenet_chain, as many layers per CU as stay resident
    NVW_ORG_CHAIN1 = 6,   // wn::wavenet_chain, one layer per CU
    NVW_ORG_PIPE = 7,     // wn::wavenet_pipe: the chain kept full (groups of tiles in flight per chain), large batches
    NVW_ORG_WG3 = 8       // wn::wavenet_wg, three tiles per workgroup (fp16, R <= 64)
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
    using SC = wn::SCfg<F16, R, S, A>;     // throughput (loader/consumer) kernel
    using CC = wn::CCfg<F16, R, S, A>;     // multi-CU chain
    using PC = wn::PCfg<F16, R, S, A>;     // multi-CU chain kept full (throughput)
    using elem = typename wn::Prec<F16>::elem;

    Implementation m_implementation;
    int m_numLayers, m_maxBatch, m_maxSamples, m_maxDilation, m_tiles, m_numCUs;
    int m_org;           // resolved organisation: NVW_ORG_WG1 / WG2 / WG (by batch at run time) / STREAM / CHAIN / CHAIN1
    bool m_streamMode;   // weights / conditioning packed for wn::wavenet_stream (else: per-wave streams of wavenet_wg / chain)
    bool m_supported;    // false: this shape does not fit the CU (run() returns false, like the reference's unsupported variants)
    int m_streamNS;      // LDS ring slots of the throughput kernel
    int m_chainLpc, m_chainStages;   // layers per chain stage, stages (layer stages + head)
    int m_pipeChains, m_pipeGroups;  // wavenet_pipe: chains, groups of PC::G tiles per chain (0: not this organisation)
    bool m_tanhEmbed;
    int m_num_samples_per_chunk;
    int m_ringSlots;
    int m_lastStride;    // row stride of m_yOut in the latest launch (= its num_samples)

    elem* m_wblob;      // packed weight fragments: L layers then the head
    float* m_bias;      // fp32 biases
    elem* m_embedPrev;  // [A][R]
    elem* m_embedCur;
    elem* m_cond;       // packed conditioning
    int m_condRawSamples;
    const float* m_condRaw;   // or: the caller's fp32 [N][L][maxBatch][2R] device tensor, consumed in place (setInputsDirect)
    float* m_outputSelectors;
    elem* m_ring;
    int *m_yInPrev, *m_yInCur, *m_yOut;
    float *m_XtOut, *m_skipOut, *m_Zs, *m_Za, *m_p;
    unsigned long long* m_mail;   // chain mailboxes
    unsigned* m_chainStatus;      // [0] first time-out code of a chain launch (0 = fine)
    size_t m_mailBytes;

    float* m_stage;     // device staging for fp32 uploads from host pointers
    size_t m_stageElems;

    // extensions beyond the reference: in-kernel selectors, int16 PCM output
    bool m_useRng;
    unsigned long long m_rngSeed;
    short* m_pcm;       // [maxBatch][maxSamples] int16, allocated on first setAudioOut
    short* m_mulaw;     // [A] PCM value of every sample index
    short* m_pcmUser;   // caller's buffer (host or device), filled wherever yOut is

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
    // same for the shared stream of the throughput kernel (fragment offset inside the one stream)
    void packWeightStream(size_t fragOff, const float* src, int M, int K, int rowperm, int gate = 0) {
        const float* d = onDevice(src, (size_t)M * K);
        hipLaunchKernelGGL((wn::pack_weight_stream_kernel<F16>), dim3(gridFor((size_t)M * K)), dim3(256), 0, 0,
                           m_wblob + fragOff * SC::FRAG_ELEMS, d, M, K, rowperm, gate);
        gpuErrChk(hipGetLastError());
    }
    void convertTo(elem* dst, const float* src, size_t n) {
        const float* d = onDevice(src, n);
        hipLaunchKernelGGL((wn::convert_kernel<F16>), dim3(gridFor(n)), dim3(256), 0, 0, dst, d, n);
        gpuErrChk(hipGetLastError());
    }
    template <int BT> static size_t ldsNeed(int L, int embTables) { return wn::Cfg<F16, R, S, A, BT>::ldsBytes(L, embTables); }
    static constexpr size_t kLdsMax = 160 * 1024;
    template <int BT> bool ldsFits() const { return ldsNeed<BT>(m_numLayers, 0) <= kLdsMax; }
    // how many embedding tables fit in LDS beside everything else: 2, 1 (current tap) or 0
    template <int BT> int embTables() const {
        return ldsNeed<BT>(m_numLayers, 2) <= kLdsMax ? 2 : ldsNeed<BT>(m_numLayers, 1) <= kLdsMax ? 1 : 0;
    }
    // DUMP = false (no activation dump code at all) exists for the fp16 engine, the production path;
    // the fp32 engine is the parity mode and always carries the dump
    template <int BT, bool EMB, bool DUMP, bool RAW> bool launchK(wn::Params& p, int tiles, int nEmb, hipStream_t stream) {
        using CB = wn::Cfg<F16, R, S, A, BT>;
        const int grid = (tiles + BT - 1) / BT;
        p.embLds = nEmb;
        hipLaunchKernelGGL((wn::wavenet_wg<F16, R, S, A, BT, EMB, DUMP, RAW>), dim3(grid), dim3(CB::THREADS),
                           ldsNeed<BT>(m_numLayers, nEmb), stream, p);
        return hipGetLastError() == hipSuccess;
    }
    template <int BT, bool DUMP, bool RAW> bool launchE(wn::Params& p, int tiles, hipStream_t stream) {
        const int nEmb = embTables<BT>();
        return nEmb ? launchK<BT, true, DUMP, RAW>(p, tiles, nEmb, stream) : launchK<BT, false, DUMP, RAW>(p, tiles, 0, stream);
    }
    template <int BT> bool launch(wn::Params& p, int tiles, hipStream_t stream) {
        const bool raw = p.condRaw != NULL;
        if constexpr (F16) {
            if (!p.dump) return raw ? launchE<BT, false, true>(p, tiles, stream) : launchE<BT, false, false>(p, tiles, stream);
        }
        return raw ? launchE<BT, true, true>(p, tiles, stream) : launchE<BT, true, false>(p, tiles, stream);
    }
    template <int BT, bool EMB, bool DUMP, bool RAW> void allowLdsK() {
        const size_t need = ldsNeed<BT>(m_numLayers, EMB ? embTables<BT>() : 0);
        if (need <= kLdsMax)
            gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_wg<F16, R, S, A, BT, EMB, DUMP, RAW>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    }
    template <int BT> void allowLds() {
        allowLdsK<BT, false, true, false>();
        allowLdsK<BT, true, true, false>();
        allowLdsK<BT, false, true, true>();
        allowLdsK<BT, true, true, true>();
        if constexpr (F16) {
            allowLdsK<BT, false, false, false>();
            allowLdsK<BT, true, false, false>();
            allowLdsK<BT, false, false, true>();
            allowLdsK<BT, true, false, true>();
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
    bool chainFits(int lpc, int tiles) const { return lpc > 0 && chainStagesFor(m_numLayers, lpc) * tiles <= m_numCUs; }
    bool streamFits() const { return m_streamNS >= SC::MIN_NS; }
    // The single-workgroup organisations by batch size: up to three tiles per CU run in the latency kernel
    // (one, two or -- fp16, R <= 64 -- three tiles split over the 4 SIMDs of a CU); beyond that every SIMD gets
    // its own tile and the weights are streamed once per CU through an LDS ring.
    int singleOrg(int tiles) const {
        if (tiles > 2 * m_numCUs && tiles <= 3 * m_numCUs && wg3Fits()) return NVW_ORG_WG3;
        return (tiles > 2 * m_numCUs && streamFits()) ? NVW_ORG_STREAM : NVW_ORG_WG;
    }
    // Per-sample time models (microseconds) of the organisations that can run `tiles` tiles, from the
    // shape: weight bytes per sample W, layers L, CUs.  Constants measured on MI355X:
    // a CU streams 58 B/clk of weights at ~2.1 GHz beside ~0.45 us of dependent chain per layer; a chain
    // stage costs one ~1.1 us hand-off plus ~0.5 us per layer with resident weights.
    int pickOrganisation(int tiles) const {
        const int single = singleOrg(tiles);
        const int lpc = chainLpcMax(m_numLayers);
        if (!chainFits(lpc, tiles)) return single;
        const double wBytes = sizeof(elem) * ((double)m_numLayers * (5.0 * R * R + (double)S * R) + (double)A * S + (double)A * A);
        const double tStream = wBytes / (58.0 * 2100.0) + 0.25 * m_numLayers + 4.0;    // us: L1 weight stream + chain + head
        const double tChain = 1.1 * chainStagesFor(m_numLayers, lpc) + (R >= 128 ? 0.55 : 0.4) * m_numLayers + 4.0;
        return tChain < tStream ? NVW_ORG_CHAIN : single;
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
        if (org == NVW_ORG_STREAM && !streamFits()) org = NVW_ORG_WG;
        if (org == NVW_ORG_CHAIN && !chainFits(chainLpcMax(m_numLayers), tiles)) org = singleOrg(tiles);
        if (org == NVW_ORG_CHAIN1 && !(CC::SUPPORTED && chainFits(1, tiles))) org = singleOrg(tiles);
        m_pipeChains = m_pipeGroups = 0;
        if (org == NVW_ORG_PIPE) {
            if (!pipeGeometry(tiles, m_pipeChains, m_pipeGroups)) {
                m_pipeChains = m_pipeGroups = 0;
                org = singleOrg(tiles);
            }
        }
        m_org = org;
        m_streamMode = org == NVW_ORG_STREAM;
        m_chainLpc = org == NVW_ORG_CHAIN ? chainLpcMax(m_numLayers) : org == NVW_ORG_CHAIN1 ? 1 : org == NVW_ORG_PIPE ? pipeLpc(m_numLayers) : 0;
        m_chainStages = m_chainLpc ? chainStagesFor(m_numLayers, m_chainLpc) : 0;
    }
    bool isChain() const { return m_chainLpc > 0 && m_pipeGroups == 0; }
    bool isPipe() const { return m_pipeGroups > 0; }
    // wavenet_pipe geometry for `tiles` tiles: as many chains as the GPU holds, groups of PC::G tiles per chain
    static int pipeLpc(int L) {
        if (!PC::SUPPORTED) return 0;
        const int ns = (L + PC::LPC - 1) / PC::LPC;
        return (L + ns - 1) / ns;
    }
    bool pipeGeometry(int tiles, int& chains, int& groups) const {
        const int lpc = pipeLpc(m_numLayers);
        if (lpc == 0) return false;
        const int maxChains = m_numCUs / chainStagesFor(m_numLayers, lpc);
        if (maxChains < 1) return false;
        const int perChain = (tiles + maxChains - 1) / maxChains;
        groups = (perChain + PC::G - 1) / PC::G;
        if (groups > PC::MAX_GROUPS) return false;
        chains = (tiles + groups * PC::G - 1) / (groups * PC::G);
        return true;
    }
    // tiles per workgroup of wn::wavenet_wg for a batch of `tiles` tiles
    static constexpr bool WG3 = F16 && R <= 64;   // shapes with a three-tile instantiation
    bool wg3Fits() const {
        if constexpr (WG3) return ldsFits<3>();
        return false;
    }
    int wgTiles(int tiles) const {
        if (m_org == NVW_ORG_WG3 && wg3Fits()) return 3;
        const bool two = m_org == NVW_ORG_WG2 || m_org == NVW_ORG_WG3 || (m_org == NVW_ORG_WG && tiles > m_numCUs);
        return (two && ldsFits<2>()) ? 2 : 1;
    }

public:
    nvWavenetInfer(int numLayers, int maxDilation, int batchSize, int numSamples, int impl = 0,
                   bool tanhEmbed = true, int organisation = NVW_ORG_AUTO)
        : m_implementation((Implementation)impl), m_numLayers(numLayers), m_maxBatch(batchSize),
          m_maxSamples(numSamples), m_maxDilation(maxDilation), m_tanhEmbed(tanhEmbed),
          m_num_samples_per_chunk(0), m_lastStride(numSamples), m_condRaw(NULL), m_mail(NULL), m_chainStatus(NULL), m_mailBytes(0),
          m_stage(NULL), m_stageElems(0), m_useRng(false), m_rngSeed(0), m_pcm(NULL), m_mulaw(NULL), m_pcmUser(NULL),
          m_stageUsed(0) {
        assert(numLayers >= 2 && batchSize > 0 && numSamples > 0 && maxDilation > 0);
        assert(numLayers <= wn::kMaxLayers);
        {
            int dev = 0;
            hipDeviceProp_t prop;
            gpuErrChk(hipGetDevice(&dev));
            gpuErrChk(hipGetDeviceProperties(&prop, dev));
            m_numCUs = prop.multiProcessorCount;
        }
        {
            const size_t biasBytes = ((size_t)numLayers * SC::BIAS_L + 2 * A) * sizeof(float);
            long ns = ((long)kLdsMax - (long)biasBytes) / ((long)SC::CH * 1024);
            m_streamNS = (int)(ns > 6 ? 6 : ns);
        }
        resolveOrganisation(organisation);
        // The bias table of the whole model lives in the LDS of a wavenet_wg workgroup: a model whose
        // table does not fit cannot run there (the reference prints and returns false for shapes a
        // variant does not support, nv_wavenet_singleblock.cuh:273-286)
        m_supported = isChain() || isPipe() || m_streamMode || ldsFits<1>();
        if (!m_supported)
            fprintf(stderr, "nvWavenetInfer: R=%d S=%d A=%d with %d layers needs %zu bytes of LDS (> 160 KiB): unsupported\n", R,
                    S, A, numLayers, ldsNeed<1>(numLayers, 0));

        // conditioning / ring are allocated for whole workgroups: 4 tiles (throughput kernel), 2 (two tiles
        // per workgroup may be chosen), else exactly the tiles of the batch
        {
            const int tiles = (batchSize + 15) / 16;
            const int group = m_streamMode ? 4 : (isChain() || isPipe()) ? 1 : wgTiles(tiles);
            m_tiles = (tiles + group - 1) / group * group;
            if (isPipe()) m_tiles = m_pipeChains * m_pipeGroups * PC::G;
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

        const size_t wElems = m_streamMode ? SC::streamFrags(numLayers) * SC::FRAG_ELEMS
                                           : (size_t)C::NW * C::waveStreamFrags(numLayers) * C::FRAG_ELEMS;
        gpuErrChk(hipMalloc(&m_wblob, wElems * sizeof(elem)));
        gpuErrChk(hipMemset(m_wblob, 0, wElems * sizeof(elem)));
        const size_t bElems = (size_t)numLayers * C::BIAS_L + 2 * A;
        gpuErrChk(hipMalloc(&m_bias, bElems * sizeof(float)));
        gpuErrChk(hipMemset(m_bias, 0, bElems * sizeof(float)));
        gpuErrChk(hipMalloc(&m_embedPrev, (size_t)A * R * sizeof(elem)));
        gpuErrChk(hipMalloc(&m_embedCur, (size_t)A * R * sizeof(elem)));
        gpuErrChk(hipMemset(m_embedPrev, 0, (size_t)A * R * sizeof(elem)));
        gpuErrChk(hipMemset(m_embedCur, 0, (size_t)A * R * sizeof(elem)));

        const size_t condElems = (size_t)(numSamples + 1) * numLayers * m_tiles * 16 * 2 * R;   // + one padding sample
        gpuErrChk(hipMalloc(&m_cond, condElems * sizeof(elem)));
        gpuErrChk(hipMemset(m_cond, 0, condElems * sizeof(elem)));
        gpuErrChk(hipMalloc(&m_outputSelectors, (size_t)numSamples * batchSize * sizeof(float)));
        gpuErrChk(hipMemset(m_outputSelectors, 0, (size_t)numSamples * batchSize * sizeof(float)));

        const size_t ringElems = (size_t)m_tiles * m_ringSlots * R * 16;
        gpuErrChk(hipMalloc(&m_ring, ringElems * sizeof(elem)));
        gpuErrChk(hipMemset(m_ring, 0, ringElems * sizeof(elem)));

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
        if (isPipe()) {
            m_mailBytes = PC::mailGranules(m_pipeChains, m_chainStages, m_pipeGroups) * sizeof(unsigned long long);
            gpuErrChk(hipMalloc(&m_mail, m_mailBytes));
            gpuErrChk(hipMemset(m_mail, 0, m_mailBytes));
            gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_pipe<F16, R, S, A, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)PC::ldsBytes()));
            if constexpr (F16)
                gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_pipe<F16, R, S, A, false>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)PC::ldsBytes()));
        }
        if (isChain()) {
            m_mailBytes = CC::mailGranules((batchSize + 15) / 16, m_chainStages) * sizeof(unsigned long long);
            gpuErrChk(hipMalloc(&m_mail, m_mailBytes));
            gpuErrChk(hipMemset(m_mail, 0, m_mailBytes));
            gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_chain<F16, R, S, A, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)CC::ldsBytes()));
            if constexpr (F16)
                gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_chain<F16, R, S, A, false>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)CC::ldsBytes()));
        }

        hipLaunchKernelGGL(wn::silence_kernel, dim3(1), dim3(256), 0, 0, m_yInPrev, m_yInCur, m_maxBatch);
        gpuErrChk(hipGetLastError());

        if (m_supported && !isChain() && !isPipe() && !m_streamMode) {
            allowLds<1>();
            allowLds<2>();
            if constexpr (WG3) allowLds<3>();
        }
        if (m_streamMode) {
            gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_stream<F16, R, S, A, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)SC::ldsBytes(numLayers, m_streamNS)));
            if constexpr (F16)
                gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_stream<F16, R, S, A, false>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)SC::ldsBytes(numLayers, m_streamNS)));
        }
        gpuErrChk(hipDeviceSynchronize());
    }

    virtual ~nvWavenetInfer() {
        gpuErrChk(hipDeviceSynchronize());
        gpuErrChk(hipFree(m_wblob));
        gpuErrChk(hipFree(m_bias));
        gpuErrChk(hipFree(m_embedPrev));
        gpuErrChk(hipFree(m_embedCur));
        gpuErrChk(hipFree(m_cond));
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
        if (m_stage) gpuErrChk(hipFree(m_stage));
        if (m_pcm) gpuErrChk(hipFree(m_pcm));
        if (m_mulaw) gpuErrChk(hipFree(m_mulaw));
    }

    // false: the shape does not fit this GPU's CUs in the chosen organisation; run() returns false
    bool supported() const { return m_supported; }
    // 0 when every multi-CU launch so far ran to completion; else the code of the first hand-off that
    // timed out (0x100+stage: x, 0x200+stage: skip sums, 0x300: head).  Synchronises the device.
    unsigned chainStatus() {
        unsigned s = 0;
        gpuErrChk(hipDeviceSynchronize());
        gpuErrChk(hipMemcpy(&s, m_chainStatus, sizeof(unsigned), hipMemcpyDeviceToHost));
        return s;
    }

    // ---- model upload: fp32 in, host or device pointers, data is copied ---------------------
    // embedPrev / embedCur: [A][R]   (nv_wavenet.cuh:396-399)
    virtual void setEmbeddings(float* embedPrev, float* embedCur) {
        stageBegin((size_t)2 * A * R);
        convertTo(m_embedPrev, embedPrev, (size_t)A * R);
        convertTo(m_embedCur, embedCur, (size_t)A * R);
        gpuErrChk(hipStreamSynchronize(0));
    }
    // col-major Wprev,Wcur 2RxR; Bh 2R; Wres RxR; Bres R; Wskip SxR; Bskip S (nv_wavenet.cuh:400-409)
    virtual void setLayerWeights(int layer, float* Wprev, float* Wcur, float* Bh, float* Wres, float* Bres,
                                 float* Wskip, float* Bskip) {
        assert(layer >= 0 && layer < m_numLayers);
        stageBegin((size_t)5 * R * R + (size_t)S * R + 3 * R + S + 32);
        float* b = m_bias + (size_t)layer * C::BIAS_L;
        if (m_streamMode) {
            const size_t sf = (size_t)layer * SC::FLP;
            packWeightStream(sf + SC::O_PREV, Wprev, 2 * R, R, 0, 1);
            packWeightStream(sf + SC::O_CUR, Wcur, 2 * R, R, 0, 1);
            packWeightStream(sf + SC::O_RES, Wres, R, R, 0);
            // the skip GEMM of layer l is consumed one body later (the head body after the last layer)
            const size_t skipAt = (layer + 1 < m_numLayers) ? sf + SC::FLP + SC::O_SKIP
                                                            : (size_t)m_numLayers * SC::FLP + SC::H_SKIP;
            packWeightStream(skipAt, Wskip, S, R, 0);
            gpuErrChk(hipMemcpyAsync(b, Bh, 2 * R * sizeof(float), hipMemcpyDefault, 0));
            if constexpr (F16) {   // the fp16 gate works on pre-scaled pre-activations (wn::gate1)
                hipLaunchKernelGGL((wn::scale_gate_bias_kernel<F16>), dim3(1), dim3(256), 0, 0, b, R);
                gpuErrChk(hipGetLastError());
            }
            gpuErrChk(hipMemcpyAsync(b + 2 * R, Bres, R * sizeof(float), hipMemcpyDefault, 0));
            gpuErrChk(hipMemcpyAsync(b + 3 * R, Bskip, S * sizeof(float), hipMemcpyDefault, 0));
        } else {
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
        }
        gpuErrChk(hipStreamSynchronize(0));
    }
    // col-major Wzs AxS, Bzs A, Wza AxA, Bza A (nv_wavenet.cuh:410-415)
    virtual void setOutWeights(float* Wzs, float* Bzs, float* Wza, float* Bza) {
        stageBegin((size_t)A * S + (size_t)A * A + 16);
        const size_t hf = C::headOffsetFrags(m_numLayers);
        if (m_streamMode) {
            const size_t sh = (size_t)m_numLayers * SC::FLP;
            packWeightStream(sh + SC::H_ZS, Wzs, A, S, 0);
            packWeightStream(sh + SC::H_ZA, Wza, A, A, 1);   // lane-contiguous logit rows
        } else {
            packWeight(hf + C::O_ZS, Wzs, A, S, 0);
            packWeight(hf + C::O_ZA, Wza, A, A, 0);
        }
        gpuErrChk(hipMemcpyAsync(headBias(), Bzs, A * sizeof(float), hipMemcpyDefault, 0));
        gpuErrChk(hipMemcpyAsync(headBias() + A, Bza, A * sizeof(float), hipMemcpyDefault, 0));
        gpuErrChk(hipStreamSynchronize(0));
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
        m_condRaw = NULL;
        gpuErrChk(hipMemsetAsync(m_chainStatus, 0, sizeof(unsigned), stream));   // a new utterance starts from a clean state
        hipLaunchKernelGGL(wn::silence_kernel, dim3(1), dim3(256), 0, stream, m_yInPrev, m_yInCur, m_maxBatch);
        gpuErrChk(hipGetLastError());
        packConditioning(Lh, 0, numSamples, stream);
        gpuErrChk(hipStreamSynchronize(stream));
    }
    // Packs samples [firstSample, firstSample + count) of the conditioning (Lh points at sample firstSample)
    // into the engine's fragment order, asynchronously on `stream` when Lh is device memory: lets a caller
    // stream the conditioning chunk by chunk behind run_partial() of the previous chunk.
    void packConditioning(float* Lh, int firstSample, int count, hipStream_t stream = 0) {
        assert(firstSample >= 0 && count > 0 && firstSample + count <= m_maxSamples);
        const size_t rows = (size_t)count * m_numLayers;
        const size_t srcPerRow = (size_t)m_maxBatch * 2 * R;
        const size_t dstPerRow = (size_t)m_tiles * 16 * 2 * R;
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
            if (m_streamMode)
                hipLaunchKernelGGL((wn::pack_cond_tiled_kernel<F16, R, true>), dim3(grid), dim3(256), 0, stream,
                                   m_cond + ((size_t)firstSample * m_numLayers + r0) * dstPerRow, src, nr, m_maxBatch, m_tiles);
            else
                hipLaunchKernelGGL((wn::pack_cond_tiled_kernel<F16, R, false>), dim3(grid), dim3(256), 0, stream,
                                   dst0 + r0 * dstPerRow, src, nr, m_maxBatch, m_tiles);
            gpuErrChk(hipGetLastError());
        }
    }
    // Device-resident conditioning WITHOUT the copy (the reference's own recommendation, README.md:44; SURVEY.md 8f
    // rank 1): Lh is the caller's fp32 [numSamples][L][maxBatch][2R] tensor in device memory; the kernels read it in
    // place (16 bytes per lane and gate tile) and nothing is packed.  The caller keeps it alive and unchanged until
    // the run calls that follow have completed.  Resets the sample history like setInputs.  The loader / consumer
    // kernel and wavenet_pipe have no in-place path: there (and for host pointers) this is setConditioning.
    void setConditioningDirect(float* Lh, int numSamples) {
        assert(numSamples > 0 && numSamples <= m_maxSamples);
        if (m_streamMode || isPipe() || !isDevicePtr(Lh)) {
            setConditioning(Lh, numSamples);
            return;
        }
        hipLaunchKernelGGL(wn::silence_kernel, dim3(1), dim3(256), 0, 0, m_yInPrev, m_yInCur, m_maxBatch);
        gpuErrChk(hipGetLastError());
        gpuErrChk(hipMemsetAsync(m_chainStatus, 0, sizeof(unsigned), 0));
        gpuErrChk(hipStreamSynchronize(0));
        m_condRaw = Lh;
        m_condRawSamples = numSamples;
    }
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
    // (pytorch/utils.py:62-70, inference.py:58-60).  pcmOut: caller-owned [maxBatch][maxSamples]
    // int16, host or device; filled by run / run_partial / run_chunks wherever yOut is; NULL disables.
    void setAudioOut(short* pcmOut) {
        m_pcmUser = pcmOut;
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
        const bool dump = F16 ? dumpActivations : true;
        if (isPipe()) {
            snprintf(buf, n, "wn::wavenet_pipe<%s,%d,%d,%d,DUMP=%d> stages=%d layers/stage=%d chains=%d groups=%d tiles/group=%d wgs=%d lds=%zu",
                     F16 ? "fp16" : "fp32", R, S, A, dump ? 1 : 0, m_chainStages, m_chainLpc, m_pipeChains, m_pipeGroups, PC::G,
                     m_chainStages * m_pipeChains, PC::ldsBytes());
            return;
        }
        if (isChain()) {
            snprintf(buf, n, "wn::wavenet_chain<%s,%d,%d,%d,DUMP=%d> stages=%d layers/stage=%d chains=%d wgs=%d lds=%zu",
                     F16 ? "fp16" : "fp32", R, S, A, dump ? 1 : 0, m_chainStages, m_chainLpc, tiles, m_chainStages * tiles,
                     CC::ldsBytes());
            return;
        }
        if (m_streamMode) {
            snprintf(buf, n, "wn::wavenet_stream<%s,%d,%d,%d,DUMP=%d> tiles/wg=4 wgs=%d lds=%zu", F16 ? "fp16" : "fp32", R,
                     S, A, dump ? 1 : 0, (tiles + 3) / 4, SC::ldsBytes(m_numLayers, m_streamNS));
            return;
        }
        const int bt = wgTiles(tiles);
        int nEmb = bt == 2 ? embTables<2>() : embTables<1>();
        size_t lds = bt == 2 ? ldsNeed<2>(m_numLayers, nEmb) : ldsNeed<1>(m_numLayers, nEmb);
        if constexpr (WG3) {
            if (bt == 3) {
                nEmb = embTables<3>();
                lds = ldsNeed<3>(m_numLayers, nEmb);
            }
        }
        snprintf(buf, n, "wn::wavenet_wg<%s,%d,%d,%d,BT=%d,EMBLDS=%d,DUMP=%d> tiles/wg=%d wgs=%d lds=%zu",
                 F16 ? "fp16" : "fp32", R, S, A, bt, nEmb, dump ? 1 : 0, bt, (tiles + bt - 1) / bt, lds);
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
            gpuErrChk(hipEventCreateWithFlags(&pc.generated, hipEventDisableTiming));
            gpuErrChk(hipEventCreateWithFlags(&pc.delivered, hipEventDisableTiming));
            pieces.push_back(pc);
        }
        hipStream_t genStream = stream, outStream;
        if (!genStream) gpuErrChk(hipStreamCreate(&genStream));
        gpuErrChk(hipStreamCreate(&outStream));

        bool ok = true;
        for (size_t k = 0; k < pieces.size(); k++) {
            const Piece& pc = pieces[k];
            m_num_samples_per_chunk = pc.count;
            // The reference dumps activations in every chunk (nv_wavenet.cuh:471, hard-coded true) and
            // its test reads them back afterwards; only the last chunk's dump can be observed, so only
            // the last chunk runs the dump-capable kernel variant.
            ok = run_partial(pc.first, num_samples, batch_size, NULL, batch_size_per_block, k + 1 == pieces.size(), genStream) && ok;
            gpuErrChk(hipEventRecord(pc.generated, genStream));
            gpuErrChk(hipStreamWaitEvent(outStream, pc.generated, 0));
            if (yOut) getYOut(yOut, pc.first, pc.count, outStream);
            if (m_pcmUser) getAudioOut(m_pcmUser, pc.first, pc.count, outStream);
            gpuErrChk(hipEventRecord(pc.delivered, outStream));
        }
        m_num_samples_per_chunk = 0;
        for (size_t k = 0; k < pieces.size(); k++) {
            gpuErrChk(hipEventSynchronize(pieces[k].delivered));
            consume(yOut, pieces[k].first, pieces[k].count);
        }
        for (size_t k = 0; k < pieces.size(); k++) {
            gpuErrChk(hipEventDestroy(pieces[k].generated));
            gpuErrChk(hipEventDestroy(pieces[k].delivered));
        }
        if (!stream) gpuErrChk(hipStreamDestroy(genStream));
        gpuErrChk(hipStreamDestroy(outStream));
        if ((isChain() || isPipe()) && chainStatus() != 0) ok = false;   // (everything has completed: the check costs nothing)
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
        if (m_implementation == SINGLE_BLOCK) assert(S <= 4 * R);
        if (!m_supported) return false;

        wn::Params p;
        p.wblob = m_wblob;
        p.bias = m_bias;
        p.embPrev = m_embedPrev;
        p.embCur = m_embedCur;
        p.cond = m_cond;
        p.condRaw = m_condRaw;
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
        p.condSamples = m_condRaw ? m_condRawSamples : m_maxSamples;
        p.initSample = init_sample;
        p.count = m_num_samples_per_chunk ? m_num_samples_per_chunk : num_samples;
        if (p.initSample + p.count > num_samples) p.count = num_samples - p.initSample;
        p.ringSlots = m_ringSlots;
        p.tiles = m_tiles;
        p.tanhEmbed = m_tanhEmbed ? 1 : 0;
        p.dump = dumpActivations ? 1 : 0;
        p.embLds = 0;
        p.useRng = m_useRng ? 1 : 0;
        p.rngKey0 = (unsigned)m_rngSeed;
        p.rngKey1 = (unsigned)(m_rngSeed >> 32);
        // rings + conditioning of many tiles stream through HBM: keep them from evicting the weights
        p.ntStream = ((size_t)((batch_size + 15) / 16) * m_ringSlots * R * 16 * sizeof(elem) > ((size_t)16 << 20)) ? 1 : 0;
        m_lastStride = num_samples;
        if (p.count <= 0) return true;

        const int tiles = (batch_size + 15) / 16;
        bool result;
        if (isPipe()) {
            result = launchPipe(p, tiles, stream);
        } else if (isChain()) {
            result = launchChain(p, tiles, stream);
        } else if (m_streamMode) {
            bool noDump = false;
            if constexpr (F16) noDump = !p.dump;
            if (noDump) {
                if constexpr (F16)
                    hipLaunchKernelGGL((wn::wavenet_stream<F16, R, S, A, false>), dim3((tiles + 3) / 4), dim3(512),
                                       SC::ldsBytes(m_numLayers, m_streamNS), stream, p, m_streamNS);
            } else {
                hipLaunchKernelGGL((wn::wavenet_stream<F16, R, S, A, true>), dim3((tiles + 3) / 4), dim3(512),
                                   SC::ldsBytes(m_numLayers, m_streamNS), stream, p, m_streamNS);
            }
            result = hipGetLastError() == hipSuccess;
        } else if (wgTiles(tiles) == 3) {
            result = false;
            if constexpr (WG3) result = launch<3>(p, tiles, stream);
        } else if (wgTiles(tiles) == 2) result = launch<2>(p, tiles, stream);
        else result = launch<1>(p, tiles, stream);
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
    // the sample history back to 128 (what setInputs does), asynchronously on `stream`
    void resetHistory(hipStream_t stream = 0) {
        hipLaunchKernelGGL(wn::silence_kernel, dim3(1), dim3(256), 0, stream, m_yInPrev, m_yInCur, m_maxBatch);
        gpuErrChk(hipGetLastError());
    }

    bool run(int num_samples, int batch_size, int* yOut = NULL, int batch_size_per_block = 1,
             bool dumpActivations = false, hipStream_t stream = 0) {
        m_num_samples_per_chunk = 0;
        return run_partial(0, num_samples, batch_size, yOut, batch_size_per_block, dumpActivations, stream);
    }

protected:
    // the chain kept full: one launch, every (chain, stage) workgroup resident at the same time
    bool launchPipe(wn::Params& p, int tiles, hipStream_t stream) {
        wn::PipeParams pp;
        pp.mail = m_mail;
        pp.status = m_chainStatus;
        pp.stages = m_chainStages;
        pp.lpc = m_chainLpc;
        pp.chains = m_pipeChains;
        pp.groups = m_pipeGroups;
        pp.tiles = tiles;
        p.embLds = PC::embTables();
        bool dump = true;
        if constexpr (F16) dump = p.dump != 0;
        gpuErrChk(hipMemsetAsync(m_mail, 0, m_mailBytes, stream));
        const int grid = 8 * m_chainStages * ((m_pipeChains + 7) / 8);
        if (dump) {
            hipLaunchKernelGGL((wn::wavenet_pipe<F16, R, S, A, true>), dim3(grid), dim3(C::THREADS), PC::ldsBytes(), stream, p, pp);
        } else {
            if constexpr (F16)
                hipLaunchKernelGGL((wn::wavenet_pipe<F16, R, S, A, false>), dim3(grid), dim3(C::THREADS), PC::ldsBytes(), stream, p, pp);
        }
        return hipGetLastError() == hipSuccess;
    }

    // the multi-CU chain: every (tile, stage) workgroup must be resident at the same time, so tiles are
    // launched in groups of at most CUs / stages chains; mailboxes are re-zeroed before every launch
    bool launchChain(wn::Params& p, int tiles, hipStream_t stream) {
        wn::ChainParams cp;
        cp.mail = m_mail;
        cp.status = m_chainStatus;
        cp.stages = m_chainStages;
        cp.lpc = m_chainLpc;
        p.embLds = CC::embTables();
        const int perLaunch = m_numCUs / m_chainStages;
        if (perLaunch < 1) return false;
        bool dump = true;
        if constexpr (F16) dump = p.dump != 0;
        for (int t0 = 0; t0 < tiles; t0 += perLaunch) {
            cp.tile0 = t0;
            cp.chains = tiles - t0 < perLaunch ? tiles - t0 : perLaunch;
            gpuErrChk(hipMemsetAsync(m_mail, 0, CC::mailGranules(cp.chains, m_chainStages) * sizeof(unsigned long long), stream));
            const int grid = 8 * m_chainStages * ((cp.chains + 7) / 8);
            if (dump) {
                hipLaunchKernelGGL((wn::wavenet_chain<F16, R, S, A, true>), dim3(grid), dim3(C::THREADS), CC::ldsBytes(), stream, p, cp);
            } else {
                if constexpr (F16)
                    hipLaunchKernelGGL((wn::wavenet_chain<F16, R, S, A, false>), dim3(grid), dim3(C::THREADS), CC::ldsBytes(), stream, p, cp);
            }
            if (hipGetLastError() != hipSuccess) return false;
        }
        return true;
    }
};
