// nv_wavenet.hpp -- host engine: nvWavenetInfer<T_weight, T_data, R, S, A> for MI355X (gfx950).
//
// Drop-in for the class of the same name in /root/reference/nv_wavenet.cuh:220-640: same template
// parameters, Implementation enum values, constructor, setEmbeddings / setLayerWeights /
// setOutWeights / setInputs, run / run_partial / run_chunks and debug getters, with the same
// argument meaning, layouts (col-major fp32 weights in, host OR device pointers, data copied),
// defaults and error convention (HIP errors print "GPUassert: ..." and exit, precondition
// violations assert, nv_wavenet_util.cuh:34-40).  Streams are hipStream_t.
//
// What is different underneath (see wn_kernels.hpp): all Implementation values run the
// MFMA engine (one 4-wave workgroup per tile of 16 utterances, M split over the waves);
// batch_size_per_block is validated like the reference (nv_wavenet.cuh:559-561) but the batch
// tile is fixed by the MFMA shape (16 utterances, 1 or 2 tiles per workgroup chosen from the
// batch size), so it is a no-op hint.  Device buffers are laid out for that engine, not for the
// reference's kernels; the getters return the reference's layouts.
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

#include "wn_kernels.hpp"
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
    static constexpr int MAXBT = 4;        // tiles are allocated in groups of 4 (one workgroup of the throughput kernel)
    using elem = typename wn::Prec<F16>::elem;

    Implementation m_implementation;
    int m_numLayers, m_maxBatch, m_maxSamples, m_maxDilation, m_tiles, m_numCUs;
    bool m_streamMode;   // true: wn::wavenet_stream (>= 1 tile per SIMD), false: wn::wavenet_wg (lowest latency)
    int m_forceBt;       // tiles per workgroup of wn::wavenet_wg forced by NVW_MODE=wg1|wg2 (0: by batch size)
    int m_streamNS;      // LDS ring slots of the throughput kernel
    bool m_tanhEmbed;
    int m_num_samples_per_chunk;
    int m_ringSlots;

    elem* m_wblob;      // packed weight fragments: L layers then the head
    float* m_bias;      // fp32 biases
    elem* m_embedPrev;  // [A][R]
    elem* m_embedCur;
    elem* m_cond;       // packed conditioning
    float* m_outputSelectors;
    elem* m_ring;
    int m_dil[wn::kMaxLayers], m_ringOff[wn::kMaxLayers];
    int *m_yInPrev, *m_yInCur, *m_yOut;
    float *m_XtOut, *m_skipOut, *m_Zs, *m_Za, *m_p;

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
    // returns a device pointer holding n floats of src (src itself when already on the device)
    const float* onDevice(const float* src, size_t n) {
        if (isDevicePtr(src)) return src;
        if (n > m_stageElems) {
            if (m_stage) gpuErrChk(hipFree(m_stage));
            gpuErrChk(hipMalloc(&m_stage, n * sizeof(float)));
            m_stageElems = n;
        }
        gpuErrChk(hipMemcpy(m_stage, src, n * sizeof(float), hipMemcpyHostToDevice));
        return m_stage;
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
        gpuErrChk(hipStreamSynchronize(0));
    }
    // same for the shared stream of the throughput kernel (fragment offset inside the one stream)
    void packWeightStream(size_t fragOff, const float* src, int M, int K, int rowperm, int gate = 0) {
        const float* d = onDevice(src, (size_t)M * K);
        hipLaunchKernelGGL((wn::pack_weight_stream_kernel<F16>), dim3(gridFor((size_t)M * K)), dim3(256), 0, 0,
                           m_wblob + fragOff * SC::FRAG_ELEMS, d, M, K, rowperm, gate);
        gpuErrChk(hipGetLastError());
        gpuErrChk(hipStreamSynchronize(0));
    }
    void convertTo(elem* dst, const float* src, size_t n) {
        const float* d = onDevice(src, n);
        hipLaunchKernelGGL((wn::convert_kernel<F16>), dim3(gridFor(n)), dim3(256), 0, 0, dst, d, n);
        gpuErrChk(hipGetLastError());
        gpuErrChk(hipStreamSynchronize(0));
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
    template <int BT, bool EMB, bool DUMP> bool launchK(wn::Params& p, int tiles, int nEmb, hipStream_t stream) {
        using CB = wn::Cfg<F16, R, S, A, BT>;
        const int grid = (tiles + BT - 1) / BT;
        p.embLds = nEmb;
        hipLaunchKernelGGL((wn::wavenet_wg<F16, R, S, A, BT, EMB, DUMP>), dim3(grid), dim3(CB::THREADS),
                           ldsNeed<BT>(m_numLayers, nEmb), stream, p);
        return hipGetLastError() == hipSuccess;
    }
    template <int BT> bool launch(wn::Params& p, int tiles, hipStream_t stream) {
        const int nEmb = embTables<BT>();
        if constexpr (F16) {
            if (!p.dump) return nEmb ? launchK<BT, true, false>(p, tiles, nEmb, stream) : launchK<BT, false, false>(p, tiles, 0, stream);
        }
        return nEmb ? launchK<BT, true, true>(p, tiles, nEmb, stream) : launchK<BT, false, true>(p, tiles, 0, stream);
    }
    template <int BT, bool EMB, bool DUMP> void allowLdsK() {
        const size_t need = ldsNeed<BT>(m_numLayers, EMB ? embTables<BT>() : 0);
        if (need <= kLdsMax)
            gpuErrChk(hipFuncSetAttribute((const void*)wn::wavenet_wg<F16, R, S, A, BT, EMB, DUMP>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    }
    template <int BT> void allowLds() {
        allowLdsK<BT, false, true>();
        allowLdsK<BT, true, true>();
        if constexpr (F16) {
            allowLdsK<BT, false, false>();
            allowLdsK<BT, true, false>();
        }
    }
    float* headBias() { return m_bias + (size_t)m_numLayers * C::BIAS_L; }

public:
    nvWavenetInfer(int numLayers, int maxDilation, int batchSize, int numSamples, int impl = 0,
                   bool tanhEmbed = true)
        : m_implementation((Implementation)impl), m_numLayers(numLayers), m_maxBatch(batchSize),
          m_maxSamples(numSamples), m_maxDilation(maxDilation), m_tanhEmbed(tanhEmbed),
          m_num_samples_per_chunk(0), m_stage(NULL), m_stageElems(0), m_useRng(false), m_rngSeed(0), m_pcm(NULL),
          m_mulaw(NULL), m_pcmUser(NULL) {
        assert(numLayers >= 2 && batchSize > 0 && numSamples > 0 && maxDilation > 0);
        m_tiles = ((batchSize + 15) / 16 + MAXBT - 1) / MAXBT * MAXBT;   // whole workgroups of MAXBT tiles
        {
            int dev = 0;
            hipDeviceProp_t prop;
            gpuErrChk(hipGetDevice(&dev));
            gpuErrChk(hipGetDeviceProperties(&prop, dev));
            m_numCUs = prop.multiProcessorCount;
        }

        // dilation schedule (nv_wavenet.cuh:99,110-111): d doubles per layer, back to 1 past maxDilation
        assert(numLayers <= wn::kMaxLayers);
        int d = 1, slots = 0;
        for (int l = 0; l < wn::kMaxLayers; l++) m_dil[l] = 1, m_ringOff[l] = 0;
        for (int l = 0; l < numLayers; l++) {
            m_dil[l] = d;
            m_ringOff[l] = slots;
            slots += d;
            d <<= 1;
            if (d > maxDilation) d = 1;
        }
        m_ringSlots = slots;

        // kernel organisation: up to two tiles per CU run in the latency kernel (wn_kernels.hpp: one or
        // two tiles split over the 4 SIMDs of a CU); beyond that every SIMD gets its own tile and the
        // weights are streamed once per CU through an LDS ring (wn_stream.hpp).  NVW_MODE=stream|wg
        // overrides (tests, experiments).
        {
            const size_t biasBytes = ((size_t)numLayers * SC::BIAS_L + 2 * A) * sizeof(float);
            long ns = ((long)kLdsMax - (long)biasBytes) / ((long)SC::CH * 1024);
            m_streamNS = (int)(ns > 6 ? 6 : ns);
            const char* mode = getenv("NVW_MODE");
            m_streamMode = (batchSize + 15) / 16 > 2 * m_numCUs;   // up to two tiles per CU: the latency kernel
            m_forceBt = 0;
            if (mode && !strcmp(mode, "stream")) m_streamMode = true;
            if (mode && !strncmp(mode, "wg", 2)) {
                m_streamMode = false;
                m_forceBt = mode[2] == '1' ? 1 : mode[2] == '2' ? 2 : 0;
            }
            if (const char* fb = getenv("NVW_FORCE_BT")) m_forceBt = atoi(fb);   // experiments
            if (m_streamNS < SC::MIN_NS) m_streamMode = false;
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

        hipLaunchKernelGGL(wn::silence_kernel, dim3(1), dim3(256), 0, 0, m_yInPrev, m_yInCur, m_maxBatch);
        gpuErrChk(hipGetLastError());

        if (!ldsFits<1>()) {
            fprintf(stderr, "nvWavenetInfer: R=%d S=%d A=%d with %d layers needs %zu bytes of LDS (> 160 KiB)\n", R, S,
                    A, numLayers, ldsNeed<1>(numLayers, 0));
            exit(1);
        }
        allowLds<1>();
        allowLds<2>();
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
        if (m_stage) gpuErrChk(hipFree(m_stage));
        if (m_pcm) gpuErrChk(hipFree(m_pcm));
        if (m_mulaw) gpuErrChk(hipFree(m_mulaw));
    }

    // ---- model upload: fp32 in, host or device pointers, data is copied ---------------------
    // embedPrev / embedCur: [A][R]   (nv_wavenet.cuh:396-399)
    virtual void setEmbeddings(float* embedPrev, float* embedCur) {
        convertTo(m_embedPrev, embedPrev, (size_t)A * R);
        convertTo(m_embedCur, embedCur, (size_t)A * R);
    }
    // col-major Wprev,Wcur 2RxR; Bh 2R; Wres RxR; Bres R; Wskip SxR; Bskip S (nv_wavenet.cuh:400-409)
    virtual void setLayerWeights(int layer, float* Wprev, float* Wcur, float* Bh, float* Wres, float* Bres,
                                 float* Wskip, float* Bskip) {
        assert(layer >= 0 && layer < m_numLayers);
        const size_t lf = (size_t)layer * C::FLW;
        if (m_streamMode) {
            const size_t sf = (size_t)layer * SC::FLP;
            packWeightStream(sf + SC::O_PREV, Wprev, 2 * R, R, 0, 1);
            packWeightStream(sf + SC::O_CUR, Wcur, 2 * R, R, 0, 1);
            packWeightStream(sf + SC::O_RES, Wres, R, R, 0);
            // the skip GEMM of layer l is consumed one body later (the head body after the last layer)
            const size_t skipAt = (layer + 1 < m_numLayers) ? sf + SC::FLP + SC::O_SKIP
                                                            : (size_t)m_numLayers * SC::FLP + SC::H_SKIP;
            packWeightStream(skipAt, Wskip, S, R, 0);
        } else {
            packWeight(lf + C::O_PREV, Wprev, 2 * R, R, C::RT);
            packWeight(lf + C::O_CUR, Wcur, 2 * R, R, C::RT);
            packWeight(lf + C::O_RES, Wres, R, R, 0);
            packWeight(lf + C::O_SKIP, Wskip, S, R, 0);
        }
        float* b = m_bias + (size_t)layer * C::BIAS_L;
        gpuErrChk(hipMemcpy(b, Bh, 2 * R * sizeof(float), hipMemcpyDefault));
        if constexpr (F16) {   // the fp16 gate works on pre-scaled pre-activations (wn::gate1)
            hipLaunchKernelGGL((wn::scale_gate_bias_kernel<F16>), dim3(1), dim3(256), 0, 0, b, R);
            gpuErrChk(hipGetLastError());
            gpuErrChk(hipStreamSynchronize(0));
        }
        gpuErrChk(hipMemcpy(b + 2 * R, Bres, R * sizeof(float), hipMemcpyDefault));
        gpuErrChk(hipMemcpy(b + 3 * R, Bskip, S * sizeof(float), hipMemcpyDefault));
    }
    // col-major Wzs AxS, Bzs A, Wza AxA, Bza A (nv_wavenet.cuh:410-415)
    virtual void setOutWeights(float* Wzs, float* Bzs, float* Wza, float* Bza) {
        const size_t hf = C::headOffsetFrags(m_numLayers);
        if (m_streamMode) {
            const size_t sh = (size_t)m_numLayers * SC::FLP;
            packWeightStream(sh + SC::H_ZS, Wzs, A, S, 0);
            packWeightStream(sh + SC::H_ZA, Wza, A, A, 1);   // lane-contiguous logit rows
        } else {
            packWeight(hf, Wzs, A, S, 0);
            packWeight(hf + C::FW_ZS, Wza, A, A, 0);
        }
        gpuErrChk(hipMemcpy(headBias(), Bzs, A * sizeof(float), hipMemcpyDefault));
        gpuErrChk(hipMemcpy(headBias() + A, Bza, A * sizeof(float), hipMemcpyDefault));
    }

    // Lh: [maxSamples][L][maxBatch][2R] conditioning, outputSelectors: [maxSamples][maxBatch]
    // uniform draws; resets the sample history to 128 (nv_wavenet.cuh:417-422).
    void setInputs(float* Lh, float* outputSelectors) {
        setConditioning(Lh);
        m_useRng = false;
        gpuErrChk(hipMemcpy(m_outputSelectors, outputSelectors, (size_t)m_maxSamples * m_maxBatch * sizeof(float),
                            hipMemcpyDefault));
    }

    // ---- extensions beyond the reference --------------------------------
    // The conditioning half of setInputs (also resets the history to 128); pair it with
    // setSelectorSeed() and no [N][B] selector matrix is ever built or uploaded.
    void setConditioning(float* Lh) {
        hipLaunchKernelGGL(wn::silence_kernel, dim3(1), dim3(256), 0, 0, m_yInPrev, m_yInCur, m_maxBatch);
        gpuErrChk(hipGetLastError());
        const size_t rows = (size_t)m_maxSamples * m_numLayers;
        const size_t srcPerRow = (size_t)m_maxBatch * 2 * R;
        const size_t dstPerRow = (size_t)m_tiles * 16 * 2 * R;
        const bool dev = isDevicePtr(Lh);
        // host sources go through the staging buffer in chunks of <= 64 Mi floats
        size_t chunkRows = dev ? rows : ((size_t)64 << 20) / srcPerRow;
        if (chunkRows < 1) chunkRows = 1;
        for (size_t r0 = 0; r0 < rows; r0 += chunkRows) {
            const size_t nr = (rows - r0 < chunkRows) ? rows - r0 : chunkRows;
            const float* src = onDevice(Lh + r0 * srcPerRow, nr * srcPerRow);
            if (m_streamMode)
                hipLaunchKernelGGL((wn::pack_cond_stream_kernel<F16>), dim3(gridFor(nr * dstPerRow)), dim3(256), 0, 0,
                                   m_cond + r0 * dstPerRow, src, nr, m_maxBatch, m_tiles, 2 * R);
            else
                hipLaunchKernelGGL((wn::pack_cond_kernel<F16>), dim3(gridFor(nr * dstPerRow)), dim3(256), 0, 0,
                                   m_cond + r0 * dstPerRow, src, nr, m_maxBatch, m_tiles, R, C::NW);
            gpuErrChk(hipGetLastError());
            gpuErrChk(hipStreamSynchronize(0));
        }
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
                table[y] = (short)(int)v;   // truncation; the top bin wraps to -32768 like numpy's cast
            }
            gpuErrChk(hipMalloc(&m_mulaw, A * sizeof(short)));
            gpuErrChk(hipMemcpy(m_mulaw, table.data(), A * sizeof(short), hipMemcpyHostToDevice));
        }
    }
    void getAudioOut(short* pcm, int offset, int size, hipStream_t stream = 0) {
        gpuErrChk(hipMemcpy2DAsync(pcm + offset, m_maxSamples * sizeof(short), m_pcm + offset,
                                   m_maxSamples * sizeof(short), size * sizeof(short), m_maxBatch, hipMemcpyDefault,
                                   stream));
    }

    // Which device code run(num_samples, batch_size, ..., dumpActivations) launches: kernel name with its
    // template arguments, tiles per workgroup, workgroups, dynamic LDS bytes (for benchmarks / logs).
    void kernelInfo(int batch_size, bool dumpActivations, char* buf, int n) const {
        const int tiles = (batch_size + 15) / 16;
        const bool dump = F16 ? dumpActivations : true;
        if (m_streamMode) {
            snprintf(buf, n, "wn::wavenet_stream<%s,%d,%d,%d,DUMP=%d> tiles/wg=4 wgs=%d lds=%zu", F16 ? "fp16" : "fp32", R,
                     S, A, dump ? 1 : 0, (tiles + 3) / 4, SC::ldsBytes(m_numLayers, m_streamNS));
            return;
        }
        const bool two = (m_forceBt ? m_forceBt == 2 : tiles > m_numCUs) && ldsFits<2>();
        const int bt = two ? 2 : 1;
        const int nEmb = two ? embTables<2>() : embTables<1>();
        snprintf(buf, n, "wn::wavenet_wg<%s,%d,%d,%d,BT=%d,EMBLDS=%d,DUMP=%d> tiles/wg=%d wgs=%d lds=%zu",
                 F16 ? "fp16" : "fp32", R, S, A, bt, nEmb, dump ? 1 : 0, bt, (tiles + bt - 1) / bt,
                 two ? ldsNeed<2>(m_numLayers, nEmb) : ldsNeed<1>(m_numLayers, nEmb));
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
    void getYOut(int* yOut, int offset, int size, hipStream_t stream = 0) {
        size_t cpy_pitch = m_maxSamples * sizeof(int);  // spacing between chunk first elements
        size_t cpy_width = size * sizeof(int);          // size of individual chunk
        size_t cpy_height = m_maxBatch;
        gpuErrChk(hipMemcpy2DAsync(yOut + offset, cpy_pitch, m_yOut + offset, cpy_pitch, cpy_width, cpy_height,
                                   hipMemcpyDefault, stream));
    }

    // ---- generation --------------------------------------------------------------------------
    // Streams chunks of num_samples_per_chunk samples; the copy of chunk j overlaps the compute of
    // chunk j+1; consume(yOut, firstSample, count) runs on the host thread per finished chunk
    // (nv_wavenet.cuh:445-497).
    template <class Callback>
    bool run_chunks(int num_samples_per_chunk, Callback consume, int num_samples, int batch_size, int* yOut = NULL,
                    int batch_size_per_block = 1, bool dumpActivations = false, hipStream_t stream = 0) {
        (void)dumpActivations;
        bool result = true;
        hipStream_t stream_compute, stream_copy;
        if (!stream) {
            gpuErrChk(hipStreamCreate(&stream_compute));
        } else {
            stream_compute = stream;
        }
        gpuErrChk(hipStreamCreate(&stream_copy));
        const int num_chunks = (num_samples + num_samples_per_chunk - 1) / num_samples_per_chunk;
        std::vector<hipEvent_t> event_compute(num_chunks), event_copy(num_chunks);
        for (int j = 0; j < num_chunks; j++) {
            gpuErrChk(hipEventCreateWithFlags(&event_compute[j], hipEventDisableTiming));
            gpuErrChk(hipEventCreateWithFlags(&event_copy[j], hipEventDisableTiming));
        }
        for (int j = 0; j < num_chunks; j++) {
            const int initSample = j * num_samples_per_chunk;
            const int n = (j == num_chunks - 1) ? num_samples - initSample : num_samples_per_chunk;
            m_num_samples_per_chunk = n;
            // The reference dumps activations in every chunk (nv_wavenet.cuh:471, hard-coded true) and
            // its test reads them back afterwards; only the last chunk's dump can be observed, so only
            // the last chunk runs the dump-capable kernel variant.
            result = result && run_partial(initSample, num_samples, batch_size, NULL, batch_size_per_block,
                                           j == num_chunks - 1, stream_compute);
            gpuErrChk(hipEventRecord(event_compute[j], stream_compute));
            gpuErrChk(hipStreamWaitEvent(stream_copy, event_compute[j], 0));
            if (yOut != NULL) getYOut(yOut, initSample, n, stream_copy);
            if (m_pcmUser != NULL) getAudioOut(m_pcmUser, initSample, n, stream_copy);
            gpuErrChk(hipEventRecord(event_copy[j], stream_copy));
        }
        for (int j = 0; j < num_chunks; j++) {
            const int initSample = j * num_samples_per_chunk;
            const int n = (j == num_chunks - 1) ? num_samples - initSample : num_samples_per_chunk;
            gpuErrChk(hipEventSynchronize(event_copy[j]));
            consume(yOut, initSample, n);
        }
        m_num_samples_per_chunk = 0;
        for (int j = 0; j < num_chunks; j++) {
            gpuErrChk(hipEventDestroy(event_compute[j]));
            gpuErrChk(hipEventDestroy(event_copy[j]));
        }
        if (stream != stream_compute) gpuErrChk(hipStreamDestroy(stream_compute));
        gpuErrChk(hipStreamDestroy(stream_copy));
        return result;
    }

    // Generates samples [init_sample, init_sample + chunk) continuing from device-resident state
    // (history, dilation ring); chunk = the run_chunks chunk, or num_samples (nv_wavenet.cuh:499-635).
    // Asynchronous on `stream`.  yOut (host or device, [maxBatch][maxSamples] ints) receives the
    // whole sample buffer when non-NULL.
    bool run_partial(int init_sample, int num_samples, int batch_size, int* yOut = NULL, int batch_size_per_block = 1,
                     bool dumpActivations = false, hipStream_t stream = 0) {
        assert(batch_size_per_block > 0 && batch_size_per_block < 5);
        assert(batch_size % batch_size_per_block == 0);
        assert(batch_size > 0 && batch_size <= m_maxBatch);
        assert(num_samples <= m_maxSamples);
        if (m_implementation == SINGLE_BLOCK) assert(S <= 4 * R);

        wn::Params p;
        p.wblob = m_wblob;
        p.bias = m_bias;
        p.embPrev = m_embedPrev;
        p.embCur = m_embedCur;
        p.cond = m_cond;
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
        p.condSamples = m_maxSamples;
        p.initSample = init_sample;
        p.count = m_num_samples_per_chunk ? m_num_samples_per_chunk : num_samples;
        if (p.initSample + p.count > num_samples) p.count = num_samples - p.initSample;
        p.ringSlots = m_ringSlots;
        p.tiles = m_tiles;
        p.tanhEmbed = m_tanhEmbed ? 1 : 0;
        p.dump = dumpActivations ? 1 : 0;
        // rings + conditioning of many tiles stream through HBM: keep them from evicting the weights
        p.embLds = 0;
        p.useRng = m_useRng ? 1 : 0;
        p.rngKey0 = (unsigned)m_rngSeed;
        p.rngKey1 = (unsigned)(m_rngSeed >> 32);
        p.ntStream = ((size_t)((batch_size + 15) / 16) * m_ringSlots * R * 16 * sizeof(elem) > ((size_t)16 << 20)) ? 1 : 0;
        if (p.count <= 0) return true;

        // one tile of 16 utterances per workgroup while the CUs are not all busy (lowest latency);
        // two tiles per workgroup share one pass over the weights beyond that
        const int tiles = (batch_size + 15) / 16;
        bool result;
        const bool two = m_forceBt ? m_forceBt == 2 : tiles > m_numCUs;
        if (m_streamMode) {
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
        } else if (two && ldsFits<2>()) result = launch<2>(p, tiles, stream);
        else result = launch<1>(p, tiles, stream);
        if (m_pcmUser != NULL) {
            // the indices of a finished sample are final: the expansion is a per-element map of yOut
            hipLaunchKernelGGL(wn::mulaw_pcm_kernel, dim3(gridFor((size_t)batch_size * p.count)), dim3(256), 0, stream,
                               m_yOut, m_pcm, m_mulaw, batch_size, num_samples, p.initSample, p.count);
            result = result && hipGetLastError() == hipSuccess;
        }
        if (yOut != NULL) {
            gpuErrChk(hipMemcpyAsync(yOut, m_yOut, (size_t)m_maxSamples * m_maxBatch * sizeof(int), hipMemcpyDefault,
                                     stream));
            if (m_pcmUser != NULL)
                gpuErrChk(hipMemcpyAsync(m_pcmUser, m_pcm, (size_t)m_maxSamples * m_maxBatch * sizeof(short),
                                         hipMemcpyDefault, stream));
        }
        return result;
    }

    bool run(int num_samples, int batch_size, int* yOut = NULL, int batch_size_per_block = 1,
             bool dumpActivations = false, hipStream_t stream = 0) {
        m_num_samples_per_chunk = 0;
        return run_partial(0, num_samples, batch_size, yOut, batch_size_per_block, dumpActivations, stream);
    }
};
