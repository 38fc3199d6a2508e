// engine_base.hpp -- type-erased door onto nvWavenetInfer<...> instantiations for the C ABI
// (include/nv_wavenet_c.h).  One translation unit per instantiation (engine_inst.hip) so the
// big kernel templates compile in parallel.
#pragma once
#include "../../include/nv_wavenet_c.h"
#include "nv_wavenet.hpp"

struct nvw_engine {
    virtual ~nvw_engine() {}
    virtual void setEmbeddings(float*, float*) = 0;
    virtual void setLayerWeights(int, float*, float*, float*, float*, float*, float*, float*) = 0;
    virtual void setOutWeights(float*, float*, float*, float*) = 0;
    virtual void setInputs(float*, float*, int) = 0;
    virtual void setConditioning(float*, int) = 0;
    virtual void packConditioning(float*, int, int, hipStream_t) = 0;
    virtual void setConditioningDirect(const void*, int, int) = 0;
    virtual void setConditioningPacked(const void*, int) = 0;
    virtual int condTiles() = 0;
    virtual bool setConditioningWeights(const float*, const float*, int) = 0;
    virtual int featureFragments() = 0;
    virtual size_t featureElems(int) = 0;
    virtual void setConditioningFeatures(const void*, int) = 0;
    virtual void packFeatures(const void*, int, long long, long long, long long, int, int, hipStream_t) = 0;
    virtual void setFeatures(const void*, int, long long, long long, long long, int) = 0;
    virtual int conditioningChannels() = 0;
    virtual bool setUpsampling(const float*, const float*, int, int) = 0;
    virtual int upsamplingStride() = 0;
    virtual int melSamples() = 0;
    virtual bool hasFeatureBuffer() = 0;
    virtual int maxBatch() = 0;
    virtual void getFeatures(void*, int, int) = 0;
    virtual void setMel(const void*, int, long long, long long, long long, int) = 0;
    virtual void upsampleFeatures(int, int, hipStream_t) = 0;
    virtual bool run_stream(int, nvw_consume_fn, void*, int, int, int*, hipStream_t) = 0;
    virtual size_t condPackedElems(int) = 0;
    virtual void setSelectors(float*, int) = 0;
    virtual bool run_range(int, int, int, int, hipStream_t) = 0;
    virtual void resetHistory(hipStream_t) = 0;
    virtual bool supported() = 0;
    virtual unsigned chainStatus() = 0;
    virtual unsigned chainFallbacks() = 0;
    virtual unsigned chainLastTimeout() = 0;
    virtual void setChainTimeoutMs(double) = 0;
    virtual void setClockProbe(bool) = 0;
    virtual void setRingInLds(int) = 0;
    virtual double lastLaunchClockGHz() = 0;
    virtual int precisionBits() = 0;
    virtual int maxSamples() = 0;
    virtual void setSelectorSeed(unsigned long long) = 0;
    virtual void setAudioOut(short*, size_t) = 0;
    virtual void kernelInfo(int, bool, char*, int) = 0;
    virtual bool run(int, int, int*, int, bool, hipStream_t) = 0;
    virtual bool run_partial(int, int, int, int*, int, bool, hipStream_t) = 0;
    virtual bool run_chunks(int, nvw_consume_fn, void*, int, int, int*, int, bool, hipStream_t) = 0;
    virtual void getXtOut(int, float*) = 0;
    virtual void getSkipOut(int, float*) = 0;
    virtual void getZs(float*) = 0;
    virtual void getZa(float*) = 0;
    virtual void getP(float*) = 0;
    virtual void getYOut(int*, int, int, hipStream_t) = 0;
};

template <typename Tw, typename Td, int R, int S, int A>
struct EngineImpl : nvw_engine {
    nvWavenetInfer<Tw, Td, R, S, A> eng;
    int cap;
    EngineImpl(int L, int maxD, int B, int N, int impl, bool tanhEmbed, int org) : eng(L, maxD, B, N, impl, tanhEmbed, org), cap(N) {}
    void setEmbeddings(float* p, float* c) override { eng.setEmbeddings(p, c); }
    void setLayerWeights(int l, float* a, float* b, float* c, float* d, float* e, float* f, float* g) override {
        eng.setLayerWeights(l, a, b, c, d, e, f, g);
    }
    void setOutWeights(float* a, float* b, float* c, float* d) override { eng.setOutWeights(a, b, c, d); }
    void setInputs(float* Lh, float* sel, int n) override { eng.setInputs(Lh, sel, n); }
    void setConditioning(float* Lh, int n) override { eng.setConditioning(Lh, n); }
    void packConditioning(float* Lh, int first, int count, hipStream_t s) override { eng.packConditioning(Lh, first, count, s); }
    void setConditioningDirect(const void* Lh, int n, int prec) override { eng.setConditioningDirect(Lh, n, prec); }
    void setConditioningPacked(const void* frags, int n) override { eng.setConditioningPacked(frags, n); }
    int condTiles() override { return eng.condTiles(); }
    bool setConditioningWeights(const float* W, const float* b, int n) override { return eng.setConditioningWeights(W, b, n); }
    int featureFragments() override { return eng.featureFragments(); }
    size_t featureElems(int n) override { return eng.featureElems(n); }
    void setConditioningFeatures(const void* f, int n) override { eng.setConditioningFeatures(f, n); }
    void packFeatures(const void* x, int prec, long long bS, long long cS, long long tS, int first, int count, hipStream_t s) override {
        eng.packFeatures(x, prec, bS, cS, tS, first, count, s);
    }
    void setFeatures(const void* x, int prec, long long bS, long long cS, long long tS, int n) override { eng.setFeatures(x, prec, bS, cS, tS, n); }
    int conditioningChannels() override { return eng.conditioningChannels(); }
    bool setUpsampling(const float* W, const float* b, int window, int stride) override { return eng.setUpsampling(W, b, window, stride); }
    int upsamplingStride() override { return eng.upsamplingStride(); }
    int melSamples() override { return eng.melSamples(); }
    bool hasFeatureBuffer() override { return eng.hasFeatureBuffer(); }
    int maxBatch() override { return eng.maxBatch(); }
    void getFeatures(void* d, int first, int count) override { eng.getFeatures(d, first, count); }
    void setMel(const void* mel, int prec, long long bS, long long cS, long long fS, int frames) override { eng.setMel(mel, prec, bS, cS, fS, frames); }
    void upsampleFeatures(int first, int count, hipStream_t s) override { eng.upsampleFeatures(first, count, s); }
    bool run_stream(int chunk, nvw_consume_fn fn, void* user, int n, int b, int* y, hipStream_t s) override {
        return eng.run_stream(chunk, [fn, user](int* yo, int i, int c) { if (fn) fn(yo, i, c, user); }, n, b, y, s);
    }
    size_t condPackedElems(int n) override { return eng.condPackedElems(n); }
    void setSelectors(float* sel, int n) override { eng.setSelectors(sel, n); }
    bool run_range(int i, int c, int n, int b, hipStream_t s) override { return eng.run_range(i, c, n, b, s); }
    void resetHistory(hipStream_t s) override { eng.resetHistory(s); }
    bool supported() override { return eng.supported(); }
    unsigned chainStatus() override { return eng.chainStatus(); }
    unsigned chainFallbacks() override { return eng.chainFallbacks(); }
    unsigned chainLastTimeout() override { return eng.chainLastTimeout(); }
    void setChainTimeoutMs(double ms) override { eng.setChainTimeoutMs(ms); }
    void setClockProbe(bool on) override { eng.setClockProbe(on); }
    void setRingInLds(int mode) override { eng.setRingInLds(mode); }
    double lastLaunchClockGHz() override { return eng.lastLaunchClockGHz(); }
    int precisionBits() override { return std::is_same<Td, float>::value ? 32 : 16; }
    int maxSamples() override { return cap; }
    void setSelectorSeed(unsigned long long seed) override { eng.setSelectorSeed(seed); }
    void setAudioOut(short* pcm, size_t n) override { eng.setAudioOut(pcm, n); }
    void kernelInfo(int b, bool dump, char* buf, int n) override { eng.kernelInfo(b, dump, buf, n); }
    bool run(int n, int b, int* y, int bspb, bool dump, hipStream_t s) override {
        return eng.run(n, b, y, bspb, dump, s);
    }
    bool run_partial(int i, int n, int b, int* y, int bspb, bool dump, hipStream_t s) override {
        return eng.run_partial(i, n, b, y, bspb, dump, s);
    }
    bool run_chunks(int chunk, nvw_consume_fn fn, void* user, int n, int b, int* y, int bspb, bool dump,
                    hipStream_t s) override {
        return eng.run_chunks(chunk, [fn, user](int* yo, int i, int c) { if (fn) fn(yo, i, c, user); }, n, b, y,
                              bspb, dump, s);
    }
    void getXtOut(int l, float* d) override { eng.getXtOut(l, d); }
    void getSkipOut(int l, float* d) override { eng.getSkipOut(l, d); }
    void getZs(float* d) override { eng.getZs(d); }
    void getZa(float* d) override { eng.getZa(d); }
    void getP(float* d) override { eng.getP(d); }
    void getYOut(int* y, int off, int size, hipStream_t s) override { eng.getYOut(y, off, size, s); }
};

typedef nvw_engine* (*nvw_factory_fn)(int L, int maxD, int B, int N, int impl, int tanhEmbed, int organisation);

#define WN_CAT2(a, b) a##b
#define WN_CAT(a, b) WN_CAT2(a, b)
#define WN_FACTORY_NAME(R, S, A, P) WN_CAT(WN_CAT(WN_CAT(WN_CAT(WN_CAT(WN_CAT(WN_CAT(nvw_make_, R), _), S), _), A), _p), P)
