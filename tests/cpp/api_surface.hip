// Compile-only translation unit (tests/test_capi_cpu.py): a reference-style host program written against
// nv_wavenet.hpp exactly as /root/reference/nv_wavenet_test.cu:44-329 and pytorch/wavenet_infer.cu:40-100 are
// written against nv_wavenet.cuh -- every public member of nvWavenetInfer with the reference's defaults
// (nv_wavenet.cuh:311,396-444,636-639), a lambda run_chunks consumer, both precisions.
#include <vector>

#include "nv_wavenet.hpp"

template <typename T_weight, typename T_data, int R, int S, int A>
static bool drive(int num_layers, int batch_size, int num_samples, int impl) {
    typedef nvWavenetInfer<T_weight, T_data, R, S, A> Infer;
    Infer* infer = new Infer(num_layers, /*maxDilation*/ 8, batch_size, num_samples, impl);   // tanhEmbed defaults to true
    Infer defaults(num_layers, 8, batch_size, num_samples);                                    // impl defaults to AUTO
    (void)defaults;
    std::vector<float> embP(A * R), embC(A * R);
    infer->setEmbeddings(embP.data(), embC.data());
    std::vector<float> Wprev(2 * R * R), Wcur(2 * R * R), Bh(2 * R), Wres(R * R), Bres(R), Wskip(S * R), Bskip(S);
    for (int l = 0; l < num_layers; l++)
        infer->setLayerWeights(l, Wprev.data(), Wcur.data(), Bh.data(), Wres.data(), Bres.data(), Wskip.data(), Bskip.data());
    std::vector<float> Wzs(A * S), Bzs(A), Wza(A * A), Bza(A);
    infer->setOutWeights(Wzs.data(), Bzs.data(), Wza.data(), Bza.data());
    std::vector<float> Lh((size_t)num_samples * num_layers * batch_size * 2 * R), sel((size_t)num_samples * batch_size);
    infer->setInputs(Lh.data(), sel.data());

    std::vector<int> yOut((size_t)batch_size * num_samples);
    bool ok = infer->run(num_samples, batch_size);                                    // every default
    ok = infer->run(num_samples, batch_size, yOut.data(), 1, true, (hipStream_t)0) && ok;
    ok = infer->run_partial(0, num_samples, batch_size, yOut.data(), 1, false, (hipStream_t)0) && ok;
    int consumed = 0;
    ok = infer->run_chunks(7, [&consumed](int* y, int first, int count) { consumed += count; (void)y; (void)first; },
                           num_samples, batch_size, yOut.data(), 1, true) && ok;
    ok = infer->run_chunks(7, [](int*, int, int) {}, num_samples, batch_size) && ok;   // yOut defaults to NULL

    std::vector<float> xt((size_t)batch_size * R), sk((size_t)batch_size * S), za((size_t)batch_size * A);
    infer->getXtOut(0, xt.data());
    infer->getSkipOut(0, sk.data());
    infer->getZs(za.data());
    infer->getZa(za.data());
    infer->getP(za.data());
    infer->getYOut(yOut.data(), 0, num_samples);                                       // stream defaults to 0
    infer->getYOut(yOut.data(), 0, num_samples, (hipStream_t)0);
    gpuErrChk(hipDeviceSynchronize());
    static_assert(Infer::AUTO == 0 && Infer::SINGLE_BLOCK == 1 && Infer::DUAL_BLOCK == 2 && Infer::PERSISTENT == 3 &&
                      Infer::MANYBLOCK_NONPERSISTENT == 4,
                  "Implementation enum values of nv_wavenet.cuh:223-229");
    delete infer;
    return ok && consumed == num_samples;
}

// The physical order of a wave's weight stream (wn::Cfg::streamPos: wavenet_wg's consumption order, shared by the
// chain and the pipe through the same function) must be a permutation of the L * FLW layer fragments, start with the
// current tap of layer 0 and end with the skip matrix of the last layer -- checked at compile time for a few depths.
template <typename C> constexpr bool stream_layout_ok(int L) {
    const int n = L * C::FLW;
    for (int pos = 0; pos < n; pos++) {
        int hits = 0;
        for (int l = 0; l < L; l++)
            for (int i = 0; i < C::FLW; i++) hits += C::streamPos(l, i, L) == (size_t)pos;
        if (hits != 1) return false;
    }
    return C::streamPos(0, C::O_CUR, L) == 0 && C::streamPos(L - 1, C::O_SKIP + C::FW_SKIP - 1, L) == (size_t)n - 1 &&
           C::streamPos(0, C::O_PREV, L) == (size_t)(L - 1) * C::FLW + C::FW_GATE + C::FW_RES + C::FW_COND &&
           // consumption order of wavenet_wg inside layer l >= 1's part: cur(l) skip(l-1) cond(l+1) res(l) prev(l+1)
           (L < 3 || (C::streamPos(1, C::O_CUR, L) + C::FW_GATE == C::streamPos(0, C::O_SKIP, L) &&
                      C::streamPos(0, C::O_SKIP, L) + C::FW_SKIP + C::FW_COND == C::streamPos(1, C::O_RES, L) &&
                      C::streamPos(1, C::O_RES, L) + C::FW_RES == C::streamPos(2, C::O_PREV, L) &&
                      (C::FW_COND == 0 || C::streamPos(0, C::O_SKIP, L) + C::FW_SKIP == C::streamPos(2, C::O_COND, L)))) &&
           (C::FW_COND == 0 || (C::streamPos(1, C::O_COND, L) == (size_t)C::FW_GATE && C::streamPos(0, C::O_RES, L) == (size_t)C::FW_GATE + C::FW_COND));
}
static_assert(stream_layout_ok<wn::Cfg<true, 64, 256, 256, 1>>(2) && stream_layout_ok<wn::Cfg<true, 64, 256, 256, 3>>(3) &&
                  stream_layout_ok<wn::Cfg<true, 64, 256, 256, 2>>(20) && stream_layout_ok<wn::Cfg<false, 32, 256, 256, 1>>(6) &&
                  stream_layout_ok<wn::Cfg<true, 128, 256, 256, 1>>(7),
              "Cfg::streamPos is not a permutation of the layer fragments");
// ... and with the conditioning weights in the stream (in-kernel conditioning: KFC k-fragments per gate tile)
static_assert(stream_layout_ok<wn::Cfg<true, 64, 256, 256, 3, wn::feat_kfc<true>()>>(5) && stream_layout_ok<wn::Cfg<true, 64, 256, 256, 1, 3>>(2) &&
                  stream_layout_ok<wn::Cfg<false, 32, 128, 256, 1, wn::feat_kfc<false>()>>(3) && stream_layout_ok<wn::Cfg<true, 128, 256, 256, 1, 3>>(3) &&
                  wn::Cfg<true, 64, 256, 256, 3, 3>::FLW == 24 && wn::feat_kfc<true>() == 3 && wn::feat_kfc<false>() == 5,
              "Cfg<.., KFC>::streamPos is not a permutation of the layer fragments");

int main() {
    bool ok = drive<float, float, 64, 128, 256>(4, 4, 16, 1);      // the default R,S,A of the class template
    ok = drive<half2, half, 64, 128, 256>(4, 4, 16, 3) && ok;
    typedef nvWavenetInfer<float, float> Defaults;                 // template defaults R=64, S=128, A=256
    (void)sizeof(Defaults);
    return ok ? 0 : 1;
}
