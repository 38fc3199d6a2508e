// A reference-style host program written against nv_wavenet.hpp exactly as /root/reference/nv_wavenet_test.cu:44-329 and
// pytorch/wavenet_infer.cu:40-100 are written against nv_wavenet.cuh -- every public member of nvWavenetInfer with the
// reference's defaults (nv_wavenet.cuh:311,396-444,636-639), a lambda run_chunks consumer, both precisions.
//   * compile-only in tests/test_capi_cpu.py (hipcc cross-compiles without a GPU);
//   * built into tests/cpp/api_surface by __graft_entry__.build() and RUN on the GPU by
//     tests/test_parity_gpu.py::test_native_host_program_against_the_c_abi (round 6; role of nv_wavenet_test.cu:331-395 as a native
//     binary): `api_surface` alone drives the whole surface on zero weights and exits 0 when every call returned true;
//     `api_surface run <precision 32|16> <impl> <L> <maxD> <B> <N> <out.bin>` generates from a seeded model (wn_test_uniform below,
//     restated in numpy by the test) through the CLASS -- run, then run_chunks with a lambda consumer, which must agree -- and writes
//     yOut [B][N] int32, which the test compares with the C-ABI run of the same tensors and, in fp32, with the oracle
//     (nv_wavenet_test.cu:302-304 is the bar: identical samples).
#include <stdint.h>
#include <stdio.h>
#include <string.h>

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

// Round 6, LDS budget arithmetic the host plans launches with (nvWavenetInfer::ldsNeed / placeLdsRing), checked at compile time at C3:
// the dump-free bias table holds one row of skip-bias sums; three tiles then take 144 320 B with the current tap's embedding table and
// four tiles 142 592 B without it; the ring slots of the layers with dilation <= D (nv_wavenet.cuh:99,110-111's schedule): two layers of
// dilation 1 at maxDilation 512, 14 slots up to dilation 4, and the whole ring (15 slots) of a 7-layer model with maxDilation 4.
using C3x3 = wn::Cfg<true, 64, 256, 256, 3>;
using C3x4 = wn::Cfg<true, 64, 256, 256, 4>;
static_assert(C3x3::biasFloats(20, false) == 20 * 192 + 256 + 512 && C3x3::biasFloats(20, true) == 20 * 448 + 512, "bias table");
static_assert(C3x3::LDS_FIXED + C3x3::biasFloats(20, false) * 4 + 64 * 256 * 2 == 144320 && C3x3::LDS_FIXED + C3x3::biasFloats(20, true) * 4 + 64 * 256 * 2 == 163776,
              "three tiles per workgroup: LDS of the dump-free / dumping kernel");
static_assert(C3x4::LDS_FIXED + C3x4::biasFloats(20, false) * 4 == 142592 && C3x4::RING_SLOT == 8192 && C3x3::RING_SLOT == 6144, "four tiles per workgroup");
static_assert(C3x3::ldsRingSlots(20, 512, 1) == 2 && C3x3::ldsRingSlots(20, 512, 4) == 14 && C3x3::ldsRingSlots(7, 4, 4) == 15 && C3x3::ldsRingSlots(20, 512, 0) == 0,
              "ring slots of the short dilations");

// uniform [0,1) of (tensor id, element index): a counter-based hash (splitmix64 finaliser), so that numpy restates it elementwise
static inline float wn_test_uniform(uint32_t tensor, uint64_t i) {
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)tensor * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
static std::vector<float> wn_test_tensor(uint32_t id, size_t n, float scale) {      // uniform in [-scale, scale)
    std::vector<float> v(n);
    for (size_t i = 0; i < n; i++) v[i] = (2.0f * wn_test_uniform(id, i) - 1.0f) * scale;
    return v;
}

// tensor ids: 1 embPrev, 2 embCur, 3 Wzs, 4 Bzs, 5 Wza, 6 Bza, 7 Lh, 8 selectors (plain [0,1)), 100 + 7*l + {0..6}: Wprev Wcur Bh Wres Bres Wskip Bskip
template <typename T_weight, typename T_data, int R, int S, int A>
static int generate(int impl, int L, int maxD, int B, int N, const char* out) {
    typedef nvWavenetInfer<T_weight, T_data, R, S, A> Infer;
    Infer infer(L, maxD, B, N, impl);
    const float sR = sqrtf(3.0f / R), sS = sqrtf(3.0f / S), sA = sqrtf(3.0f / A);
    std::vector<float> embP = wn_test_tensor(1, (size_t)A * R, 1.0f), embC = wn_test_tensor(2, (size_t)A * R, 1.0f);
    infer.setEmbeddings(embP.data(), embC.data());
    for (int l = 0; l < L; l++) {
        const uint32_t id = 100 + 7 * l;
        std::vector<float> Wprev = wn_test_tensor(id, (size_t)2 * R * R, sR), Wcur = wn_test_tensor(id + 1, (size_t)2 * R * R, sR),
                           Bh = wn_test_tensor(id + 2, 2 * R, 0.1f), Wres = wn_test_tensor(id + 3, (size_t)R * R, sR),
                           Bres = wn_test_tensor(id + 4, R, 0.1f), Wskip = wn_test_tensor(id + 5, (size_t)S * R, sR),
                           Bskip = wn_test_tensor(id + 6, S, 0.1f);
        infer.setLayerWeights(l, Wprev.data(), Wcur.data(), Bh.data(), Wres.data(), Bres.data(), Wskip.data(), Bskip.data());
        // (the caller may free its buffers as soon as a set* call returns: nv_wavenet_test.cu:143-144)
    }
    std::vector<float> Wzs = wn_test_tensor(3, (size_t)A * S, sS), Bzs = wn_test_tensor(4, A, 0.1f),
                       Wza = wn_test_tensor(5, (size_t)A * A, 4.0f * sA), Bza = wn_test_tensor(6, A, 0.1f);
    infer.setOutWeights(Wzs.data(), Bzs.data(), Wza.data(), Bza.data());
    std::vector<float> Lh = wn_test_tensor(7, (size_t)N * L * B * 2 * R, 0.5f), sel((size_t)N * B);
    for (size_t i = 0; i < sel.size(); i++) sel[i] = wn_test_uniform(8, i);
    infer.setInputs(Lh.data(), sel.data());

    std::vector<int> y((size_t)B * N, -1), y2((size_t)B * N, -2);
    const int bspb = (B % 4) == 0 ? 4 : (B % 2) == 0 ? 2 : 1;                   // as pytorch/wavenet_infer.cu:96 picks it
    bool ok = infer.run(N, B, y.data(), bspb, true);
    gpuErrChk(hipDeviceSynchronize());
    std::vector<float> p((size_t)B * A);
    infer.getP(p.data());                                                        // the dump of the last sample: a distribution
    for (int b = 0; b < B; b++) {
        double sum = 0;
        for (int a = 0; a < A; a++) sum += p[(size_t)b * A + a];
        ok = ok && fabs(sum - 1.0) < 1e-3;
    }
    int consumed = 0;
    infer.setInputs(Lh.data(), sel.data());                                      // a new utterance: history back to silence
    ok = infer.run_chunks((N + 2) / 3, [&consumed](int*, int, int count) { consumed += count; }, N, B, y2.data(), bspb) && ok;
    ok = ok && consumed == N && memcmp(y.data(), y2.data(), y.size() * sizeof(int)) == 0;
    FILE* f = fopen(out, "wb");
    if (!f) return 3;
    fwrite(y.data(), sizeof(int), y.size(), f);
    fclose(f);
    return ok ? 0 : 2;
}

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "run") == 0) {
        if (argc != 9) {
            fprintf(stderr, "usage: %s run <precision 32|16> <impl 0..4> <L> <maxD> <B> <N> <out.bin>   (R=64 S=128 A=256)\n", argv[0]);
            return 64;
        }
        const int prec = atoi(argv[2]), impl = atoi(argv[3]), L = atoi(argv[4]), maxD = atoi(argv[5]), B = atoi(argv[6]), N = atoi(argv[7]);
        return prec == 16 ? generate<half2, half, 64, 128, 256>(impl, L, maxD, B, N, argv[8])
                          : generate<float, float, 64, 128, 256>(impl, L, maxD, B, N, argv[8]);
    }
    bool ok = drive<float, float, 64, 128, 256>(4, 4, 16, 1);      // the default R,S,A of the class template
    ok = drive<half2, half, 64, 128, 256>(4, 4, 16, 3) && ok;
    typedef nvWavenetInfer<float, float> Defaults;                 // template defaults R=64, S=128, A=256
    (void)sizeof(Defaults);
    return ok ? 0 : 1;
}
