/*
 * ref_driver.cpp -- thin extern "C" door onto the REFERENCE's own CPU implementation.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/wavenet_oracle.c header).  This file contains no
 * WaveNet arithmetic of its own: it is compiled together with the reference's
 * nv_wavenet_reference.cpp and matrix.cpp *where they lie* under /root/reference
 * (oracle/Makefile, target _ref) into oracle/_ref/libnvwavenet_ref.so, which is
 * git-ignored.  It is used to (1) pin oracle/wavenet_oracle.c bit-for-bit, (2) generate
 * the fixtures in tests/golden/ (tests/golden/make_golden.py) and (3) serve as the
 * "reference"-kind CPU baseline in bench.py.
 *
 * Input generation below drives the reference's own Matrix::randomize (matrix.cpp:38-55)
 * in the order nv_wavenet_test.cu:44-111,217-219 consumes rand().
 */
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "matrix.h"
#include "nv_wavenet_reference.h"

extern "C" {

void* nvwref_create(int L, int maxBatch, int maxSamples, int R, int S, int A, int maxDilation) {
    return new nvWavenetReference(L, maxBatch, maxSamples, R, S, A, maxDilation);
}
void nvwref_destroy(void* h) { delete (nvWavenetReference*)h; }
void nvwref_set_embeddings(void* h, float* p, float* c) { ((nvWavenetReference*)h)->setEmbeddings(p, c); }
void nvwref_set_layer_weights(void* h, int layer, float* Wprev, float* Wcur, float* Bh, float* Wres,
                              float* Bres, float* Wskip, float* Bskip) {
    ((nvWavenetReference*)h)->setLayerWeights(layer, Wprev, Wcur, Bh, Wres, Bres, Wskip, Bskip);
}
void nvwref_set_out_weights(void* h, float* Wzs, float* Bzs, float* Wza, float* Bza) {
    ((nvWavenetReference*)h)->setOutWeights(Wzs, Bzs, Wza, Bza);
}
void nvwref_set_inputs(void* h, float* Lh, float* sel) { ((nvWavenetReference*)h)->setInputs(Lh, sel); }
void nvwref_run(void* h, int num_samples, int batch_size, int* yOut) {
    ((nvWavenetReference*)h)->run(num_samples, batch_size, yOut);
}
void nvwref_get_xt_out(void* h, int layer, float* d) { ((nvWavenetReference*)h)->getXtOut(layer, d); }
void nvwref_get_skip_out(void* h, int layer, float* d) { ((nvWavenetReference*)h)->getSkipOut(layer, d); }
void nvwref_get_zs(void* h, float* d) { ((nvWavenetReference*)h)->getZs(d); }
void nvwref_get_za(void* h, float* d) { ((nvWavenetReference*)h)->getZa(d); }
void nvwref_get_p(void* h, float* d) { ((nvWavenetReference*)h)->getP(d); }

void nvwref_srand(unsigned seed) { srand(seed); }

static void fill(float* dst, int rows, int cols, float mean, float scale) {
    Matrix m(rows, cols, false);
    m.randomize(mean, scale);
    if (dst) memcpy(dst, m.data(), sizeof(float) * (size_t)rows * cols);
    free(m.data());  // Matrix has no destructor
}

/* Same signature and output layout as nvw_gen_test_inputs in wavenet_oracle.c. */
void nvwref_gen_test_inputs(int R, int S, int A, int L, int B, int N, float* sel, float* embP,
                            float* embC, float* Wprev, float* Wcur, float* Bh, float* Wres,
                            float* Bres, float* Wskip, float* Bskip, float* Wzs, float* Bzs,
                            float* Wza, float* Bza, float* Lh) {
    float mean = 0.0;
    float scale = 0.5 / R;
    for (int b = 0; b < B; b++) {
        int a = rand() % A;
        int c = rand() % A;
        (void)a; (void)c;
    }
    fill(sel, B, N, 0.5, 1.0);
    fill(embP, R, A, mean, scale);
    fill(embC, R, A, mean, scale);
    for (int l = 0; l < L; l++) {
        fill(Wprev + (size_t)l * 2 * R * R, 2 * R, R, 0.0, 0.5 / (2 * R));
        fill(Wcur + (size_t)l * 2 * R * R, 2 * R, R, 0.0, 0.5 / (2 * R));
        fill(Bh + (size_t)l * 2 * R, 2 * R, 1, 0.0, 0.5 / (2 * R));
        fill(Wres + (size_t)l * R * R, R, R, 0.0, 0.5 / R);
        fill(Bres + (size_t)l * R, R, 1, 0.0, 0.5 / R);
        fill(Wskip + (size_t)l * S * R, S, R, 0.0, 0.5 / S);
        fill(Bskip + (size_t)l * S, S, 1, 0.0, 0.5 / S);
        fill(NULL, S, B, 0.0, 0.5 / S);
    }
    for (int s = 0; s < N; s++)
        for (int l = 0; l < L + 1; l++) fill(NULL, R, B, 0.0, 0.5 / R);
    fill(Wzs, A, S, mean, scale);
    fill(Bzs, A, 1, mean, scale);
    fill(Wza, A, A, mean, scale);
    fill(Bza, A, 1, mean, scale);
    fill(Lh, 2 * R, N * L * B, mean, scale);
}

}  // extern "C"
