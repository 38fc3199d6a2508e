/*
 * wavenet_oracle.c -- CPU ORACLE for the autoregressive WaveNet inference path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nv_wavenet_amd/ (the product) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * It is a plain-C restatement of the reference's own CPU implementation
 * (all citations relative to /root/reference):
 *   - nv_wavenet_reference.cpp:37-121   (embed, layer, final, select)
 *   - nv_wavenet_reference.cpp:232-304  (setInputs: history reset to 128; run loop,
 *                                        dilation schedule, zero history for t<d)
 *   - matrix.cpp:85-183                 (sequential-k fp32 mat-mul, add, bias, relu,
 *                                        softmax with max floored at 0)
 *   - matrix.cpp:38-55                  (Matrix::randomize -- two rand() per element)
 *   - nv_wavenet_test.cu:36-111,217-219 (runTest's input recipe, rand() order)
 * Arithmetic order is kept identical (same fp32 operation sequence, libm tanhf/expf),
 * so that outputs are BIT-IDENTICAL to the reference's nvWavenetReference built from
 * its own sources (oracle/_ref, see oracle/Makefile); tests/test_oracle_cpu.py pins
 * that, plus the committed fixtures in tests/golden/ generated from oracle/_ref.
 *
 * Differences from the reference class, all deliberate and observable only in
 * capacity, not in values:
 *   - activations are kept in a ring of (maxDilation+1) samples instead of one
 *     matrix per sample (nv_wavenet_reference.cpp:136-142), so long runs fit in RAM;
 *   - a failed selection returns an error code instead of assert()
 *     (nv_wavenet_reference.cpp:119);
 *   - optional teacher forcing and CDF-edge reporting (extensions for long-run parity);
 *   - optional tanhEmbed=0 (the GPU engine's ctor flag, nv_wavenet.cuh:311; the
 *     reference oracle always applies tanh, nv_wavenet_reference.cpp:52).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no FMA contraction so the
 * rounding sequence equals the reference's g++ -O2 build).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct nvw_oracle {
    int L, maxBatch, maxSamples, R, S, A, maxDilation;
    int tanhEmbed;
    float *embedPrev, *embedCur;          /* col-major R x A : e[r + a*R]            */
    float **Wprev, **Wcur, **Bh;          /* 2R x R col-major ; 2R                   */
    float **Wres, **Bres, **Wskip, **Bskip;
    float *Wzs, *Bzs, *Wza, *Bza;         /* A x S, A, A x A, A (col-major)          */
    const float *Lh;                      /* caller-owned [N][L][maxBatch][2R]       */
    float *LhCopy;                        /* or owned copy                           */
    float *sel;                           /* [N][maxBatch]                           */
    int *yInPrev, *yInCur;
    int ringSlots;                        /* maxDilation+1                           */
    float *Xt;                            /* [ringSlots][L+1][R x maxBatch col-major]*/
    float *skipOut;                       /* [L][S x maxBatch]                       */
    float *Zs, *Za, *P;                   /* A x maxBatch                            */
    int lastSlot;
    /* scratch */
    float *a_prev, *a_cur, *hprime, *h, *zeroR, *zeroS;
} nvw_oracle;

/* ---- matrix.cpp restatements (col-major, index = row + col*rows) ---------------- */

/* matrix.cpp:85-103 : C = A(MxK) * B(KxN), sequential k, fp32 */
static void mat_mul(float* C, const float* A, const float* B, int M, int K, int N) {
    for (int row = 0; row < M; ++row) {
        for (int col = 0; col < N; ++col) {
            float sum = 0;
            for (int inner = 0; inner < K; ++inner) {
                sum += A[row + inner * M] * B[inner + col * K];
            }
            C[row + col * M] = sum;
        }
    }
}
/* matrix.cpp:105-116 */
static void mat_add(float* C, const float* A, const float* B, int M, int N) {
    for (int i = 0; i < M * N; ++i) C[i] = A[i] + B[i];
}
/* matrix.cpp:118-130 */
static void mat_bias(float* C, const float* A, const float* b, int M, int N) {
    for (int col = 0; col < N; ++col)
        for (int row = 0; row < M; ++row) C[row + col * M] = A[row + col * M] + b[row];
}
/* matrix.cpp:153-164 */
static void mat_relu(float* D, const float* Sx, int M, int N) {
    for (int i = 0; i < M * N; ++i) {
        float v = Sx[i];
        D[i] = (v < 0) ? 0.f : v;
    }
}
/* matrix.cpp:166-183 : note max starts at 0.f, not -inf */
static void mat_softmax(float* D, const float* Sx, int M, int N) {
    for (int col = 0; col < N; ++col) {
        float max = 0.f;
        for (int row = 0; row < M; ++row)
            if (Sx[row + col * M] > max) max = Sx[row + col * M];
        float sum = 0.f;
        for (int row = 0; row < M; ++row) sum += expf(Sx[row + col * M] - max);
        for (int row = 0; row < M; ++row) D[row + col * M] = expf(Sx[row + col * M] - max) / sum;
    }
}

/* nv_wavenet_reference.cpp:37 */
static float sigmoidf_ref(float f) { return 1.f / (1.f + expf(-f)); }

/* ---- lifetime ------------------------------------------------------------------- */

static float* fzalloc(size_t n) { return (float*)calloc(n ? n : 1, sizeof(float)); }

nvw_oracle* nvw_oracle_create(int L, int maxBatch, int maxSamples, int R, int S, int A,
                              int maxDilation) {
    nvw_oracle* o = (nvw_oracle*)calloc(1, sizeof(nvw_oracle));
    o->L = L; o->maxBatch = maxBatch; o->maxSamples = maxSamples;
    o->R = R; o->S = S; o->A = A; o->maxDilation = maxDilation;
    o->tanhEmbed = 1;
    o->embedPrev = fzalloc((size_t)R * A);
    o->embedCur = fzalloc((size_t)R * A);
    o->Wprev = (float**)calloc(L, sizeof(float*)); o->Wcur = (float**)calloc(L, sizeof(float*));
    o->Bh = (float**)calloc(L, sizeof(float*));    o->Wres = (float**)calloc(L, sizeof(float*));
    o->Bres = (float**)calloc(L, sizeof(float*));  o->Wskip = (float**)calloc(L, sizeof(float*));
    o->Bskip = (float**)calloc(L, sizeof(float*));
    for (int l = 0; l < L; l++) {
        o->Wprev[l] = fzalloc((size_t)2 * R * R); o->Wcur[l] = fzalloc((size_t)2 * R * R);
        o->Bh[l] = fzalloc(2 * R);                o->Wres[l] = fzalloc((size_t)R * R);
        o->Bres[l] = fzalloc(R);                  o->Wskip[l] = fzalloc((size_t)S * R);
        o->Bskip[l] = fzalloc(S);
    }
    o->Wzs = fzalloc((size_t)A * S); o->Bzs = fzalloc(A);
    o->Wza = fzalloc((size_t)A * A); o->Bza = fzalloc(A);
    o->sel = fzalloc((size_t)maxSamples * maxBatch);
    o->yInPrev = (int*)calloc(maxBatch, sizeof(int));
    o->yInCur = (int*)calloc(maxBatch, sizeof(int));
    o->ringSlots = maxDilation + 1;
    o->Xt = fzalloc((size_t)o->ringSlots * (L + 1) * R * maxBatch);
    o->skipOut = fzalloc((size_t)L * S * maxBatch);
    o->Zs = fzalloc((size_t)A * maxBatch); o->Za = fzalloc((size_t)A * maxBatch);
    o->P = fzalloc((size_t)A * maxBatch);
    o->a_prev = fzalloc((size_t)2 * R * maxBatch); o->a_cur = fzalloc((size_t)2 * R * maxBatch);
    o->hprime = fzalloc((size_t)2 * R * maxBatch); o->h = fzalloc((size_t)R * maxBatch);
    o->zeroR = fzalloc((size_t)R * maxBatch);      o->zeroS = fzalloc((size_t)S * maxBatch);
    return o;
}

void nvw_oracle_destroy(nvw_oracle* o) {
    if (!o) return;
    free(o->embedPrev); free(o->embedCur);
    for (int l = 0; l < o->L; l++) {
        free(o->Wprev[l]); free(o->Wcur[l]); free(o->Bh[l]); free(o->Wres[l]);
        free(o->Bres[l]); free(o->Wskip[l]); free(o->Bskip[l]);
    }
    free(o->Wprev); free(o->Wcur); free(o->Bh); free(o->Wres); free(o->Bres);
    free(o->Wskip); free(o->Bskip);
    free(o->Wzs); free(o->Bzs); free(o->Wza); free(o->Bza);
    free(o->LhCopy); free(o->sel); free(o->yInPrev); free(o->yInCur);
    free(o->Xt); free(o->skipOut); free(o->Zs); free(o->Za); free(o->P);
    free(o->a_prev); free(o->a_cur); free(o->hprime); free(o->h); free(o->zeroR); free(o->zeroS);
    free(o);
}

/* ---- model / input upload (nv_wavenet_reference.cpp:201-245) -------------------- */

void nvw_oracle_set_tanh_embed(nvw_oracle* o, int tanhEmbed) { o->tanhEmbed = tanhEmbed; }

void nvw_oracle_set_embeddings(nvw_oracle* o, const float* embedPrev, const float* embedCur) {
    memcpy(o->embedPrev, embedPrev, sizeof(float) * o->R * o->A);
    memcpy(o->embedCur, embedCur, sizeof(float) * o->R * o->A);
}

void nvw_oracle_set_layer_weights(nvw_oracle* o, int layer, const float* Wprev, const float* Wcur,
                                  const float* Bh, const float* Wres, const float* Bres,
                                  const float* Wskip, const float* Bskip) {
    int R = o->R, S = o->S;
    memcpy(o->Wprev[layer], Wprev, sizeof(float) * 2 * R * R);
    memcpy(o->Wcur[layer], Wcur, sizeof(float) * 2 * R * R);
    memcpy(o->Bh[layer], Bh, sizeof(float) * 2 * R);
    memcpy(o->Wres[layer], Wres, sizeof(float) * R * R);
    memcpy(o->Bres[layer], Bres, sizeof(float) * R);
    memcpy(o->Wskip[layer], Wskip, sizeof(float) * S * R);
    memcpy(o->Bskip[layer], Bskip, sizeof(float) * S);
}

void nvw_oracle_set_out_weights(nvw_oracle* o, const float* Wzs, const float* Bzs, const float* Wza,
                                const float* Bza) {
    memcpy(o->Wzs, Wzs, sizeof(float) * o->S * o->A);
    memcpy(o->Bzs, Bzs, sizeof(float) * o->A);
    memcpy(o->Wza, Wza, sizeof(float) * o->A * o->A);
    memcpy(o->Bza, Bza, sizeof(float) * o->A);
}

/* nv_wavenet_reference.cpp:232-245 : Lh is [N][L][maxBatch][2R], selectors [N][maxBatch];
 * history is reset to 128 (mu-law silence).  copy=0 keeps the caller's Lh pointer. */
void nvw_oracle_set_inputs(nvw_oracle* o, const float* Lh, const float* sel, int copy) {
    for (int i = 0; i < o->maxBatch; i++) { o->yInPrev[i] = 128; o->yInCur[i] = 128; }
    size_t n = (size_t)o->maxSamples * o->L * o->maxBatch * 2 * o->R;
    free(o->LhCopy); o->LhCopy = NULL;
    if (copy) {
        o->LhCopy = (float*)malloc(n * sizeof(float));
        memcpy(o->LhCopy, Lh, n * sizeof(float));
        o->Lh = o->LhCopy;
    } else {
        o->Lh = Lh;
    }
    memcpy(o->sel, sel, sizeof(float) * (size_t)o->maxSamples * o->maxBatch);
}

/* ---- getters (nv_wavenet_reference.cpp:247-265) --------------------------------- */

void nvw_oracle_get_xt_out(nvw_oracle* o, int layer, float* dst) {
    memcpy(dst, o->Xt + ((size_t)o->lastSlot * (o->L + 1) + layer + 1) * o->R * o->maxBatch,
           sizeof(float) * o->R * o->maxBatch);
}
void nvw_oracle_get_skip_out(nvw_oracle* o, int layer, float* dst) {
    memcpy(dst, o->skipOut + (size_t)layer * o->S * o->maxBatch, sizeof(float) * o->S * o->maxBatch);
}
void nvw_oracle_get_zs(nvw_oracle* o, float* dst) { memcpy(dst, o->Zs, sizeof(float) * o->A * o->maxBatch); }
void nvw_oracle_get_za(nvw_oracle* o, float* dst) { memcpy(dst, o->Za, sizeof(float) * o->A * o->maxBatch); }
void nvw_oracle_get_p(nvw_oracle* o, float* dst) { memcpy(dst, o->P, sizeof(float) * o->A * o->maxBatch); }
void nvw_oracle_get_history(nvw_oracle* o, int* yPrev, int* yCur) {
    memcpy(yPrev, o->yInPrev, sizeof(int) * o->maxBatch);
    memcpy(yCur, o->yInCur, sizeof(int) * o->maxBatch);
}

/* ---- one timestep --------------------------------------------------------------- */

/* nv_wavenet_reference.cpp:42-56 */
static void embed(nvw_oracle* o, int B, float* x0) {
    int R = o->R;
    for (int b = 0; b < B; b++) {
        int prev = o->yInPrev[b], cur = o->yInCur[b];
        for (int r = 0; r < R; r++) {
            float e = o->embedPrev[r + (size_t)prev * R] + o->embedCur[r + (size_t)cur * R];
            x0[r + b * R] = o->tanhEmbed ? tanhf(e) : e;
        }
    }
}

/* nv_wavenet_reference.cpp:58-92 */
static void layer(nvw_oracle* o, int l, int B, const float* Lh_l, const float* Xtmd,
                  const float* Xin, float* Xout, const float* skipIn, float* skipOut, int last) {
    int R = o->R, S = o->S;
    mat_mul(o->a_prev, o->Wprev[l], Xtmd, 2 * R, R, B);
    mat_mul(o->a_cur, o->Wcur[l], Xin, 2 * R, R, B);
    mat_add(o->hprime, o->a_prev, o->a_cur, 2 * R, B);
    mat_bias(o->hprime, o->hprime, o->Bh[l], 2 * R, B);
    mat_add(o->hprime, o->hprime, Lh_l, 2 * R, B);
    for (int b = 0; b < B; b++)
        for (int row = 0; row < R; row++)
            o->h[row + b * R] = tanhf(o->hprime[row + b * 2 * R]) * sigmoidf_ref(o->hprime[row + R + b * 2 * R]);
    mat_mul(Xout, o->Wres[l], o->h, R, R, B);
    mat_bias(Xout, Xout, o->Bres[l], R, B);
    mat_add(Xout, Xout, Xin, R, B);
    mat_mul(skipOut, o->Wskip[l], o->h, S, R, B);
    mat_add(skipOut, skipOut, skipIn, S, B);
    mat_bias(skipOut, skipOut, o->Bskip[l], S, B);
    if (last) mat_relu(skipOut, skipOut, S, B);
}

/* nv_wavenet_reference.cpp:94-104 */
static void final_stage(nvw_oracle* o, int B, const float* skip) {
    int S = o->S, A = o->A;
    mat_mul(o->Zs, o->Wzs, skip, A, S, B);
    mat_bias(o->Zs, o->Zs, o->Bzs, A, B);
    mat_relu(o->Zs, o->Zs, A, B);
    mat_mul(o->Za, o->Wza, o->Zs, A, A, B);
    mat_bias(o->Za, o->Za, o->Bza, A, B);
    mat_softmax(o->P, o->Za, A, B);
}

/* nv_wavenet_reference.cpp:106-121 ; returns -1 where the scan falls off the end.
 * lo/hi (optional) receive the cumulative probability just before / after the pick. */
static int select_one(const float* p, int A, float sel, float* lo, float* hi) {
    float sum = 0.f;
    for (int row = 0; row < A; row++) {
        float before = sum;
        sum += p[row];
        if (sel < sum) {
            if (lo) *lo = before;
            if (hi) *hi = sum;
            return row;
        }
    }
    if (lo) *lo = sum;
    if (hi) *hi = sum;
    return -1;
}

/*
 * nv_wavenet_reference.cpp:269-304.  yOut is [B][num_samples].
 * yForced (optional, [B][num_samples]) teacher-forces the fed-back history;
 * cdfLo/cdfHi (optional, [B][num_samples]) report the CDF edges around each pick.
 * Returns 0, or -(1+sample) if a selection failed (the reference asserts there).
 * The batch stride of Lh / selectors / activations is maxBatch, as in the reference.
 */
int nvw_oracle_run_ex(nvw_oracle* o, int num_samples, int batch_size, int* yOut, const int* yForced,
                      float* cdfLo, float* cdfHi) {
    int L = o->L, R = o->R, S = o->S, A = o->A, MB = o->maxBatch;
    (void)S;
    size_t plane = (size_t)R * MB;
    int rc = 0;
    for (int sample = 0; sample < num_samples; sample++) {
        int slot = sample % o->ringSlots;
        float* Xs = o->Xt + (size_t)slot * (L + 1) * plane;
        embed(o, batch_size, Xs);
        int dilation = 1;
        for (int l = 0; l < L; l++) {
            const float* Xtmd = o->zeroR;
            if (sample >= dilation) {
                int pslot = (sample - dilation) % o->ringSlots;
                Xtmd = o->Xt + ((size_t)pslot * (L + 1) + l) * plane;
            }
            dilation *= 2;
            if (dilation > o->maxDilation) dilation = 1;
            const float* skipIn = (l == 0) ? o->zeroS : o->skipOut + (size_t)(l - 1) * o->S * MB;
            const float* Lh_l = o->Lh + ((size_t)sample * L + l) * MB * 2 * R;
            layer(o, l, batch_size, Lh_l, Xtmd, Xs + l * plane, Xs + (l + 1) * plane, skipIn,
                  o->skipOut + (size_t)l * o->S * MB, l == L - 1);
        }
        final_stage(o, batch_size, o->skipOut + (size_t)(L - 1) * o->S * MB);
        for (int b = 0; b < batch_size; b++) {
            float lo, hi;
            int y = select_one(o->P + (size_t)b * A, A, o->sel[(size_t)sample * MB + b], &lo, &hi);
            if (y < 0 && rc == 0) rc = -(1 + sample);
            if (cdfLo) cdfLo[(size_t)b * num_samples + sample] = lo;
            if (cdfHi) cdfHi[(size_t)b * num_samples + sample] = hi;
            yOut[(size_t)b * num_samples + sample] = y;
            int fed = yForced ? yForced[(size_t)b * num_samples + sample] : y;
            if (fed < 0) fed = 128; /* keep indices legal after a failed pick */
            o->yInPrev[b] = o->yInCur[b];
            o->yInCur[b] = fed;
        }
        o->lastSlot = slot;
    }
    return rc;
}

int nvw_oracle_run(nvw_oracle* o, int num_samples, int batch_size, int* yOut) {
    return nvw_oracle_run_ex(o, num_samples, batch_size, yOut, NULL, NULL, NULL);
}

/* ---- the reference test's input recipe ------------------------------------------ */

/* matrix.cpp:38-55 : col-major rows x cols; TWO rand() calls per element. */
void nvw_randomize(float* data, int rows, int cols, float mean, float scale) {
    for (int row = 0; row < rows; row++) {
        for (int col = 0; col < cols; col++) {
            if ((rand() % 100) < 0) {
                data[row + (size_t)col * rows] = 0.f;
            } else {
                float r = (float)rand() / (float)RAND_MAX;
                r -= 0.5;
                r = r * scale + mean;
                data[row + (size_t)col * rows] = r;
            }
        }
    }
}

void nvw_srand(unsigned seed) { srand(seed); }

/*
 * nv_wavenet_test.cu:44-111,217-219 : generate one runTest() worth of inputs, consuming
 * glibc rand() in exactly the reference's order (including the draws it throws away:
 * yInPrev/yInCur :54-57, dummy skipOut :95, dummy Xt :98-102).  Caller has called
 * nvw_srand().  All weight outputs are col-major like Matrix.
 *   sel   [N][B]              (Matrix(B,N) col-major)
 *   embP/embC [A][R]          (Matrix(R,A))
 *   per layer l: Wprev/Wcur 2R*R, Bh 2R, Wres R*R, Bres R, Wskip S*R, Bskip S, packed
 *   consecutively per layer in arrays of L*size
 *   Wzs A*S, Bzs A, Wza A*A, Bza A, Lh [N][L][B][2R]  (Matrix(2R, N*L*B))
 */
void nvw_gen_test_inputs(int R, int S, int A, int L, int B, int N, float* sel, float* embP,
                         float* embC, float* Wprev, float* Wcur, float* Bh, float* Wres,
                         float* Bres, float* Wskip, float* Bskip, float* Wzs, float* Bzs,
                         float* Wza, float* Bza, float* Lh) {
    float mean = 0.0;
    float scale = 0.5 / R;
    for (int b = 0; b < B; b++) { (void)(rand() % A); (void)(rand() % A); }
    nvw_randomize(sel, B, N, 0.5, 1.0);
    nvw_randomize(embP, R, A, mean, scale);
    nvw_randomize(embC, R, A, mean, scale);
    size_t dummyN = (size_t)(S > R ? S : R) * B;
    float* dummy = (float*)malloc(dummyN * sizeof(float));
    for (int l = 0; l < L; l++) {
        /* createMatrix(r,c): scale = 0.5 / r  (nv_wavenet_test.cu:36-42) */
        nvw_randomize(Wprev + (size_t)l * 2 * R * R, 2 * R, R, 0.0, 0.5 / (2 * R));
        nvw_randomize(Wcur + (size_t)l * 2 * R * R, 2 * R, R, 0.0, 0.5 / (2 * R));
        nvw_randomize(Bh + (size_t)l * 2 * R, 2 * R, 1, 0.0, 0.5 / (2 * R));
        nvw_randomize(Wres + (size_t)l * R * R, R, R, 0.0, 0.5 / R);
        nvw_randomize(Bres + (size_t)l * R, R, 1, 0.0, 0.5 / R);
        nvw_randomize(Wskip + (size_t)l * S * R, S, R, 0.0, 0.5 / S);
        nvw_randomize(Bskip + (size_t)l * S, S, 1, 0.0, 0.5 / S);
        nvw_randomize(dummy, S, B, 0.0, 0.5 / S);
    }
    for (int s = 0; s < N; s++)
        for (int l = 0; l < L + 1; l++) nvw_randomize(dummy, R, B, 0.0, 0.5 / R);
    free(dummy);
    nvw_randomize(Wzs, A, S, mean, scale);
    nvw_randomize(Bzs, A, 1, mean, scale);
    nvw_randomize(Wza, A, A, mean, scale);
    nvw_randomize(Bza, A, 1, mean, scale);
    nvw_randomize(Lh, 2 * R, N * L * B, mean, scale);
}

/* CRC-32 (IEEE) of a byte buffer: fixtures pin large activation dumps by checksum. */
uint32_t nvw_crc32(const void* data, size_t n, uint32_t crc) {
    const uint8_t* p = (const uint8_t*)data;
    crc = ~crc;
    for (size_t i = 0; i < n; i++) {
        crc ^= p[i];
        for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}

/* ------------------------------------------------------------------------------------------
 * SURVEY.md 8f rank 2 extensions (not in the reference's C++ path): checker side.
 * ------------------------------------------------------------------------------------------ */

/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11; Random123 philox.h).  Pinned by Random123's known-answer vectors in
 * tests/test_oracle_cpu.py.  This is what replaces libc rand() of pytorch/wavenet_infer.cu:92-94
 * when the selectors are drawn inside the kernel. */
void nvw_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        if (r) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* selector of (sample t, utterance b) = top 24 bits of word 0 of Philox(ctr = {t, b, 0, 0},
 * key = seed) scaled to [0,1): the same layout, [N][B], as the uploaded selector matrix. */
void nvw_philox_selectors(uint64_t seed, int N, int B, float* sel) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int t = 0; t < N; t++)
        for (int b = 0; b < B; b++) {
            uint32_t ctr[4] = {(uint32_t)t, (uint32_t)b, 0u, 0u}, out[4];
            nvw_philox4x32_10(ctr, key, out);
            sel[(size_t)t * B + b] = (float)(out[0] >> 8) * (1.0f / 16777216.0f);
        }
}

/* pytorch/utils.py:62-70 (mu_law_decode_numpy) followed by pytorch/inference.py:58-60
 * (MAX_WAV_VALUE * audio, astype('int16')): float64 arithmetic, truncation toward zero, and the
 * top bin (signal = +1 -> 32768.0) wraps to -32768 exactly as numpy's cast does on x86.
 * Pinned by tests/golden/mulaw_pcm.npz, generated from the reference's own utils.py. */
void nvw_mulaw_pcm_table(int A, int16_t* table) {
    const double mu = (double)A - 1.0;
    for (int y = 0; y < A; y++) {
        const double signal = 2.0 * ((double)y / mu) - 1.0;
        const double magnitude = (1.0 / mu) * (pow(1.0 + mu, fabs(signal)) - 1.0);
        const double sgn = signal > 0 ? 1.0 : (signal < 0 ? -1.0 : 0.0);
        const double v = 32768.0 * (sgn * magnitude);
        table[y] = (int16_t)(int32_t)v;
    }
}
