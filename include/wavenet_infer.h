/*
 * wavenet_infer.h -- C ABI of libwavenet_infer.so, the drop-in boundary of the PyTorch path.
 *
 * Replaces, symbol for symbol, /root/reference/pytorch/wavenet_infer.h:33-58 (implemented in the
 * reference by pytorch/wavenet_infer.cu:34-149 on top of nvWavenetInfer<float,float,R,S,A>).
 * The reference's pybind wrapper (pytorch/wavenet_infer_wrapper.cpp:32-110) binds exactly these
 * entry points; INTEGRATION.md shows that binding unchanged against this library.
 *
 * Semantics kept from the reference:
 *   - channel counts are fixed at build time (get_R/get_S/get_A; defaults R=64, S=256, A=256,
 *     wavenet_infer.cu:35-37), arithmetic is fp32 (wavenet_infer.cu:38);
 *   - all weight pointers are fp32, column-major, HOST OR DEVICE (copied; caller keeps ownership);
 *   - output-layer biases are zero (wavenet_infer.cu:75-82);
 *   - the uniform draws for sampling are produced inside with libc rand() in the order of
 *     Matrix::randomize(0.5, 1.0) on a (batch x samples) matrix (wavenet_infer.cu:92-94,
 *     matrix.cpp:38-55), so a caller that seeds srand() gets the reference's draw sequence;
 *   - `samples` is a caller-owned [batch_size][sample_count] int32 buffer, host or device
 *     (the PyTorch wrapper passes a CUDA/HIP tensor, pytorch/nv_wavenet.py:182);
 *   - a fresh engine is built per call and the call returns after the device is idle
 *     (wavenet_infer.cu:97-98).
 *   - implementation: 0 AUTO, 1 SINGLE_BLOCK, 2 DUAL_BLOCK, 3 PERSISTENT, 4 MANYBLOCK
 *     (nv_wavenet.cuh:223-229).
 * Errors: HIP failures print "GPUassert: ..." and exit(code) like the reference's gpuErrChk
 * (nv_wavenet_util.cuh:34-40); there is no CPU fallback.
 */
#ifndef WAVENET_INFER_H
#define WAVENET_INFER_H

#ifdef __cplusplus
extern "C" {
#endif

void wavenet_infer(int sample_count,
                   int batch_size,
                   float* embedding_prev,           /* [A][R]                          */
                   float* embedding_curr,           /* [A][R]                          */
                   int num_layers,
                   int max_dilation,
                   float** in_layer_weights_prev,   /* [L] -> col-major 2R x R          */
                   float** in_layer_weights_curr,   /* [L] -> col-major 2R x R          */
                   float** in_layer_biases,         /* [L] -> 2R                        */
                   float** res_layer_weights,       /* [L] -> col-major R x R           */
                   float** res_layer_biases,        /* [L] -> R                         */
                   float** skip_layer_weights,      /* [L] -> col-major S x R           */
                   float** skip_layer_biases,       /* [L] -> S                         */
                   float* conv_out_weight,          /* col-major A x S                  */
                   float* conv_end_weight,          /* col-major A x A                  */
                   int use_embed_tanh,
                   float* cond_input,               /* [samples][L][batch][2R]          */
                   int implementation,
                   int* samples);                   /* out: [batch][samples]            */

/* channel counts this build of wavenet_infer() is fixed to (wavenet_infer.h:52-57) */
int get_R(void);
int get_S(void);
int get_A(void);

#ifdef __cplusplus
}
#endif
#endif
