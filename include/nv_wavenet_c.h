/*
 * nv_wavenet_c.h -- C ABI onto every nvWavenetInfer<T_weight,T_data,R,S,A> instantiation that
 * libwavenet_infer.so carries.  One handle = one engine object; each call maps 1:1 onto the
 * member of the same name of the reference class (/root/reference/nv_wavenet.cuh:220-640):
 *
 *   nvw_create            <- nvWavenetInfer(numLayers, maxDilation, batchSize, numSamples, impl,
 *                                           tanhEmbed)                        nv_wavenet.cuh:311
 *   nvw_set_embeddings    <- setEmbeddings                                   nv_wavenet.cuh:396-399
 *   nvw_set_layer_weights <- setLayerWeights                                 nv_wavenet.cuh:400-409
 *   nvw_set_out_weights   <- setOutWeights                                   nv_wavenet.cuh:410-415
 *   nvw_set_inputs        <- setInputs                                       nv_wavenet.cuh:417-422
 *   nvw_run / nvw_run_partial / nvw_run_chunks <- run / run_partial / run_chunks
 *                                                                            nv_wavenet.cuh:445-639
 *   nvw_get_*             <- getXtOut/getSkipOut/getZs/getZa/getP/getYOut    nv_wavenet.cuh:424-444
 *
 * The reference has no such header: its only FFI is pytorch/wavenet_infer.h, which hard-wires one
 * instantiation (see include/wavenet_infer.h).  This one exists so that non-C++ hosts (Python
 * ctypes, tests, bench.py) can reach fp16, other channel counts, chunked streaming and the
 * debug getters without a compiler.  Plain pointers and ints only; `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  Pointers may be host or device memory wherever the
 * reference accepts both.  Precision is 32 (<float,float>) or 16 (<half2,half>).
 *
 * Errors: unsupported (R,S,A,precision) -> nvw_create returns NULL; run calls return 1 on
 * success and 0 when the launch failed (the reference's bool); HIP failures print
 * "GPUassert: ..." and exit like the reference's gpuErrChk.
 */
#ifndef NV_WAVENET_C_H
#define NV_WAVENET_C_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nvw_engine nvw_engine;

/* consumer callback of nvw_run_chunks: (yOut, first sample of the chunk, samples in it, user) */
typedef void (*nvw_consume_fn)(int* yOut, int init_sample, int count, void* user);

/* Revision of this interface.  It changes whenever an entry point or the meaning of an argument does -- in particular the
 * organisation codes of nvw_create_ex, which were renumbered once (round 3) and lost code 9 in round 4: a caller built against
 * another revision should check this instead of finding a different kernel behind a number.  5 = round 5 (the
 * feature-conditioning entry points below); 6 = this header (round 6: nvw_get_features returns int; nvw_upsample_features,
 * nvw_generate_stream and nvw_get_features check their ranges and refuse with 0 instead of reaching the class's asserts;
 * organisation 10; nvw_set_ring_in_lds). */
#define NVW_ABI_VERSION 6
int nvw_abi_version(void);
int nvw_supported(int R, int S, int A, int precision);
/* writes up to `max` (R,S,A,precision) quadruples into out[4*i..], returns how many exist */
int nvw_list_supported(int* out, int max);

nvw_engine* nvw_create(int R, int S, int A, int precision, int num_layers, int max_dilation,
                       int batch_size, int num_samples, int implementation, int tanh_embed);
/* The same with an explicit kernel organisation (the last, optional argument of this repo's nvWavenetInfer
 * constructor; 0 = from `implementation` and the batch size like nvw_create):
 *   1 wavenet_wg (1 to 4 tiles of 16 utterances per workgroup by batch size)   2 / 3 / 4 wavenet_wg with exactly 1 / 2 / 3
 *   (three: fp16, R <= 64; two tiles otherwise)   5 wavenet_chain (multi-CU, resident weights, fewest CUs)
 *   6 wavenet_chain with one layer per CU   7, 8, 9 retired (were wavenet_bcast -- every wave the whole network for its own
 *   tile, weights broadcast through an LDS ring -- and its variants; removed in round 5): refused like any number out of range
 *   10 wavenet_wg with four tiles per workgroup (round 6: fp16, R <= 64, dump-free launches with packed conditioning;
 *   other launches of such an engine take three).
 * Returns NULL when the shape does not fit a CU in that organisation (the reference's variants print
 * and return false for shapes they do not support, nv_wavenet_singleblock.cuh:273-286). */
nvw_engine* nvw_create_ex(int R, int S, int A, int precision, int num_layers, int max_dilation,
                          int batch_size, int num_samples, int implementation, int tanh_embed,
                          int organisation);
void nvw_destroy(nvw_engine* e);

void nvw_set_embeddings(nvw_engine* e, float* embed_prev, float* embed_cur);
void nvw_set_layer_weights(nvw_engine* e, int layer, float* Wprev, float* Wcur, float* Bh,
                           float* Wres, float* Bres, float* Wskip, float* Bskip);
void nvw_set_out_weights(nvw_engine* e, float* Wzs, float* Bzs, float* Wza, float* Bza);
void nvw_set_inputs(nvw_engine* e, float* Lh, float* output_selectors);

/* Extensions beyond the reference class (its "next" list: selectors drawn on the device instead of
 * the host rand() table of pytorch/wavenet_infer.cu:92-94; mu-law expansion of pytorch/utils.py:62-70
 * + pytorch/inference.py:58-60 done on the device):
 *   nvw_set_conditioning   the conditioning half of setInputs (Lh [N][L][B][2R]; history := 128)
 *   nvw_set_selector_seed  selectors = Philox4x32-10(counter {sample, utterance, 0, 0}, key = seed),
 *                          top 24 bits of word 0 / 2^24; stays in force until the next nvw_set_inputs
 *   nvw_set_audio_out      pcm_out: caller-owned [batch][samples] int16 (host or device), filled by
 *                          the run calls wherever yOut is: int16(32768 * mu_law_decode(y, A)); NULL
 *                          switches it off */
void nvw_set_conditioning(nvw_engine* e, float* Lh);
/* Utterances shorter than the engine's capacity: num_samples <= the num_samples of nvw_create rows of
 * Lh / output_selectors (both layouts are sample-major, so a prefix is a complete input) */
void nvw_set_inputs_n(nvw_engine* e, float* Lh, float* output_selectors, int num_samples);
void nvw_set_conditioning_n(nvw_engine* e, float* Lh, int num_samples);
/* Conditioning streamed chunk by chunk: packs samples [first_sample, first_sample + count) (Lh points at
 * sample first_sample, device memory) asynchronously on `stream`, e.g. behind nvw_run_partial of the
 * previous chunk on another stream.  Does not touch the sample history. */
void nvw_pack_conditioning(nvw_engine* e, float* Lh, int first_sample, int count, void* stream);
/* Device-resident conditioning consumed IN PLACE (no packed copy; pytorch/README.md:44 recommends keeping cond_input on
 * the device, wavenet_infer.cu:124-143 still copies it): Lh is fp32 [num_samples][L][batch][2R] in device memory, owned by
 * the caller and kept alive and unchanged until the run calls that follow have completed.  Resets the history like
 * nvw_set_inputs; pair with nvw_set_selector_seed.  Same samples as the packed path, bit for bit. */
void nvw_set_conditioning_direct(nvw_engine* e, float* Lh, int num_samples);
/* The same for a tensor of `precision` bits per element: 32 (float) or -- fp16 engines only -- 16 (IEEE half, the engine's
 * T_data: the reference keeps its conditioning in T_data, nv_wavenet.cuh:326; half the bytes of the fp32 tensor).  Returns 0
 * (and changes nothing) when the engine cannot read that element type in place. */
int nvw_set_conditioning_direct_t(nvw_engine* e, const void* Lh, int num_samples, int precision);
/* Conditioning PRODUCED in the engine's own fragment order by the caller (device memory, the engine's T_data):
 * [num_samples + 1][L][nvw_cond_tiles(e)][wave][fragment][lane][8 (fp16) | 4 (fp32)] with the gate rows pre-scaled, i.e. what
 * nvw_pack_conditioning writes (order: pack_cond_tiled_kernel in wn_kernels.hpp; nv_wavenet_amd/nv_wavenet.py:cond_fragment_order
 * gives it as a channel permutation + scale, which a model folds into its conditioning convolution for free).  The generation
 * kernels run their packed path on that buffer: no copy, no second pass.  One padding sample past the last; the buffer stays
 * alive and unchanged until the run calls that follow have completed.  Resets the history like nvw_set_inputs. */
void nvw_set_conditioning_packed(nvw_engine* e, const void* frags, int num_samples);
/* the same with the buffer's size stated (elements of the engine's T_data): returns 0 and changes nothing unless it holds
 * (num_samples + 1) x layers x nvw_cond_tiles(e) x 16 x 2R elements.  Either way the run calls that follow refuse (assert, like
 * the other preconditions of the path) to generate more samples than were handed over here. */
int nvw_set_conditioning_packed_n(nvw_engine* e, const void* frags, int num_samples, size_t elems);
int nvw_cond_tiles(nvw_engine* e);
/* The PRODUCER of such a buffer for fp16 engines (role of the model's `cond_layers` 1x1 convolution, pytorch/wavenet.py:190-202, with
 * the engine's channel order and gate pre-scale folded into its weights): one MFMA kernel from the upsampled features straight
 * into fragment order, no intermediate tensor, no permuting copy.  All pointers are device memory:
 *   x      [tiles*16][num_samples][32*kfrags] fp16: upsampled features, channels last, zero-padded to whole 32-feature fragments
 *   wfrag  [num_layers][nwf][2][kfrags][64][8] fp16 and bias [num_layers][nwf*32] fp32: the convolution's weights and bias as
 *          nv_wavenet_amd/nv_wavenet.py:cond_producer_weights arranges them (nwf = 2R/32 fragments per tile)
 *   out    [num_samples][num_layers][tiles][nwf][64][8] fp16: `num_samples` samples of the buffer handed to
 *          nvw_set_conditioning_packed_n (tiles = nvw_cond_tiles(e))
 * Asynchronous on `stream`; returns 0 when the arguments are out of range (1 <= kfrags <= 4) or the launch failed. */
int nvw_produce_conditioning_f16(const void* x, const void* wfrag, const float* bias, void* out, int tiles, int num_samples,
                                 int num_layers, int kfrags, int nwf, void* stream);
/* CONDITIONING COMPUTED IN THE GENERATION KERNEL (round 5).  The reference's pipeline builds cond_input = cond_layers(upsample(
 * features)) -- [2R][B][L][N], 2R*L values per utterance and sample -- in PyTorch and hands it to the engine
 * (pytorch/wavenet.py:190-202, inference.py:52-53).  Here the 1x1 convolution `cond_layers` moves INTO the generation kernel: the
 * caller hands over its weights once and, per utterance, only the upsampled features (n_cond values per utterance and sample);
 * Lh[t][l] = Wcond[l] c[t] + bcond[l] is computed where it is consumed (n_cond more k steps of the gate GEMM), and the
 * [N][L][B][2R] tensor is never built.  An additional contract beside nvw_set_inputs / nvw_set_conditioning*, which stay as they are.
 *   nvw_max_cond_channels          feature channels the kernels are built for (80 = the reference's config.json)
 *   nvw_set_conditioning_weights   Wcond [L][2R][n_cond] (cond_layers.weight, [2R*L][n_cond][1] as it is), bcond [L][2R]; fp32, host
 *                                  or device, copied.  0 when n_cond is out of range (nothing changes).
 *   nvw_set_features               the upsampled features of the whole utterance from a device tensor of `precision`-bit floats
 *                                  (32 | 16) addressed x[b*b_stride + c*c_stride + t*t_stride] (upsample output [B][n_cond][T]:
 *                                  strides n_cond*T, T, 1); copied into the engine's fragment order; resets the history like
 *                                  nvw_set_inputs; pair with nvw_set_selector_seed or nvw_set_selectors.
 *   nvw_pack_features              samples [first_sample, first_sample + count) only (x points at sample first_sample),
 *                                  asynchronously on `stream`, history untouched: streaming chunk by chunk like nvw_pack_conditioning
 *   nvw_set_conditioning_features  features the caller produced in fragment order itself, used in place:
 *                                  [num_samples][nvw_cond_tiles(e)][nvw_feature_fragments(e)][64][8 fp16 | 4 fp32] of the engine's
 *                                  T_data; fragment kf, lane (g, j), element e = channel (kf*TPF + (e>>2))*16 + 4g + (e&3) of utterance
 *                                  tile*16 + j (TPF = 2 fp16 | 1 fp32), zero beyond n_cond; `elems` = the buffer's size
 * All return 1 on success, 0 (after a message) when refused.  Runs that follow launch wn::wavenet_wg<.., RAW=3> whatever the
 * engine's organisation. */
int nvw_max_cond_channels(void);
int nvw_set_conditioning_weights(nvw_engine* e, const float* Wcond, const float* bcond, int n_cond);
int nvw_feature_fragments(nvw_engine* e);
size_t nvw_feature_elems(nvw_engine* e, int num_samples);
int nvw_set_conditioning_features(nvw_engine* e, const void* frags, int num_samples, size_t elems);
int nvw_pack_features(nvw_engine* e, const void* x, int precision, long long b_stride, long long c_stride, long long t_stride,
                      int first_sample, int count, void* stream);
int nvw_set_features(nvw_engine* e, const void* x, int precision, long long b_stride, long long c_stride, long long t_stride,
                     int num_samples);
/* FEATURES IN, AUDIO OUT (round 5; role of pytorch/inference.py:40-62 around run_chunks, nv_wavenet.cuh:445-497).  The other half of
 * WaveNet.get_cond_input -- the `upsample` ConvTranspose1d and the trimming of its tail, pytorch/wavenet.py:195-197 -- on the engine's
 * own MFMA kernel, writing the feature fragments the generation kernel reads:
 *   nvw_set_upsampling      upsample.weight [n_cond][n_cond][window] and .bias [n_cond] (fp32, host or device, copied); window a multiple
 *                           of stride, at most 5 strides (the reference: 800 / 200); after nvw_set_conditioning_weights
 *   nvw_set_mel             the utterances' frames before upsampling, device tensor of 16- or 32-bit floats addressed
 *                           x[b*b_stride + c*c_stride + f*f_stride] ([B][n_cond][frames]: strides n_cond*frames, frames, 1); copied;
 *                           frames * stride <= the engine's samples; resets the history like nvw_set_inputs (start of a batch)
 *   nvw_upsample_features   samples [first_sample, first_sample + count) of the upsampled features, asynchronously on `stream`
 *   nvw_generate_stream     the whole loop: per chunk of num_samples_per_chunk samples the upsampling, the generation launch, the copy of
 *                           the chunk's samples to yOut ([batch][num_samples] int32, host or device; may be NULL) and of the int16 PCM to
 *                           the buffer of nvw_set_audio_out on a second stream, and consume(yOut, first, count, user) on the calling
 *                           thread; selectors: nvw_set_selector_seed / nvw_set_selectors.  Returns when the last chunk is consumed.
 * All return 1 on success, 0 when refused. */
int nvw_set_upsampling(nvw_engine* e, const float* up_w, const float* up_b, int window, int stride);
int nvw_set_mel(nvw_engine* e, const void* mel, int precision, long long b_stride, long long c_stride, long long f_stride, int frames);
int nvw_upsample_features(nvw_engine* e, int first_sample, int count, void* stream);
/* debug getter: samples [first_sample, first_sample + count) of the engine's feature buffer (what nvw_pack_features / nvw_upsample_features
 * wrote: fragment order, the engine's T_data, nvw_feature_elems(e, count) elements) -> dst (host or device); synchronises.
 * 1, or 0 when refused: no feature buffer yet, or the range lies outside the engine's samples (ABI 6: returned void before) */
int nvw_get_features(nvw_engine* e, void* dst, int first_sample, int count);
int nvw_generate_stream(nvw_engine* e, int num_samples_per_chunk, nvw_consume_fn consume, void* user, int num_samples, int batch_size, int* yOut,
                        void* stream);
/* the selector half of nvw_set_inputs ([num_samples][batch] uniform draws, host or device); conditioning and history untouched */
void nvw_set_selectors(nvw_engine* e, float* output_selectors, int num_samples);
/* Multi-CU (wavenet_chain) launches need all their workgroups resident at once; when other work holds CUs a launch gives up
 * after a bounded wait and the engine re-runs its samples on wavenet_wg from the state the launch started with, in stream
 * order, so the samples delivered are the right ones either way.  nvw_chain_status: 0, or the code of a give-up that could
 * not be repaired; nvw_chain_fallbacks: launches that were re-run; nvw_chain_last_timeout: code of the latest of those
 * (0x100+stage x hand-off, 0x200+stage skip sums, 0x300 head, 0x400+ placement exchange).  All three synchronise the device.
 * nvw_set_chain_timeout_ms: the bound of every hand-off wait (default 1500 ms). */
unsigned nvw_chain_status(nvw_engine* e);
unsigned nvw_chain_fallbacks(nvw_engine* e);
unsigned nvw_chain_last_timeout(nvw_engine* e);
void nvw_set_chain_timeout_ms(nvw_engine* e, double ms);
/* Measurement aid: with the probe on, workgroup 0 of every single-workgroup-organisation launch records the shader-clock and the
 * constant-rate wall-clock counters at its start and end; nvw_last_launch_clock_ghz returns shader ticks per wall second of the
 * latest launch, i.e. the clock the chip granted it under its power budget (0 when nothing was probed); synchronises the device. */
void nvw_set_clock_probe(nvw_engine* e, int on);
/* The dilation ring on chip (round 6; the reference stages x[t-d] through shared memory out of a global ring, nv_wavenet.cuh:96-127,
 * 334-335): single-workgroup launches keep the ring slots of the layers with the shortest dilations in the LDS their tables leave
 * free, loaded from / spilled to the HBM ring at the launch's ends (1-KiB rows).  mode >= 0 (default): as many layers as fit -- a model
 * whose whole ring fits has no ring traffic to HBM during the launch; -1: never.  Same samples either way. */
void nvw_set_ring_in_lds(nvw_engine* e, int mode);
double nvw_last_launch_clock_ghz(nvw_engine* e);
/* Samples [init_sample, init_sample + count) of a num_samples-long utterance, asynchronously on `stream`
 * (one chunk of run_chunks, for hosts that drive the chunks themselves); nvw_reset_history starts a new
 * utterance -- sample history back to 128, dilation rings back to zero -- like nvw_set_inputs does, without
 * touching the conditioning. */
int nvw_run_range(nvw_engine* e, int init_sample, int count, int num_samples, int batch_size, void* stream);
void nvw_reset_history(nvw_engine* e, void* stream);
void nvw_set_selector_seed(nvw_engine* e, unsigned long long seed);
void nvw_set_audio_out(nvw_engine* e, short* pcm_out);
/* the same with the buffer's size in int16 values stated: every run call then checks batch * num_samples <= elems */
void nvw_set_audio_out_n(nvw_engine* e, short* pcm_out, size_t elems);
/* Introspection: the device code nvw_run(e, n, batch_size, ..., dump_activations, ...) launches, e.g.
 * "wn::wavenet_wg<fp16,64,256,256,BT=2,EMBLDS=1,DUMP=0> tiles/wg=2 wgs=256 lds=149120" */
void nvw_kernel_info(nvw_engine* e, int batch_size, int dump_activations, char* buf, int buf_size);

int nvw_run(nvw_engine* e, int num_samples, int batch_size, int* yOut, int batch_size_per_block,
            int dump_activations, void* stream);
int nvw_run_partial(nvw_engine* e, int init_sample, int num_samples, int batch_size, int* yOut,
                    int batch_size_per_block, int dump_activations, void* stream);
int nvw_run_chunks(nvw_engine* e, int num_samples_per_chunk, nvw_consume_fn consume, void* user,
                   int num_samples, int batch_size, int* yOut, int batch_size_per_block,
                   int dump_activations, void* stream);

void nvw_get_xt_out(nvw_engine* e, int layer, float* dst);     /* [maxBatch][R] */
void nvw_get_skip_out(nvw_engine* e, int layer, float* dst);   /* [maxBatch][S] */
void nvw_get_zs(nvw_engine* e, float* dst);                    /* [maxBatch][A] */
void nvw_get_za(nvw_engine* e, float* dst);                    /* [maxBatch][A] */
void nvw_get_p(nvw_engine* e, float* dst);                     /* [maxBatch][A] */
void nvw_get_y_out(nvw_engine* e, int* yOut, int offset, int size, void* stream);

/* hipDeviceSynchronize() for hosts without a HIP binding */
void nvw_device_synchronize(void);
/* time `reps` back-to-back nvw_run() launches with HIP events on `stream`; returns milliseconds
 * for all reps (used by bench.py: events on the stream the kernel is launched on) */
float nvw_time_runs(nvw_engine* e, int reps, int num_samples, int batch_size, int batch_size_per_block,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif
