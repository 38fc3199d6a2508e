// wn_experiments.hpp -- every compile-time switch of the kernels that is NOT part of the product, in one place.
//
// A default build defines none of the ablations and takes every tunable at the value below; the shipped library is built that
// way (csrc/Makefile passes no -DWN_*).  A/B builds for timing experiments go through scripts/build_variant.sh <name> "<-D...>"
// and are loaded with NVW_LIB=<that library>; the measurements behind the values are in LABNOTES.md.
#pragma once

// ---- tunables (each settled by measurement; change one with -D for an A/B build) ---------------------------------------------
#ifndef WN_PFMAX
#define WN_PFMAX 12          // wavenet_wg: largest depth of the per-wave weight prefetch ring (the divisor of a layer's stream <= this: 9)
#endif
#ifndef WN_PFMAX_FEAT
#define WN_PFMAX_FEAT 12     // ... of the kernels that compute the conditioning themselves (a layer's stream is 24 fragments at C3: 12)
#endif
#ifndef WN_HEADREGS
#define WN_HEADREGS 128      // wavenet_wg, one tile per workgroup: accumulator registers the resident A x A head matrix may take
#endif
#ifndef WN_HEADREGS2
#define WN_HEADREGS2 128     // ... two tiles per workgroup
#endif
#ifndef WN_HEADREGS3
#define WN_HEADREGS3 0       // ... three tiles per workgroup: the whole head is streamed
#endif
#ifndef WN_COND_VALU
#define WN_COND_VALU 1       // packed / in-place conditioning added by v_fma_mix_f32 (1, round 6) or through 0/1 selection MFMAs (0, rounds 1-5)
#endif
#ifndef WN_LDS_RING_MAXD
#define WN_LDS_RING_MAXD 512 // wavenet_wg: largest dilation whose ring slots may live in LDS during a launch (what fits is decided per launch; 0: none)
#endif
#ifndef WN_SOFTMAX_2PASS
#define WN_SOFTMAX_2PASS 0   // wavenet_wg, four tiles per workgroup: sample picks in two passes of two tiles (frees LDS for the embedding table); measured: no gain
#endif
#ifndef WN_CHAIN_HOIST_FROM
#define WN_CHAIN_HOIST_FROM 4  // wavenet_chain: launches with more tiles per chain than this use the HOIST instantiation (99: never).  Measured at C4,
                               // steady-state kHz per utterance with / without: 4 tiles 27.5 / 27.7, 5 tiles 24.4 - 24.7 / 23.5 - 23.8, 6 tiles 20.8 - 20.9 / 19.9 - 20.0
#endif
#ifndef WN_WG4_FROM
#define WN_WG4_FROM 3        // AUTO: four tiles per workgroup for batches beyond this many tiles per CU (3: beyond the three-tile capacity)
#endif
#ifndef WN_HEADREGS4
#define WN_HEADREGS4 0       // ... four tiles per workgroup: the whole head is streamed (428 - 440 registers without it)
#endif
#ifndef WN_ZS_B_REGS
#define WN_ZS_B_REGS 4096    // wavenet_wg head: ... of the A x S GEMM (4096: always in registers)
#endif
#ifndef WN_ZA_B_REGS
#define WN_ZA_B_REGS 128     // wavenet_wg head: the B fragments of the A x A GEMM stay in registers up to this many, else they are read from LDS as they are needed
#endif
#ifndef WN_TAKE_G
#define WN_TAKE_G 1          // wavenet_wg: weight fragments waited for together (take_group); > 1 measured slower
#endif
#ifndef WN_REQ_AT
#define WN_REQ_AT 6          // wavenet_wg: eighths of the skip GEMM behind which taps and conditioning of layer l+2 are requested
#endif
#ifndef WN_REQ_AT_FEAT
#define WN_REQ_AT_FEAT 6     // ... of the kernels that compute the conditioning: eighths of the (skip + conditioning) fragments under the gate (4: +0.7 %)
#endif
#ifndef WN_UP_PHASES
#define WN_UP_PHASES(F16) 1     // upsample_features_kernel: phases per workgroup pass (2, fp16 only: measured slower)
#endif
#ifndef WN_UP_COLS
#define WN_UP_COLS(F16) (WN_UP_PHASES(F16) > 1 ? 2 : 4)      // ... columns a wave takes per pass
#endif
#ifndef WN_UP_WAVES
#define WN_UP_WAVES(F16) (WN_UP_PHASES(F16) > 1 ? 8 : 4)     // ... waves per workgroup
#endif
#ifndef WN_UP_ATTR
#define WN_UP_ATTR              // ... extra attributes (e.g. __attribute__((amdgpu_waves_per_eu(4,4))))
#endif
// cache-policy bits of the buffer instructions (0 = default, 2 = nt / streaming, 16 = sc1)
#ifndef WN_W_AUX
#define WN_W_AUX 0           // the weight stream: must stay in L2 (nt: +20 % per sample)
#endif
#ifndef WN_RING_LD_AUX
#define WN_RING_LD_AUX 2     // dilated taps
#endif
#ifndef WN_RING_ST_AUX
#define WN_RING_ST_AUX 2     // ring stores
#endif
#ifndef WN_COND_AUX
#define WN_COND_AUX 2        // packed conditioning
#endif
#ifndef WN_RAW_AUX
#define WN_RAW_AUX 0         // conditioning read in place from the caller's [N][L][B][2R] tensor
#endif
#ifndef WN_FEAT_AUX
#define WN_FEAT_AUX 0        // upsampled features (in-kernel conditioning): every wave of a workgroup reads the same fragments
#endif

// ---- ablations --------------------------------------------------------------------------------------------------------------------
// The timing-only builds of rounds 2-4 (WN_ABL_NOACT, _NOWEIGHTLOAD, _NOXP, _NOCOND, _HOTLOADS / _HOTTAPS / _HOTCOND, _NOTAPGEMM,
// _NOBARRIER, _NOHEADRES: gate without transcendentals, no weight refills, no tap / conditioning loads or L2-resident ones, ...) were
// taken out of wn_kernels.hpp in round 5: their measurements are in LABNOTES.md (rounds 2-4), their code in commit c19e716.  What they
// were built to decide -- whether the full-chip launch is bound by its memory traffic -- was settled by removing the conditioning
// stream for real (round 5).
//   WN_EXP_TWO_WG    round 6: the one-tile wavenet_wg with a register budget for TWO workgroups per CU (two waves per SIMD, each with its own tile
//                    and weight stream; no embedding table / ring slots in LDS), AUTO launching one-tile workgroups up to two tiles per CU: the A/B of
//                    the round-5 review's item 4 (8 192 utterances: 40.2 us per sample against 27.7 for the two-tile workgroup; LABNOTES round 6)
// ---- probes (results stay right) ------------------------------------------------------------------------------------------
//   WN_TIMING        wavenet_wg: per-phase shader-clock sums of wave 0 into Params::p (scripts/quick_phase.py)
//   WN_CHAIN_TIMING  wavenet_chain: wall-clock stamps per stage (scripts/chain_phase.py)
