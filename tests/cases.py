"""Parity test matrix.

REF_CASES mirrors the 16 runTest() invocations of the reference's own parity harness
(/root/reference/nv_wavenet_test.cu:331-394): same seeds, same order of rand() consumption
(several invocations share one srand(), so a case's inputs depend on the shapes generated
before it), L=20 (L=12 for the A=1024 case), B=16, N=8 samples per iteration, maxDilation=8,
2 iterations, chunk size 7.  impl is the reference's Implementation enum value
(nv_wavenet.cuh:223-229): 1 single-block, 2 dual-block, 3 persistent, 4 many-block.

EXTRA_CASES are ours: the BASELINE.json config shapes (C1, C2, C3, C4) over horizons long
enough to exercise the large dilations and the ring wrap-around.
"""
from collections import namedtuple

Shape = namedtuple("Shape", "R S A L B N maxD")
Case = namedtuple("Case", "name seed prior shape impl iters chunk")

_T = lambda R, S, A, L=20: Shape(R, S, A, L, 16, 8, 8)

REF_CASES = []
for _seed, _shape, _impls in (
        (3, _T(32, 128, 256), (1, 2, 3, 4)),
        (10, _T(64, 128, 256), (1, 2, 3, 4)),
        (30, _T(64, 256, 256), (1, 2, 3, 4)),
        (50, _T(128, 256, 256), (3, 4)),
):
    for _k, _impl in enumerate(_impls):
        REF_CASES.append(Case("R%dS%dA%d_impl%d" % (_shape.R, _shape.S, _shape.A, _impl), _seed,
                              [_shape] * _k, _shape, _impl, 2, 7))
REF_CASES.append(Case("R64S128A512_impl3", 70, [], _T(64, 128, 512), 3, 2, 7))
REF_CASES.append(Case("R128S256A1024_impl3", 70, [_T(64, 128, 512)], _T(128, 256, 1024, 12), 3, 2, 7))

EXTRA_CASES = [
    # C1: the reference's CPU-runnable plumbing config (BASELINE.json configs[0])
    Case("C1_R32S128A256_L8_B1", 3, [], Shape(32, 128, 256, 8, 1, 64, 8), 1, 1, 64),
    # C2 shape: maxDilation=512, long enough (N>2*513) to wrap every ring and use d=512 twice
    Case("C2_R64S128A256_L20_B4_maxD512", 10, [], Shape(64, 128, 256, 20, 4, 1100, 512), 1, 1, 300),
    # C3 shape, B=16 (one full MFMA batch tile), moderate horizon, maxD 32 to wrap often
    Case("C3_R64S256A256_L20_B16", 30, [], Shape(64, 256, 256, 20, 16, 96, 32), 3, 1, 40),
    # C3 shape with a ragged batch (B not a multiple of the 16-wide tile, >1 tile)
    Case("C3_R64S256A256_L20_B21", 31, [], Shape(64, 256, 256, 20, 21, 40, 16), 3, 1, 16),
    # C4 shape: R=128, 30 layers
    Case("C4_R128S256A256_L30_B8", 50, [], Shape(128, 256, 256, 30, 8, 48, 16), 2, 1, 20),
    # R=256: only the reference's perf harness instantiates it (nv_wavenet_perf.cu:156-166)
    Case("R256S256A256_L6_B5", 90, [], Shape(256, 256, 256, 6, 5, 24, 8), 3, 1, 10),
    # odd layer counts: the engine alternates two prefetch register sets by layer parity, and an odd
    # count flips the parity from one sample to the next
    Case("R64S128A256_L7_B19_oddL", 91, [], Shape(64, 128, 256, 7, 19, 40, 4), 1, 1, 13),
    Case("R64S256A256_L3_B16_oddL", 92, [], Shape(64, 256, 256, 3, 16, 24, 2), 3, 1, 24),
    # S > 4R (S = 8R): the reference's persistent variant has latent bugs there (nv_wavenet_persistent.cuh:496,528:
    # skip block `layer = block_id*s_tiles`, `assert(S%4*R==0)` precedence) and its single-block variant refuses it
    # (nv_wavenet.cuh:511); the CPU reference handles any S, and so does this engine
    Case("R32S256A256_L6_B5_S8R", 93, [], Shape(32, 256, 256, 6, 5, 24, 8), 3, 1, 10),
]

ALL_CASES = REF_CASES + EXTRA_CASES
BY_NAME = {c.name: c for c in ALL_CASES}

# Activation tolerances of the reference harness (nv_wavenet_test.cu:273-298); the compare is
# the reference's matrix_compare (matrix.cpp:133-151): |gpu/ref| - 1 <= tol, relu-aware variant.
TOL = dict(Xout=1e-2, skipOut=1e-2, Zs=1e-4, Za=1e-4, P=1e-3)
RELU_AWARE = dict(Xout=False, skipOut=True, Zs=True, Za=False, P=False)
