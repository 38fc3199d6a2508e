"""numpy model of the ROUNDING POINTS of the fp16 engine -- TEST INFRASTRUCTURE ONLY.

Not the product and not the oracle: a third statement of the same network whose only purpose is to
size the fp16 parity bars on the CPU (tests/test_parity_bars_cpu.py) and to cross-check them without a
GPU.  It follows the oracle's arithmetic (oracle/wavenet_oracle.c, i.e. nv_wavenet_reference.cpp:42-121)
and rounds to IEEE fp16 exactly where nv_wavenet_amd/csrc/wn_kernels.hpp does:

  * gate matrices, gate conditioning: value * prescale (2 log2 e for tanh rows, -log2 e for sigmoid
    rows) rounded to fp16 (pack_weight_elem / pack_cond_tiled_kernel); gate bias prescaled in fp32;
  * every GEMM B operand (x_l[t], x_l[t-d] from the ring, h, relu(skip), relu(zs)) rounded to fp16
    (lds_put_tile<true>), the residual stream itself, every accumulator, the gate and the softmax in fp32;
  * gate = (1 - 2 / (2^a' + 1)) * 1 / (1 + 2^b')  on the prescaled pre-activations (gate1<true>).

Summation order inside a dot product is numpy's, not the MFMA's: fp32 accumulation differences
(~1e-7 relative) are three orders of magnitude below the fp16 operand rounding this model is about.
The reference itself accumulates in fp16 (matrix_math.cuh:119-157): this engine is strictly more
precise than what it replaces, and the fp32 oracle is the stricter yardstick.
"""
import numpy as np

LOG2E = np.float32(1.44269504088896340736)
f16 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def run(t, shape, y_forced, tanh_embed=True):
    """Teacher-forced on y_forced [B][N]: returns dict(y=[B][N] own picks, lo, hi = CDF edges of the own
    pick, Xout [L][B][R], skipOut [L][B][S], Zs, Za, P [B][A] of the last sample)."""
    R, S, A, L, B, N, maxD = shape[:7]
    f32 = np.float32
    # weights as the engine stores them; TestInputs keeps col-major M x K as numpy [K][M]
    pres = np.concatenate([np.full(R, 2 * LOG2E, f32), np.full(R, -LOG2E, f32)])
    Wprev = [f16(t.Wprev[l] * pres[None, :]) for l in range(L)]        # [R][2R]
    Wcur = [f16(t.Wcur[l] * pres[None, :]) for l in range(L)]
    Bh = [(t.Bh[l] * pres).astype(f32) for l in range(L)]
    Wres = [f16(t.Wres[l]) for l in range(L)]                           # [R][R]
    Wskip = [f16(t.Wskip[l]) for l in range(L)]                         # [R][S]
    Wzs, Wza = f16(t.Wzs), f16(t.Wza)                                   # [S][A], [A][A]
    embP, embC = f16(t.embP), f16(t.embC)                               # [A][R]
    dil = []
    d = 1
    for l in range(L):
        dil.append(d)
        d *= 2
        if d > maxD:
            d = 1
    hist = [dict() for _ in range(L)]          # layer -> {sample: fp16 x_l[sample]} (pruned)
    yP = np.full(B, 128, np.int64)
    yC = np.full(B, 128, np.int64)
    y = np.zeros((B, N), np.int32)
    lo = np.zeros((B, N), f32)
    hi = np.zeros((B, N), f32)
    out = {}
    for n in range(N):
        x = (embP[yP] + embC[yC]).astype(f32)                           # [B][R]
        if tanh_embed:
            e = np.exp2((2 * LOG2E * x).astype(f32)).astype(f32)
            x = (f32(1) - f32(2) / (e + f32(1))).astype(f32)
        skip = np.zeros((B, S), f32)
        Xout = np.zeros((L, B, R), f32)
        Kout = np.zeros((L, B, S), f32)
        bsum = np.zeros(S, f32)
        for l in range(L):
            xh = f16(x)
            dl = dil[l]
            xp = hist[l].get(n - dl)
            if xp is None:
                xp = np.zeros((B, R), f32)
            hist[l][n] = xh
            hist[l].pop(n - dl, None)
            cond = f16(t.Lh[n, l] * pres[None, :])                      # [B][2R]
            a = (Bh[l][None, :] + cond).astype(f32)
            a = (a + xp @ Wprev[l]).astype(f32)
            a = (a + xh @ Wcur[l]).astype(f32)
            ea = np.exp2(a[:, :R]).astype(f32)
            eb = np.exp2(a[:, R:]).astype(f32)
            h = ((f32(1) - f32(2) / (ea + f32(1))) * (f32(1) / (f32(1) + eb))).astype(f32)
            hh = f16(h)
            x = ((t.Bres[l][None, :] + x).astype(f32) + hh @ Wres[l]).astype(f32)
            skip = (skip + hh @ Wskip[l]).astype(f32)
            bsum = (bsum + t.Bskip[l]).astype(f32)
            Xout[l] = x
            Kout[l] = skip + bsum[None, :]
        sk = np.maximum(skip + bsum[None, :], 0).astype(f32)
        Kout[L - 1] = sk
        zs = np.maximum(t.Bzs[None, :] + f16(sk) @ Wzs, 0).astype(f32)
        za = (t.Bza[None, :] + f16(zs) @ Wza).astype(f32)
        m = np.maximum(za.max(axis=1, keepdims=True), 0)
        ex = np.exp((za - m).astype(f32)).astype(f32)
        p = (ex / ex.sum(axis=1, keepdims=True, dtype=f32)).astype(f32)
        cum = np.cumsum(p, axis=1, dtype=f32)
        for b in range(B):
            s = t.sel[n, b]
            k = int(np.argmax(s < cum[b])) if (s < cum[b]).any() else -1
            y[b, n] = k
            lo[b, n] = cum[b, k - 1] if k > 0 else 0.0
            hi[b, n] = cum[b, k] if k >= 0 else cum[b, -1]
        yP = yC
        yC = np.asarray(y_forced[:, n], np.int64)
        out = dict(Xout=Xout, skipOut=Kout, Zs=zs, Za=za, P=p)
    out.update(y=y, lo=lo, hi=hi)
    return out
