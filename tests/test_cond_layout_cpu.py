"""The engine's conditioning fragment order, as the Python side states it (nv_wavenet_amd/nv_wavenet.py: cond_fragment_order,
pack_cond_input, get_cond_input(layout="packed")), against a loop-by-loop transcription of what wn::pack_cond_tiled_kernel
writes (nv_wavenet_amd/csrc/wn_kernels.hpp: "[rows][tiles][wave][COND_FR][lane][EPL]; fragment c, element e of wave w: gate
slot it = c*TPF + (e>>2) -> tile = w + NW*(it>>1) + (it&1)*RT").  No GPU: the GPU tests hold the engine to the oracle on
buffers built by these functions; this file pins the functions themselves for every wave / tile split the shapes produce."""
import numpy as np
import pytest
import torch

from nv_wavenet_amd.nv_wavenet import cond_fragment_order, get_cond_input, pack_cond_input


def _kernel_order(Lh, R, precision, tiles):
    """numpy transcription of pack_cond_tiled_kernel's destination indexing (fp32 arithmetic, one rounding to T_data)"""
    N, L, B, _ = Lh.shape
    RT = R // 16
    NW = 4 if RT >= 4 else RT
    TPF, EPL = (2, 8) if precision == 16 else (1, 4)
    CF = 2 * (RT // NW) // TPF
    out = np.zeros((N + 1, L, tiles, NW, CF, 64, EPL), dtype=np.float32)
    for w in range(NW):
        for c in range(CF):
            for lane in range(64):
                g, j = lane >> 4, lane & 15
                for e in range(EPL):
                    it = c * TPF + (e >> 2)
                    ch = (w + NW * (it >> 1) + (it & 1) * RT) * 16 + g * 4 + (e & 3)
                    sc = 1.0 if precision == 32 else (-1.44269504088896340736 if ch >= R else 2.88539008177792681472)
                    for tile in range(tiles):
                        b = tile * 16 + j
                        if b < B:
                            out[:N, :, tile, w, c, lane, e] = Lh[:, :, b, ch] * np.float32(sc)
    return out


@pytest.mark.parametrize("R", [32, 64, 128, 256])
@pytest.mark.parametrize("precision", [16, 32])
def test_fragment_order_is_the_pack_kernels(R, precision):
    perm, scale = cond_fragment_order(R, precision)
    assert sorted(perm) == list(range(2 * R)) and len(scale) == 2 * R
    rng = np.random.default_rng(R + precision)
    N, L, B, tiles = 3, 2, 19, 3                     # ragged batch, one whole padding tile
    Lh = rng.standard_normal((N, L, B, 2 * R)).astype(np.float32)
    want = _kernel_order(Lh, R, precision, tiles)
    got = pack_cond_input(torch.from_numpy(Lh), precision, tiles)
    assert got.dtype == (torch.float16 if precision == 16 else torch.float32) and got.is_contiguous()
    assert tuple(got.shape[:3]) == (N + 1, L, tiles) and got.numel() == want.size
    got = got.float().numpy().reshape(want.shape)
    if precision == 16:
        want = want.astype(np.float16).astype(np.float32)
    assert np.array_equal(got, want)
    assert not got[N].any(), "the padding sample is zero"


def test_get_cond_input_emits_the_fragment_order_of_its_own_output():
    """layout="packed" = pack_cond_input of layout="NLBC", with the permutation and the pre-scale folded into the 1x1
    convolution's weights: equal up to the fp32 rounding of (s w) x against s (w x) in front of the cast to fp16."""
    g = torch.Generator().manual_seed(5)
    R, L, B, frames, stride, n_cond, tiles = 64, 3, 5, 4, 4, 10, 2
    rnd = lambda *s, sc=1.0: (torch.rand(*s, generator=g) - 0.5) * sc
    feats = rnd(B, n_cond, frames)
    up_w, up_b = rnd(n_cond, n_cond, 2 * stride, sc=0.5), rnd(n_cond, sc=0.1)
    cw, cb = rnd(2 * R * L, n_cond, 1, sc=2.0), rnd(2 * R * L, sc=0.5)
    nlbc = get_cond_input(feats, up_w, up_b, stride, cw, cb, L, layout="NLBC")
    for precision in (16, 32):
        a = get_cond_input(feats, up_w, up_b, stride, cw, cb, L, layout="packed", precision=precision, tiles=tiles)
        b = pack_cond_input(nlbc, precision, tiles)
        assert a.shape == b.shape and a.dtype == b.dtype
        tol = (2.0 ** -9 if precision == 16 else 1e-5) * float(b.float().abs().max())
        assert float((a.float() - b.float()).abs().max()) <= tol


@pytest.mark.parametrize("n_cond,R,L,B,frames,win,stride", [(80, 64, 3, 32, 1, 1024, 16), (40, 64, 2, 16, 3, 16, 4), (100, 32, 2, 16, 2, 8, 4)])
def test_fused_producer_operands_reproduce_the_packed_layout(n_cond, R, L, B, frames, win, stride):
    """csrc/cond_producer.hip computes the conditioning convolution with MFMAs and stores each lane's result quads as the packed
    layout's 16 bytes.  Its operand arrangement (nv_wavenet.py: cond_producer_weights) and its addressing, transcribed here with
    the MFMA's lane roles (A: lane (g, i) = row i, k-slice g; B: lane (g, j) = column j, k-slice g; D: lane (g, j) = rows
    4g..4g+3 of column j), must give what get_cond_input(layout="packed") gives (fp16 rounding apart); the GPU suite runs the
    kernel itself against the same reference."""
    from nv_wavenet_amd.nv_wavenet import _upsample_trimmed_gemm, cond_producer_weights
    g = torch.Generator().manual_seed(7)
    rnd = lambda *s, sc=1.0: (torch.rand(*s, generator=g) - 0.5) * sc
    up_w, up_b = rnd(n_cond, n_cond, win, sc=0.2), rnd(n_cond, sc=0.2)
    cw, cb = rnd(2 * R * L, n_cond, 1, sc=0.8), rnd(2 * R * L)
    f = rnd(B, n_cond, frames, sc=2.0)
    tiles, N = B // 16, frames * stride
    NWF = 2 * R // 32
    ref = torch.zeros(N, L, tiles, NWF, 4, 16, 8, dtype=torch.float16)
    get_cond_input(f, up_w, up_b, stride, cw, cb, L, layout="packed", precision=16, tiles=tiles, out=ref, via_gemm=False, fused=False)
    wfrag, bpos, KF, nwf = cond_producer_weights(cw, cb, L, 16)
    assert nwf == NWF and KF == (n_cond + 31) // 32 and wfrag.shape == (L, NWF, 2, KF, 64, 8) and wfrag.dtype == torch.float16
    x = torch.nn.functional.pad(_upsample_trimmed_gemm(f, up_w, up_b, stride), (0, 32 * KF - n_cond)).half().float()      # [B][N][32 KF]
    A = wfrag.float().reshape(L, NWF, 2, KF, 4, 16, 8)                              # [l][wf][tt][kf][g][i][e]
    xb = x.reshape(tiles, 16, N, KF, 4, 8)                                          # [tile][j][n][kf][g][e]
    D = torch.einsum("lwtkgie,cjnkge->nlcwtij", A, xb)                              # rows i, columns j of result tile tt
    out = torch.zeros(N, L, tiles, NWF, 4, 16, 8)
    for gq in range(4):
        for tt in range(2):
            for r in range(4):
                bias = bpos.reshape(L, NWF, 4, 8)[:, :, gq, tt * 4 + r]             # position (wf*4 + g)*8 + tt*4 + r
                out[:, :, :, :, gq, :, tt * 4 + r] = D[:, :, :, :, tt, 4 * gq + r, :] + bias[None, :, None, :, None]
    tol = 2.0 ** -9 * float(ref.float().abs().max())
    assert float((out - ref.float()).abs().max()) <= tol
