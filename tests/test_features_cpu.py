"""CPU side of the in-kernel conditioning (round 5): the fixtures recorded from the REFERENCE's own WaveNet.get_cond_input
(tests/golden/cond_*.npz, made by tests/golden/make_golden_cond.py) pin this repo's evaluations of the same convolutions; the
feature fragment order is transcribed index by index; the reference module itself is the live comparand where its tree exists."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

import cases
import condgen
import util

REF_PY = "/root/reference/pytorch"


def _tensors(cc):
    s = cases.BY_NAME[cc.case_name].shape
    m = condgen.make_cond_model(cc, s)
    return s, m, {k: torch.from_numpy(v) for k, v in m.items()}


@pytest.mark.parametrize("cc", condgen.COND_CASES, ids=lambda c: c.name)
def test_get_cond_input_of_this_repo_matches_the_reference_modules_record(cc):
    """nv_wavenet.get_cond_input (torch convolutions, and the matrix-product form the GPU runs) and upsample_features + a plain
    matrix product (what the generation kernel computes) against the record of the reference module's output."""
    from nv_wavenet_amd import nv_wavenet as NW
    s, m, tt = _tensors(cc)
    rec = util.load_golden(cc.name)
    for via_gemm in (False, True):
        ci = NW.get_cond_input(tt["features"], tt["up_w"], tt["up_b"], cc.stride, tt["cond_w"], tt["cond_b"], s.L, layout="CBLN", via_gemm=via_gemm)
        condgen.check_against_record(ci.contiguous().numpy(), rec, "%s via_gemm=%s" % (cc.name, via_gemm))
    x = NW.upsample_features(tt["features"], tt["up_w"], tt["up_b"], cc.stride)                       # [B][n_cond][N]
    assert x.shape == (s.B, cc.n_cond, s.N)
    lh = torch.einsum("oc,bct->bot", tt["cond_w"][:, :, 0], x) + tt["cond_b"][None, :, None]          # Wcond c + bcond
    condgen.check_against_record(lh.view(s.B, s.L, 2 * s.R, s.N).permute(2, 0, 1, 3).contiguous().numpy(), rec, cc.name + " W c + b")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_PY, "wavenet.py")), reason="no reference tree on this machine")
def test_the_reference_module_itself_live():
    """The reference's WaveNet class, imported: get_cond_input of the module == this repo's get_cond_input on the CPU, bit for bit
    (the same torch operations), and == the committed record (the fixture is current)."""
    from nv_wavenet_amd import nv_wavenet as NW
    sys.path.insert(0, REF_PY)
    try:
        import wavenet as ref_wavenet
    finally:
        sys.path.remove(REF_PY)
    cc = condgen.COND_BY_NAME["cond_oddL_B19"]
    s, m, tt = _tensors(cc)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = ref_wavenet.WaveNet(s.A, s.L, s.maxD, s.R, s.S, s.A, cc.n_cond, cc.window, cc.stride)
    with torch.no_grad():
        net.upsample.weight.copy_(tt["up_w"]), net.upsample.bias.copy_(tt["up_b"])
        net.cond_layers.conv.weight.copy_(tt["cond_w"]), net.cond_layers.conv.bias.copy_(tt["cond_b"])
        ref = net.get_cond_input(tt["features"])
    mine = NW.get_cond_input(tt["features"], tt["up_w"], tt["up_b"], cc.stride, tt["cond_w"], tt["cond_b"], s.L, layout="CBLN", via_gemm=False)
    assert torch.equal(ref, mine)
    condgen.check_against_record(ref.contiguous().numpy(), util.load_golden(cc.name), "live reference module")


@pytest.mark.parametrize("precision", [16, 32])
def test_feature_fragment_order_index_by_index(precision):
    """nv_wavenet.feature_fragments against the definition in include/nv_wavenet_c.h (and pack_features_kernel): fragment kf, lane
    (g, j), element e holds channel (kf*TPF + (e>>2))*16 + 4g + (e&3) of utterance tile*16 + j; zero beyond n_cond / the batch."""
    from nv_wavenet_amd import nv_wavenet as NW
    B, Cn, N, tiles = 21, 37, 5, 3
    TPF, EPL = (2, 8) if precision == 16 else (1, 4)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cn, N, generator=g)
    f = NW.feature_fragments(x, tiles, precision)
    KFC = f.shape[2]
    assert f.shape == (N, tiles, KFC, 4, 16, EPL) and KFC == (80 + 16 * TPF - 1) // (16 * TPF)
    xd = x.to(f.dtype)
    for n in range(N):
        for tile in range(tiles):
            for kf in range(KFC):
                for gg in range(4):
                    for j in range(16):
                        for e in range(EPL):
                            c = (kf * TPF + (e >> 2)) * 16 + 4 * gg + (e & 3)
                            b = tile * 16 + j
                            want = xd[b, c, n] if (b < B and c < Cn) else 0
                            assert f[n, tile, kf, gg, j, e] == want
