"""Inputs for the feature-conditioning tests: a `cond_layers` + `upsample` pair and mel-like features from a seed (numpy PCG64:
the same numbers on every machine).  tests/golden/make_golden_cond.py loads exactly these tensors into the REFERENCE's own
WaveNet module (pytorch/wavenet.py) and records what its get_cond_input returns; the tests rebuild the tensors from the seed and
hold this repo's evaluation to that record before the oracle is fed with it."""
from collections import namedtuple

import numpy as np

CondCase = namedtuple("CondCase", "name seed case_name n_cond window stride")

# case_name: the tests/cases.py shape the conditioning belongs to (its L, B, N, R); N must be a multiple of stride
COND_CASES = [
    CondCase("cond_C3_B16", 501, "C3_R64S256A256_L20_B16", 80, 8, 2),                # the headline shape
    CondCase("cond_C3_B21_n37", 502, "C3_R64S256A256_L20_B21", 37, 12, 4),           # ragged batch, fewer channels than the kernels' 80
    CondCase("cond_oddL_B19", 503, "R64S128A256_L7_B19_oddL", 80, 10, 5),            # odd layer count
    CondCase("cond_C4_B8", 504, "C4_R128S256A256_L30_B8", 80, 8, 4),                 # R = 128
    CondCase("cond_C1_B1", 505, "C1_R32S128A256_L8_B1", 80, 16, 8),                  # R = 32 (two waves per workgroup)
]
COND_BY_NAME = {c.name: c for c in COND_CASES}


def make_cond_model(cc, shape):
    """features [B][n_cond][frames], upsample weight [n_cond][n_cond][window] + bias, cond weight [2R*L][n_cond][1] + bias; fp32.
    Magnitudes: upsampled features of order one, conditioning of standard deviation ~0.5 (the O(1) recipe's, tests/util.py)."""
    assert shape.N % cc.stride == 0 and cc.window % cc.stride == 0
    rng = np.random.Generator(np.random.PCG64(cc.seed))
    frames = shape.N // cc.stride
    taps = cc.window // cc.stride
    u = lambda size, std: (rng.random(size, dtype=np.float64) * 2.0 - 1.0).astype(np.float32) * np.float32(std * np.sqrt(3.0))
    return dict(
        features=rng.standard_normal((shape.B, cc.n_cond, frames)).astype(np.float32),
        up_w=u((cc.n_cond, cc.n_cond, cc.window), 1.0 / np.sqrt(taps * cc.n_cond)),
        up_b=u((cc.n_cond,), 0.1),
        cond_w=u((2 * shape.R * shape.L, cc.n_cond, 1), 0.5 / np.sqrt(cc.n_cond)),
        cond_b=u((2 * shape.R * shape.L,), 0.1),
    )


def record_of(cond_input, n_pick=4096, seed=7):
    """What the fixture keeps of a conditioning tensor [2R][B][L][N]: its size, float64 sums, and n_pick (index, value) pairs."""
    flat = np.ascontiguousarray(cond_input, dtype=np.float32).reshape(-1)
    rng = np.random.Generator(np.random.PCG64(seed))
    idx = np.sort(rng.choice(flat.size, size=min(n_pick, flat.size), replace=False)).astype(np.int64)
    return dict(shape=np.array(cond_input.shape, dtype=np.int64), sum=np.array([flat.astype(np.float64).sum()]),
                abs_sum=np.array([np.abs(flat.astype(np.float64)).sum()]), idx=idx, val=flat[idx].copy())


def check_against_record(cond_input, rec, what=""):
    """cond_input [2R][B][L][N] fp32 against a fixture record: every sampled value within 2e-6 of the tensor's magnitude (another
    machine's convolution may sum in another order), the sums within 1e-6 relative."""
    flat = np.ascontiguousarray(cond_input, dtype=np.float32).reshape(-1)
    assert tuple(cond_input.shape) == tuple(int(v) for v in rec["shape"]), (what, cond_input.shape, rec["shape"])
    scale = float(rec["abs_sum"][0]) / flat.size
    err = np.abs(flat[rec["idx"]].astype(np.float64) - rec["val"].astype(np.float64)).max()
    assert err <= 2e-5 * scale, "%s: sampled conditioning values differ from the reference's get_cond_input by %.3g (scale %.3g)" % (what, err, scale)
    assert abs(np.abs(flat.astype(np.float64)).sum() - float(rec["abs_sum"][0])) <= 1e-6 * float(rec["abs_sum"][0]), what + ": |sum| differs"
    assert abs(flat.astype(np.float64).sum() - float(rec["sum"][0])) <= 1e-6 * float(rec["abs_sum"][0]), what + ": sum differs"
