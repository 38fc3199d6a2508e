#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own CPU implementation.

Runs only in the authoring container (needs /root/reference to build oracle/_ref):
    python tests/golden/make_golden.py

For every case in tests/cases.py the inputs are produced by the reference's Matrix::randomize
under the case's srand() seed (oracle/ref_driver.cpp), the reference's nvWavenetReference is
run for the case's iterations, and we store
    yOut   [iters][B][N] int32   -- exact sample indices
    Za, P  [B][A] float32        -- logits / probabilities of the last sample, last iteration
    crc_*                        -- CRC-32 of the fp32 bytes of every getter output
                                    (Xout, skipOut, Zs, Za, P) per iteration, and of the inputs
The GPU-box tests regenerate the inputs from the seed with oracle/liboracle.so and check the
input CRC before trusting anything else.

Long cases (N > 64) exceed what the reference class can hold comfortably (it keeps every
sample's activations); they are generated with oracle/liboracle.so *after* it has been pinned
bit-exact against the reference on all short cases in this same script.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
import cases  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run_case(case, which):
    s = case.shape
    t = O.gen_test_inputs(case.seed, case.prior, s, which)
    cls = O.RefOracle if which == "ref" else O.Oracle
    eng = cls(s.L, s.B, s.N, s.R, s.S, s.A, s.maxD)
    eng.set_model(t)
    eng.set_inputs(t.Lh, t.sel)
    ys, crcs, g = [], [], None
    for _ in range(case.iters):
        ys.append(eng.run(s.N))
        g = eng.getters()
        crcs.append([O.crc32(g[k]) for k in ("Xout", "skipOut", "Zs", "Za", "P")])
    eng.close()
    return dict(yOut=np.stack(ys), Za=g["Za"], P=g["P"], crc_act=np.array(crcs, dtype=np.uint32),
                crc_inputs=np.array([t.crc()], dtype=np.uint32))


def main():
    O.build(force=True)
    assert O.have_ref(), "oracle/_ref missing: run where /root/reference exists"
    for case in cases.ALL_CASES:
        s = case.shape
        short = s.N <= 64
        got = run_case(case, "ref" if short else "oracle")
        if short:
            mine = run_case(case, "oracle")
            for k in got:
                assert np.array_equal(got[k], mine[k]), (case.name, k)
        got["generated_by"] = np.array(["reference" if short else "oracle(pinned)"])
        np.savez_compressed(os.path.join(OUT, case.name + ".npz"), **got)
        print("%-36s %-16s y[0,0,:4]=%s" % (case.name, got["generated_by"][0], got["yOut"][0, 0, :4]))


if __name__ == "__main__":
    main()
