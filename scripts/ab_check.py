#!/usr/bin/env python3
"""checksum of the fp16 samples of a small O(1) run per organisation (library from NVW_LIB): experiment builds must reproduce the shipped one's"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, util
import test_parity_gpu as T
case = cases.Case("ab", 33, [], cases.Shape(64, 256, 256, 20, 48, 600, 512), 3, 1, 600)
t = util.gen_o1(case, half=True)
out = []
for mode in (sys.argv[1:] or ["wg3"]):
    e = T._engine_o1(case, t, 16, mode)
    y = np.full((48, 600), -1, dtype=np.int32)
    assert e.run(600, 48, y, 1, False)
    e.synchronize()
    e.close()
    out.append("%s %08x" % (mode, zlib.crc32(y.tobytes())))
print("samples crc:", "  ".join(out))
