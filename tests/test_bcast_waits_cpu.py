"""wn::wavenet_bcast waits for its weight copies (LDS-DMA) and its conditioning / tap loads with hand-placed `s_waitcnt vmcnt(N)`
whose counts come from a constexpr model of what every chunk boundary issues (wn_bcast.hpp: BCfg::opsAt / waitAt / kWaitUse).
A count that is too LARGE is a silent race (a fragment read before its copy has landed); one that is too small stalls the wave
on loads it does not need yet.  tests/cpp/bcast_waits.hip prints the model's tables as the kernel's templates compute them;
this test replays a wave's in-order vector-memory queue over three samples of a model and checks, at every wait,

  * safety:     the count is never larger than the number of operations issued behind the youngest operation waited for;
  * exactness:  inside the steady state of the generic layers it is equal to it (no wait for anything it does not need)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def model_tables(tmp_path, R, S, A, fp16):
    exe = str(tmp_path / ("bcast_waits_%d_%d_%d" % (R, S, A)))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-unused-result", "-w",
                           "-I" + os.path.join(ROOT, "nv_wavenet_amd", "csrc"), "-DWN_R=%d" % R, "-DWN_S=%d" % S, "-DWN_A=%d" % A,
                           os.path.join(ROOT, "tests", "cpp", "bcast_waits.hip"), "-o", exe], timeout=600)
    out = subprocess.run([exe, "1" if fp16 else "0"], capture_output=True, text=True, check=True).stdout.splitlines()
    if out[0] != "supported 1":
        return None, None
    toks = out[1].split()[1:]
    const = {toks[i]: int(toks[i + 1]) for i in range(0, len(toks), 2)}
    table = {}
    for ln in out[2:]:
        _, part, bp, _, ops, _, wait = ln.split()
        table[(part, int(bp))] = (int(ops), int(wait))
    return const, table


def replay(c, table, L, samples=3):
    """The queue of one wave: every consumed position issues one copy (the stream NSLOT - CH positions on); every CH-th position
    ends a chunk: wait + barrier, then the boundary's loads.  Returns (violations, inexact generic waits, checks)."""
    CH, NSLOT, FLW, FHWP = c["CH"], c["NSLOT"], c["FLW"], c["FHWP"]
    T = FLW * L + FHWP                                  # positions of a sample's stream
    parts = [("L0", 0, c["P0_END"], 0)] + [("GEN", c["P_CUR"], c["P_CUR"] + FLW, l) for l in range(1, L)] + \
            [("TAIL", c["P_CUR"], FLW, None), ("HEAD", 0, FHWP, None)]
    ops = []                                            # ("piece", stream index of its data) | ("load", (sample, layer) it belongs to)
    piece_at = {}                                       # stream index of the data -> index in ops (the first turn is in the prologue: -1)
    youngest_load = {}                                  # (sample, layer) of a request -> index in ops of its youngest load
    bad, inexact, checks = [], [], 0
    g = 0                                               # stream positions consumed so far (all samples)
    for s in range(samples):
        for part, first, last, layer in parts:
            for pos in range(first, last):
                data = g + NSLOT - CH
                piece_at[data] = len(ops)
                ops.append(("piece", data))
                g += 1
                if (pos + 1) % CH:
                    continue
                n_ops, wait = table[(part, pos + 1)]
                k = g // CH - 1                         # the chunk just consumed; chunk k + 2 must have landed
                need = [piece_at[d] for d in range((k + 2) * CH, (k + 3) * CH) if d in piece_at]
                assert all(d < NSLOT or d in piece_at for d in range((k + 2) * CH, (k + 3) * CH)), "a chunk is needed before its copies are issued"
                younger = len(ops) - 1 - max(need) if need else 1 << 30      # (the ring's first turn is copied and awaited in the prologue)
                checks += 1
                if wait > younger:
                    bad.append(("boundary", s, part, layer, pos + 1, wait, younger))
                elif wait < younger and part == "GEN" and layer >= 2 and s >= 1:
                    inexact.append((s, layer, pos + 1, wait, younger))
                # the boundary's loads: the request of layer l + 2 (layer 0: from REQ_HEAD on; the head's last boundary: layer 0's
                # request of the next sample, i.e. for its layer 2)
                if n_ops:
                    if part == "HEAD":
                        tag = (s + 1, 2)
                    else:
                        tag = (s, layer + 2) if layer + 2 < L else (s + 1, layer + 2 - L)
                    for _ in range(n_ops):
                        youngest_load[tag] = len(ops)
                        ops.append(("load", tag))
                # conditioning / taps of layer l + 1 are used behind the boundary in front of this layer's tap GEMM
                if part in ("L0", "GEN") and pos + 1 == (c["P0_PREV"] if part == "L0" else c["P_PREV"]):
                    tag = (s, layer + 1) if layer + 1 < L else (s + 1, 0)
                    if tag in youngest_load:            # (the first sample's layers 0 and 1 are requested and awaited in the prologue)
                        younger = len(ops) - 1 - youngest_load[tag]
                        checks += 1
                        if c["kWaitUse"] > younger:
                            bad.append(("use", s, part, layer, pos + 1, c["kWaitUse"], younger))
                    else:
                        assert s == 0 and layer == 0, (s, layer)
    # every request was issued in full
    per_tag = {}
    for kind, tag in ops:
        if kind == "load":
            per_tag[tag] = per_tag.get(tag, 0) + 1
    short = {t: n for t, n in per_tag.items() if n != c["REQ_LOADS"] and t[0] < samples and not (t[0] == 0 and t[1] < 3)}
    return bad, inexact, checks, short


@pytest.mark.parametrize("R,S,A,fp16,L", [(64, 256, 256, True, 20), (64, 256, 256, True, 7), (64, 256, 256, False, 20), (64, 128, 256, True, 20),
                                          (64, 128, 256, True, 3)])
def test_hand_placed_waits_are_safe_and_exact(tmp_path, R, S, A, fp16, L):
    c, table = model_tables(tmp_path, R, S, A, fp16)
    if c is None:
        pytest.skip("wavenet_bcast does not exist for this shape")
    bad, inexact, checks, short = replay(c, table, L)
    assert checks > 100
    assert not bad, "a wait allows more operations in flight than were issued behind what it waits for: %s" % bad[:5]
    assert not inexact, "a generic layer's boundary waits for more than it needs: %s" % inexact[:5]
    assert not short, "requests issued with another number of loads than REQ_LOADS: %s" % short


def test_the_replay_catches_a_count_that_is_too_large(tmp_path):
    """negative control: one more operation allowed in flight at a generic boundary must be reported"""
    c, table = model_tables(tmp_path, 64, 256, 256, True)
    key = ("GEN", (c["P_CUR"] // c["CH"] + 3) * c["CH"])
    ops, wait = table[key]
    table[key] = (ops, wait + 1)
    bad, _, _, _ = replay(c, table, 20)
    assert bad and all(b[0] == "boundary" and b[2] == "GEN" for b in bad)
    c2 = dict(c)
    c2["kWaitUse"] = c["kWaitUse"] + 40
    table[key] = (ops, wait)
    bad, _, _, _ = replay(c2, table, 20)
    assert bad and all(b[0] == "use" for b in bad)
