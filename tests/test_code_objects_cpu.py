"""Static properties of the shipped gfx950 code objects (no GPU): scripts/isa_stats.py reads registers and scratch from the ELF
notes of every kernel.  Production kernels (the ones launches without the activation dump run) must not touch scratch memory:
a spilled register in a hot loop is a vector-memory operation the hand-tuned schedules do not know about -- the two-tile
wavenet_bcast variant of round 4 lost 15 % to exactly that before it was removed -- and must fit the register file."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "nv_wavenet_amd", "csrc", "build")


def kernel_table(obj):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_stats.py"), "regs", obj], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = []
    for ln in out.stdout.splitlines()[1:]:
        parts = ln.split()
        if len(parts) < 6:
            continue
        name = " ".join(parts[:-5])
        vgpr, agpr, sgpr, scratch, spill = (int(x) for x in parts[-5:])
        rows.append((name, vgpr, agpr, sgpr, scratch, spill))
    return rows


# production kernels (DUMP = false) that DO spill today, all outside the BASELINE configurations' launches: the wider shapes of
# SURVEY.md 8f rank 4 (R = 32 / A = 512 / A = 1024 chains, R = 256), correct but untuned.  Listed so that the list can only shrink
# (round 5: the two-tile instantiations of R >= 128 -- 15 to 560 spilled registers -- are no longer compiled).
KNOWN_SPILLS = {
    "wn::wavenet_chain<true, 128, 256, 1024, false, false>", "wn::wavenet_chain<true, 32, 128, 256, false, false>",
    "wn::wavenet_chain<true, 32, 256, 256, false, false>", "wn::wavenet_chain<true, 64, 128, 512, false, false>",
    "wn::wavenet_wg<true, 256, 256, 256, 1, false, false, 1, false>", "wn::wavenet_wg<true, 256, 256, 256, 1, true, false, 1, false>",
    "wn::wavenet_wg<true, 256, 256, 256, 1, false, false, 1, true>", "wn::wavenet_wg<true, 256, 256, 256, 1, true, false, 1, true>",
}
STRICT = ("inst_64_128_256_p16.o", "inst_64_256_256_p16.o")          # BASELINE C2, C3 / C5 (the headline)


@pytest.mark.parametrize("obj", sorted(os.path.basename(p) for p in glob.glob(os.path.join(BUILD, "inst_*_p16.o"))) or ["(library not built)"])
def test_production_kernels_of_the_fp16_engines_use_no_scratch(obj):
    if not os.path.exists(os.path.join(BUILD, obj)):
        pytest.skip("build the library first (__graft_entry__.build())")
    rows = kernel_table(obj)
    assert len(rows) >= 10, rows
    gen = [r for r in rows if "wavenet_wg<" in r[0] or "wavenet_chain<" in r[0]]
    assert gen, "no generation kernel in " + obj
    for name, vgpr, agpr, sgpr, scratch, spill in gen:
        # wavenet_wg<F16, R, S, A, BT, EMBLDS, DUMP, RAW, LR>: DUMP is the seventh argument; the fifth of wavenet_chain<F16, R, S, A, DUMP, HOIST>
        flags = name[name.index("<") + 1:name.rindex(">")].replace(" ", "").split(",")
        dump = flags[6] if "wavenet_wg<" in name else flags[4]
        assert dump in ("true", "false"), name
        assert vgpr <= 512 and agpr <= 256 and sgpr <= 106, (name, vgpr, agpr, sgpr)
        if dump == "false" and (scratch or spill):
            assert obj not in STRICT and name in KNOWN_SPILLS, "%s: scratch %d B per lane, %d spilled registers" % (name, scratch, spill)


def test_the_conditioning_producer_uses_no_scratch_and_two_waves_fit_a_simd():
    obj = os.path.join(BUILD, "cond_producer.o")
    if not os.path.exists(obj):
        pytest.skip("build the library first (__graft_entry__.build())")
    rows = [r for r in kernel_table(obj) if "cond_producer_kernel" in r[0]]
    assert len(rows) == 4, rows
    for name, vgpr, agpr, sgpr, scratch, spill in rows:
        assert scratch == 0 and spill == 0, name
    three = [r for r in rows if "<3>" in r[0].replace(" ", "") or "ILi3E" in r[0]]          # (anonymous-namespace names stay mangled)
    assert three and three[0][1] + three[0][2] <= 256, three          # n_cond = 80: the shape bench.py runs


def test_round6_instantiations_exist_and_use_no_scratch():
    """The kernels round 6 added are in the shipped code objects and spill nothing: four tiles per workgroup (dump-free, packed
    conditioning) with and without ring slots in LDS, the LR variants of the three-tile kernels for every conditioning path of the fp16
    engine, the dump-free fp32 kernels behind wavenet_infer(), and the chain instantiation that requests a unit's conditioning up
    front where stages hold two layers (R = 128)."""
    if not os.path.exists(os.path.join(BUILD, "inst_64_256_256_p16.o")):
        pytest.skip("build the library first (__graft_entry__.build())")
    rows = {r[0].replace(" ", ""): r for r in kernel_table("inst_64_256_256_p16.o")}
    want = ["wn::wavenet_wg<true,64,256,256,4,%s,false,0,%s>" % (emb, lr) for emb in ("true", "false") for lr in ("true", "false")]
    want += ["wn::wavenet_wg<true,64,256,256,3,true,false,%d,true>" % raw for raw in (0, 1, 2, 3)]
    for name in want:
        assert name in rows, name
        assert rows[name][4] == 0 and rows[name][5] == 0, rows[name]
    assert not any(k.startswith("wn::wavenet_wg<true,64,256,256,4,") and ",false,3," in k for k in rows), "no four-tile kernel computes the conditioning itself"
    rows32 = {r[0].replace(" ", ""): r for r in kernel_table("inst_64_256_256_p32.o")}
    for name in ("wn::wavenet_wg<false,64,256,256,1,true,false,0,false>", "wn::wavenet_wg<false,64,256,256,1,true,false,0,true>",
                 "wn::wavenet_chain<false,64,256,256,false,false>"):
        assert name in rows32, name
    rows4 = {r[0].replace(" ", ""): r for r in kernel_table("inst_128_256_256_p16.o")}
    hoist = "wn::wavenet_chain<true,128,256,256,false,true>"
    assert hoist in rows4 and rows4[hoist][4] == 0 and rows4[hoist][5] == 0, rows4.get(hoist)
    assert "wn::wavenet_chain<true,64,256,256,false,true>" not in rows, "five-layer stages spill with the hoisted conditioning: not built"
