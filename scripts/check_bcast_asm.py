#!/usr/bin/env python3
"""Static check of wn::wavenet_bcast's hand-managed vector memory (no GPU needed).

The kernel issues its conditioning / dilated-tap loads from inline assembly and waits for them by hand (wn_bcast.hpp), so
the compiler believes the destination registers hold their value from the load statement on.  If it ever copied or read one
of them between the load and the hand-placed s_waitcnt that covers it (a live-range split, a copy on a loop edge), the kernel
would compute on registers the load has not written yet.  This script compiles the C3 fp16 instantiation to assembly and, for
every such load of every wavenet_bcast kernel, walks the instruction stream in layout order up to the next covering wait
(s_waitcnt vmcnt(N) with N <= kWaitUse of the build, taken from the listing) and reports any instruction in between that
touches the load's destination registers.  Exit status 1 if there is one.

usage: check_bcast_asm.py [file.s]      (without an argument: compiles engine_inst.hip for 64/256/256 fp16 into /tmp)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(tok):
    """'a[4:7]' -> {('a',4),..}; 'v12' -> {('v',12)}"""
    m = re.fullmatch(r"([av])\[(0x[0-9a-f]+|\d+):(0x[0-9a-f]+|\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2), 0), int(m.group(3), 0) + 1)}
    m = re.fullmatch(r"([av])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def all_regs(line):
    out = set()
    for tok in re.findall(r"\b[av]\[(?:0x[0-9a-f]+|\d+):(?:0x[0-9a-f]+|\d+)\]|\b[av]\d+\b", line):
        out |= regs_of(tok)
    return out


VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)(load|store|atomic)")


def check(body, name):
    lines = [l.strip() for l in body.split("\n")]
    insts = [(i, l) for i, l in enumerate(lines) if l and not l.startswith((";", ".", "_")) and not l.endswith(":")]
    bad = 0
    nloads = 0
    for k, (i, l) in enumerate(insts):
        m = re.match(r"buffer_load_dwordx4 ([av]\[(?:0x[0-9a-f]+|\d+):(?:0x[0-9a-f]+|\d+)\]), .* offen.* nt$", l)
        if not m or " lds" in l:
            continue
        nloads += 1
        dst = regs_of(m.group(1))
        younger = 0            # vector-memory operations issued behind the load so far (layout order)
        for i2, l2 in insts[k + 1:]:
            mw = re.match(r"s_waitcnt vmcnt\((\d+)\)", l2)
            if mw:
                if younger >= int(mw.group(1)):      # all but the youngest N have completed: the load is among them
                    break
                continue
            if all_regs(l2) & dst:
                print("%s: line %d `%s` touches the destination of the load at line %d `%s` with only %d younger operations "
                      "and no covering wait in between" % (name, i2 + 1, l2, i + 1, l, younger))
                bad += 1
                break
            if VMEM.match(l2):
                younger += 1
            if l2.startswith("s_endpgm"):
                break
    return nloads, bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = "/tmp/check_bcast.s"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-mllvm",
                               "-amdgpu-mfma-vgpr-form", "-DWN_R=64", "-DWN_S=256", "-DWN_A=256", "-DWN_P=16", "-S", "--cuda-device-only",
                               os.path.join(ROOT, "nv_wavenet_amd", "csrc", "engine_inst.hip"), "-o", path], stderr=subprocess.DEVNULL)
    USE_WAIT = int(os.environ.get("WN_USE_WAIT", "30"))
    s = open(path).read()
    total_bad = 0
    for m in re.finditer(r"^(_ZN2wn13wavenet_bcast\w+):.*?^\.Lfunc_end\d+:", s, re.S | re.M):
        n, bad = check(m.group(0), m.group(1)[20:60])
        print("%s: %d assembly-issued register loads checked, %d violations" % (m.group(1)[20:60], n, bad))
        total_bad += bad
    sys.exit(1 if total_bad else 0)
