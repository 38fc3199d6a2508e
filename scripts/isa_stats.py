#!/usr/bin/env python3
"""Static view of the shipped gfx950 code objects (no GPU needed).

  isa_stats.py regs  [inst]                 registers / scratch / LDS of every kernel of an instantiation object
  isa_stats.py mix   [inst] <kernel-substr> instruction mix of the kernel whose demangled name contains the substring,
                                            whole kernel and its largest loop body (the per-sample loop of wavenet_wg)
  isa_stats.py sha   [inst] <kernel-substr> sha256 of that kernel's disassembly (addresses stripped): the identity of the device code
                                            a measurement belongs to (profiles/traffic_rNN.json; bench.py drops a measurement taken on
                                            other code)

inst: name of an object under nv_wavenet_amd/csrc/build (default inst_64_256_256_p16.o).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def code_object(inst):
    obj = inst if os.path.exists(inst) else os.path.join(ROOT, "nv_wavenet_amd", "csrc", "build", inst)
    tmp = tempfile.mkdtemp(prefix="isa_")
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "gfx950.co")
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return co


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(wn::Params.*$", "", name)
    return name.replace("(bool)", "").replace("(int)", "")


def regs(inst):
    co = code_object(inst)
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    kernels = []
    cur = {}
    for line in notes.split("\n"):
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("agpr_count") is not None:
            kernels.append(cur)
            cur = {}
        if k in ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "name",
                 "vgpr_spill_count", "sgpr_spill_count"):
            cur[k] = v
        if k == "wavefront_size":
            kernels.append(cur)
            cur = {}
    kernels = [k for k in kernels if "name" in k]
    dm = demangle([k["name"] for k in kernels])
    print("%-100s %5s %5s %5s %7s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "spill"))
    for k in sorted(kernels, key=lambda k: dm[k["name"]]):
        print("%-100s %5s %5s %5s %7s %6s" % (short(dm[k["name"]])[:100], k.get("vgpr_count"), k.get("agpr_count"), k.get("sgpr_count"),
                                             k.get("private_segment_fixed_size"), k.get("vgpr_spill_count")))


CLASSES = [("mfma", r"^v_mfma"), ("v_pk_f32", r"^v_pk_(add|mul|fma)_f32"), ("v_pk_other", r"^v_pk_"), ("trans", r"^v_(exp|rcp|log|sqrt|rsq)_"),
           ("accvgpr", r"^v_accvgpr"), ("cndmask", r"^v_cndmask"), ("v_cvt", r"^v_cvt"), ("dpp/perm", r"_dpp|^v_perm|^v_readlane|^v_readfirstlane"),
           ("valu_other", r"^v_"), ("buffer_load", r"^buffer_load"), ("buffer_store", r"^buffer_store"), ("global/flat", r"^(global|flat)_"),
           ("ds_read", r"^ds_read|^ds_load"), ("ds_write", r"^ds_write|^ds_store"), ("ds_other", r"^ds_"), ("s_waitcnt", r"^s_waitcnt"),
           ("s_nop", r"^s_nop"), ("s_barrier", r"^s_barrier"), ("branch", r"^s_cbranch|^s_branch"), ("salu/smem", r"^s_")]


def classify(op):
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "other"


def mix(inst, sub):
    co = code_object(inst)
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", dis)
    syms = []
    for b in blocks:
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", b)
        if m:
            syms.append((m.group(1), b))
    dm = demangle([s for s, _ in syms])
    hits = [(s, b) for s, b in syms if sub in dm[s]]
    if not hits:
        sys.exit("no kernel matches %r" % sub)
    for s, body in hits:
        lines = []          # (address, opcode, operands, branch target or None)
        for ln in body.split("\n")[1:]:
            m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<[^>+]+\+0x([0-9a-f]+)>)?", ln)
            if m:
                lines.append((int(m.group(3), 16), m.group(1), m.group(2), int(m.group(4), 16) if m.group(4) else None))
        base = lines[0][0]
        addr_to_idx = {a: i for i, (a, _, _, _) in enumerate(lines)}
        loops = []          # (first index, last index) of every backward branch
        for i, (a, op, args, off) in enumerate(lines):
            if (op.startswith("s_cbranch") or op == "s_branch") and off is not None and base + off <= a and base + off in addr_to_idx:
                loops.append((addr_to_idx[base + off], i))
        print("== %s" % short(dm[s]))

        def report(title, lo, hi, excl=()):
            idx = [i for i in range(lo, hi + 1) if not any(a <= i <= b for a, b in excl)]
            cnt = collections.Counter(classify(lines[i][1]) for i in idx)
            tot = sum(cnt.values())
            valu = sum(v for k, v in cnt.items() if k in ("v_pk_f32", "v_pk_other", "trans", "accvgpr", "cndmask", "v_cvt", "dpp/perm", "valu_other"))
            print("  %s: %d instructions, VALU %d, MFMA %d, VALU:MFMA %.2f" % (title, tot, valu, cnt["mfma"], valu / max(1, cnt["mfma"])))
            print("   " + "  ".join("%s %d" % (k, cnt[k]) for k, _ in CLASSES if cnt[k]))
        report("whole kernel", 0, len(lines) - 1)
        big = sorted(loops, key=lambda l: l[0] - l[1])[:3]
        if big:
            outer = big[0]
            inner = [l for l in big[1:] if outer[0] <= l[0] and l[1] <= outer[1] and l[1] - l[0] > 200]
            report("per-sample loop (lines %d-%d)" % outer, outer[0], outer[1])
            for l in inner:
                report("  inner loop (lines %d-%d: layer pair)" % l, l[0], l[1])
            if inner:
                report("  per-sample loop outside the inner loop(s) (first layer, odd tail, head, softmax)", outer[0], outer[1], inner)


def kernel_sha(inst, sub):
    """sha256 over the instruction stream (mnemonics + operands, no addresses) of the ONE kernel whose demangled name contains `sub`;
    None when the object is missing or the match is not unique."""
    import hashlib
    try:
        co = code_object(inst)
        dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, timeout=600).stdout
    except Exception:
        return None
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", dis)
    syms = []
    for b in blocks:
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", b)
        if m:
            syms.append((m.group(1), b))
    dm = demangle([s_ for s_, _ in syms])
    hits = [b for s_, b in syms if sub in dm[s_].replace(" ", "")]
    if len(hits) != 1:
        return None
    text = "\n".join(re.sub(r"\s*//.*$", "", ln).strip() for ln in hits[0].split("\n")[1:])
    text = re.sub(r"<[^>]*>", "", text)              # (branch targets print as symbol+offset: keep the offset-free form)
    return hashlib.sha256(text.encode()).hexdigest()


def kernel_sub_of(kernel_info_name):
    """'wn::wavenet_wg<fp16,64,256,256,BT=3,EMBLDS=1,DUMP=0,RAW=0[,LR=1]>' (nvw_kernel_info) -> the demangled-name substring of that kernel"""
    m = re.match(r"wn::(\w+)<(fp16|fp32),(\d+),(\d+),(\d+),BT=(\d+),EMBLDS=(\d+),DUMP=(\d+),RAW=(\d+)(,LR=1)?>", kernel_info_name)
    if not m:
        return None
    k, prec, r_, s_, a_, bt, emb, dump, raw, lr = m.groups()
    return "wn::%s<%s,%s,%s,%s,%s,%s,%s,%s,%s>" % (k, "true" if prec == "fp16" else "false", r_, s_, a_, bt, "true" if int(emb) else "false",
                                                      "true" if int(dump) else "false", raw, "true" if lr else "false")


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "sha":
        print(kernel_sha(a[1] if len(a) > 2 else "inst_64_256_256_p16.o", a[-1].replace(" ", "")))
        sys.exit(0)
    if not a or a[0] not in ("regs", "mix"):
        sys.exit(__doc__)
    if a[0] == "regs":
        regs(a[1] if len(a) > 1 else "inst_64_256_256_p16.o")
    else:
        inst = a[1] if len(a) > 2 else "inst_64_256_256_p16.o"
        mix(inst, a[-1])
