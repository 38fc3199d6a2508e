#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (never the product): builds the REFERENCE's own PyTorch binding against this repo's
libwavenet_infer.so, to prove the drop-in claim of INTEGRATION.md 2a by compiling the reference side.

  oracle/_ref/nv_wavenet_ext.so   <- /root/reference/pytorch/wavenet_infer_wrapper.cpp (pybind11), with the two
                                     edits INTEGRATION.md documents for current PyTorch (no <THC/THC.h>,
                                     .data<T>() -> .data_ptr<T>()), compiled against the REFERENCE's own
                                     pytorch/wavenet_infer.h and linked to nv_wavenet_amd/libwavenet_infer.so

The patched copy of the wrapper lives in a temporary directory only; nothing of the reference's sources enters the
repository (oracle/_ref/ is git-ignored build output -- a compiled .so -- that travels to the GPU box like our own .so files).
The reference's PYTHON side (pytorch/nv_wavenet.py) does not travel in any form: no copy, no bytecode (rounds 4-5 shipped a .pyc of
it; removed in round 6, and a stale one is deleted here).  It is imported where it lies, in the authoring container only
(tests/test_capi_cpu.py::test_reference_python_wrapper_prepares_the_same_tensors); on the GPU box the reference's compiled
extension is driven by this repo's mirror of that class (tests/test_parity_gpu.py::test_reference_pybind_extension_on_this_library)."""
import os
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("REFERENCE", "/root/reference")


def main():
    stale = os.path.join(HERE, "_ref", "nv_wavenet_ref.pyc")
    if os.path.exists(stale):
        os.remove(stale)
    src = os.path.join(REF, "pytorch", "wavenet_infer_wrapper.cpp")
    if not os.path.exists(src):
        print("reference tree %s absent: keeping prebuilt oracle/_ref binding (if any)" % REF)
        return 0
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    out_so = os.path.join(out_dir, "nv_wavenet_ext.so")
    lib = os.path.join(ROOT, "nv_wavenet_amd", "libwavenet_infer.so")
    assert os.path.exists(lib), "build nv_wavenet_amd/libwavenet_infer.so first"
    newest_in = max(os.path.getmtime(src), os.path.getmtime(__file__))
    if not (os.path.exists(out_so) and os.path.getmtime(out_so) >= newest_in):
        import pybind11
        from torch.utils import cpp_extension
        import torch
        text = open(src).read()
        assert "#include <THC/THC.h>" in text and ".data<" in text
        text = text.replace("#include <THC/THC.h>\n", "").replace(".data<", ".data_ptr<")
        with tempfile.TemporaryDirectory() as tmp:
            patched = os.path.join(tmp, "wavenet_infer_wrapper.cpp")
            open(patched, "w").write(text)
            inc = cpp_extension.include_paths() + [pybind11.get_include(), sysconfig.get_paths()["include"],
                                                   os.path.join(REF, "pytorch")]
            tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
            cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-w", "-DTORCH_EXTENSION_NAME=nv_wavenet_ext",
                   "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
                   "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"]
            cmd += ["-I" + p for p in inc] + ["-I/opt/rocm/include"]
            cmd += [patched, "-o", out_so, "-L" + tlib, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10",
                    "-L" + os.path.dirname(lib), "-lwavenet_infer",
                    "-Wl,-rpath,$ORIGIN/../../nv_wavenet_amd", "-Wl,-rpath," + tlib]
            subprocess.check_call(cmd)
        print("built oracle/_ref/nv_wavenet_ext.so from %s" % src)
    return 0


if __name__ == "__main__":
    sys.exit(main())
