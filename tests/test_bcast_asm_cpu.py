"""wn::wavenet_bcast waits for its assembly-issued loads by hand (wn_bcast.hpp).  scripts/check_bcast_asm.py compiles the C3
fp16 instantiation to assembly (hipcc cross-compiles without a GPU) and verifies on the emitted code that no instruction
touches the destination of such a load before a wait that covers it -- the way a compiler-inserted copy on a loop edge did
in the first version of the kernel."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_touches_an_assembly_issued_load_before_its_wait(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_bcast_asm.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "register loads checked" in l]
    assert len(lines) >= 4 and all(" 0 violations" in l for l in lines), out.stdout[-3000:]
    assert all(int(l.split(":")[1].split()[0]) >= 30 for l in lines), "the checker found no loads to check: " + out.stdout[-2000:]
