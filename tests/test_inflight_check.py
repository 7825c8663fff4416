"""The scan kernel issues its row / mask-record loads through inline asm and waits for them with hand-written
s_waitcnt (cds_scan.hip); the compiler does not know those destination registers are pending.  This test re-compiles the
kernel to ISA (no GPU needed) and checks that nothing touches such a register between its load and its wait."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_no_access_to_registers_with_a_load_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 accesses to registers with a load in flight" in r.stdout


def test_checker_detects_a_planted_violation():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_inflight as C
    asm = """_ZN3cds11scan_kernelILb1ELb1ELb0ELi3EEEvNS_10ScanParamsE:
\t;;#ASMSTART
\tglobal_load_dwordx3 v[10:12], v1, s[2:3] nt
\t;;#ASMEND
\tv_mov_b32_e32 v20, v11
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_mov_b32_e32 v21, v11
\t.end_amdhsa_kernel
"""
    kernels, problems = C.check(asm)
    assert kernels == 1 and len(problems) == 1 and problems[0][3] == 11
