"""The C++ host mirror (include/cds_hip.hpp): compiles against the C ABI on CPU; runs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "mirror_test.cpp")
LIBDIR = os.path.join(ROOT, "colormipsearch_amd")


def _build(tmp_path):
    exe = str(tmp_path / "mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", LIBDIR, "-lcds_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path):
    from colormipsearch_amd import capi
    capi.load()
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cpp_mirror_runs(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    tag, self_score, mir_score, mir_flag, qsize = out.stdout.split()
    assert tag == "OK" and int(self_score) == int(mir_score) and mir_flag == "1" and int(self_score) > 0.9 * int(qsize)
