"""The header-only C++ mirror of the reference interface (include/sparrowhawk_asm.hpp) over the C ABI:
builds with plain g++, fails loudly without a HIP device, and on the GPU gives the oracle's bytes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(tmp):
    exe = os.path.join(tmp, "mirror_main")
    libdir = os.path.join(ROOT, "sparrowhawk_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "mirror_main.cpp"), "-o", exe,
                           "-L", libdir, "-l:libshk_hip.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def env():
    e = dict(os.environ)
    e["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + e.get("LD_LIBRARY_PATH", "")
    return e


def test_cpp_mirror_builds_and_has_no_cpu_fallback(tmp_path):
    import torch
    exe = build(str(tmp_path))
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"@r0\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n")
    p = subprocess.run([exe, str(fq), "31", "0", str(tmp_path / "o")], env=env(), capture_output=True, text=True)
    if torch.cuda.is_available():
        assert p.returncode == 0, p.stderr
    else:
        assert p.returncode == 3 and "HIP device" in p.stderr, (p.returncode, p.stderr)


@pytest.mark.gpu
def test_cpp_mirror_matches_oracle(tmp_path):
    from util import make_dataset, run_oracle
    exe = build(str(tmp_path))
    g, data = make_dataset(20000, 30, err=0.005, seed=11)
    fq = tmp_path / "r.fq"
    fq.write_bytes(data)
    out = str(tmp_path / "o")
    p = subprocess.run([exe, str(fq), "31", "3", out], env=env(), capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    o = run_oracle([data], k=31, min_count=3)
    o.assemble()
    assert open(out + ".pre.json").read() == o.preprocessing_json()
    assert open(out + ".asm.json").read() == o.assembly_json()
    states = open(out + ".states").read().split()
    assert states[0] == "preprocess:start" and states[-1] == "assembly:end"
