"""The nine-limb field arithmetic of the round-4 kernels (otti_amd/csrc/fr9.h, fp9.h), host build, against the 8 x u32 arithmetic that the
oracle pins (tests/test_host.py::otti_host_selftest, tests/test_oracle.py).  The device bodies (generated asm) are covered on the GPU by
tests/test_gpu_kernels.py / test_gpu_parity.py and tools/limbbench."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nine_limb_arithmetic_matches_the_eight_word_form(tmp_path):
    exe = tmp_path / "limb9_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "otti_amd", "csrc"), os.path.join(ROOT, "tests", "limb9_check.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), "20000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout


def test_generated_asm_is_up_to_date(tmp_path):
    """gen_limb9.py regenerates byte-identical .inc files (nobody edited them by hand)."""
    src = os.path.join(ROOT, "otti_amd", "csrc")
    before = {f: open(os.path.join(src, f)).read() for f in ("fr9_mul_gfx950.inc", "f9_mul_gfx950.inc")}
    subprocess.run([sys.executable, os.path.join(src, "gen_limb9.py")], check=True, capture_output=True)
    for f, text in before.items():
        assert open(os.path.join(src, f)).read() == text, f
