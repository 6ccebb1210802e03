"""field.cuh on the host: the lazy four-product QM31 multiplication (round 6) against the textbook tower formula, boundary values in every
coordinate + random operands (tests/native/field_selftest.cpp, built with hipcc; the functions are __host__ __device__)."""
import os, shutil, subprocess, sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_q_mul_equals_the_tower_formula_on_boundary_and_random_operands(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "field_selftest")
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "native", "field_selftest.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout
