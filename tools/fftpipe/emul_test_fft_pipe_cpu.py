"""CPU replay of the pipelined Circle-FFT kernels (csrc/fft_pipe.hip): the lane-level code the GPU runs (csrc/fft_pipe.cuh, host +
device) driven phase by phase on plain arrays, against the oracle's interpolate / evaluate.  Also: the work-item decoding is a
bijection, and the LDS bank model says the tile swizzle is conflict free for every access pattern of the kernels."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "fftpipe_emul.cpp")
OUT = os.path.join(ROOT, "tests", "_build", "libfftpipe_emul.so")


@pytest.fixture(scope="module")
def emul():
    deps = [SRC, os.path.join(ROOT, "nexus-zkvm_amd", "csrc", "fft_pipe.cuh"), os.path.join(ROOT, "nexus-zkvm_amd", "csrc", "field.cuh"),
            os.path.join(ROOT, "oracle", "poly.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", OUT, SRC])
    lib = C.CDLL(OUT)
    lib.fftpipe_emul_lde.argtypes = [C.c_int, C.c_int, C.c_uint64]
    lib.fftpipe_emul_item_bijection.argtypes = [C.c_uint32, C.c_uint32]
    return lib


@pytest.mark.parametrize("n", [17, 18, 19, 20])   # K = 4..7: every remainder-round shape of the middle launch; 21 / 22 below
def test_replayed_lde_equals_oracle(emul, n):
    assert emul.fftpipe_emul_lde(n, 2 if n <= 18 else 1, 1000 + n) == 0


@pytest.mark.parametrize("n", [21, 22])
def test_replayed_lde_equals_oracle_headline_sizes(emul, n):
    assert emul.fftpipe_emul_lde(n, 1, 77 + n) == 0


def test_unsupported_sizes_are_refused(emul):
    assert emul.fftpipe_emul_lde(16, 1, 1) == -1 and emul.fftpipe_emul_lde(23, 1, 1) == -1


@pytest.mark.parametrize("tiles,n_cols", [(16, 1), (16, 3), (512, 2), (1024, 5), (4, 7), (64, 438)])
def test_item_decoding_is_a_bijection(emul, tiles, n_cols):
    assert emul.fftpipe_emul_item_bijection(tiles, n_cols) == 0


def test_tile_swizzle_is_bank_conflict_free():
    sys.path.insert(0, os.path.join(ROOT, "tools", "fftpipe"))
    import lds_conflicts as M
    assert M.worst_case(M.swz) == 1       # every round / store / hand-over pattern: one LDS cycle per lane group
    assert M.worst_case(M.swz_none) > 1   # the model does see conflicts without the swizzle
