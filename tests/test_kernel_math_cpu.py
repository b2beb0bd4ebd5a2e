"""The element arithmetic of the HIP kernels (hercules_amd/csrc/hq_kernels.h: the butterfly form of
K = A D A^T) compiled for the HOST by g++ -- the header's HQ_KERNEL_MATH_HOST_CHECK hook -- and checked
against the reference's element matrices: f = -(c1 K1 + c2 K2) w with K1, K2 as compute_K builds them
(psolve.c:3100-3225; pinned on the reference's print_matrix_k dump in test_oracle_golden.py).
No GPU, nothing of the product runs here except that header's arithmetic."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import herc_oracle as ho

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#define HQ_KERNEL_MATH_HOST_CHECK
#include "hq_kernels.h"
extern "C" void element_force(double* X, double* Y, double* Z, double c1, double c2) { hq_element_force(X, Y, Z, c1, c2); }
"""


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("kmath")
    src = d / "harness.cpp"
    src.write_text(HARNESS)
    so = d / "libkmath.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "hercules_amd", "csrc"),
                           "-o", str(so), str(src)])
    return ctypes.CDLL(str(so))


def test_butterfly_product_equals_the_element_matrices(lib):
    K1, K2 = ho.compute_K()                      # [8][8][3][3] fmatrix_t blocks, as the reference lays them out
    K1 = np.asarray(K1).reshape(8, 8, 3, 3)
    K2 = np.asarray(K2).reshape(8, 8, 3, 3)
    rng = np.random.default_rng(2024)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.element_force.argtypes = [dp, dp, dp, ctypes.c_double, ctypes.c_double]
    for c1, c2 in [(1.0, 0.0), (0.0, 1.0), (3.7e9, 1.3e10), (2.5e8, -4.0e7)]:
        w = rng.uniform(-1, 1, (8, 3))
        want = np.zeros((8, 3))
        for i in range(8):
            for j in range(8):
                want[i] -= (c1 * K1[i, j] + c2 * K2[i, j]) @ w[j]      # stiffness.c:121-174, sign of the force
        X, Y, Z = [np.ascontiguousarray(w[:, d]) for d in range(3)]
        lib.element_force(X.ctypes.data_as(dp), Y.ctypes.data_as(dp), Z.ctypes.data_as(dp), c1, c2)
        got = np.stack([X, Y, Z], 1)
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
    # rigid translations and the null space: no force
    for d in range(3):
        w = np.zeros((8, 3)); w[:, d] = 1.0
        X, Y, Z = [np.ascontiguousarray(w[:, k]) for k in range(3)]
        lib.element_force(X.ctypes.data_as(dp), Y.ctypes.data_as(dp), Z.ctypes.data_as(dp), 2.0, 3.0)
        assert max(np.abs(X).max(), np.abs(Y).max(), np.abs(Z).max()) <= 1e-14
