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
extern "C" void material_coef(float rho, float Vs, float Vp, double A, double h, double dt, double bbase, double thr_damp,
                              double thr_vpvs, double* out)
{
    hq_mat_const K = { A, h, dt, bbase, thr_damp, thr_vpvs };
    hq_material_coef(rho, Vs, Vp, K, out, out + 1, out + 2);
}
extern "C" void element_force_zmodes(double* X, double* Y, double* Z, double c1, double c2) { hq_element_force<true>(X, Y, Z, c1, c2); }
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


def test_z_modes_of_the_transposed_butterfly(lib):
    """hq_k_brick_het sums the corner forces over elements BEFORE the z stage of the transposed butterfly
    (hq_element_force<true>: X[0..3] = a, X[4..7] = b): f(near-z corner n) = a[n] - b[n], f(far-z corner n + 4) =
    a[n] + b[n] must be the full product."""
    rng = np.random.default_rng(7)
    dp = ctypes.POINTER(ctypes.c_double)
    for fn in (lib.element_force, lib.element_force_zmodes):
        fn.argtypes = [dp, dp, dp, ctypes.c_double, ctypes.c_double]
    for c1, c2 in [(1.0, 0.0), (0.0, 1.0), (3.7e9, 1.3e10)]:
        w = rng.uniform(-1, 1, (8, 3))
        full = [np.ascontiguousarray(w[:, d]) for d in range(3)]
        modes = [np.ascontiguousarray(w[:, d]) for d in range(3)]
        lib.element_force(*[a.ctypes.data_as(dp) for a in full], c1, c2)
        lib.element_force_zmodes(*[a.ctypes.data_as(dp) for a in modes], c1, c2)
        for f, m in zip(full, modes):
            scale = np.abs(f).max()
            assert np.abs((m[:4] - m[4:]) - f[:4]).max() <= 4e-16 * scale
            assert np.abs((m[:4] + m[4:]) - f[4:]).max() <= 4e-16 * scale


def test_assembled_stencil_coefficients_equal_the_assembled_element_matrices():
    """hq_k_patch_stencil steps uniform lattice patches with the 27-point stencil S = c1 S1 + c2 S2 assembled
    from the element matrix.  Its sixteen coefficients (libhq_solver.so builds them on the host from the
    kernels' own element arithmetic) against the SAME assembly of the reference's K1, K2 (compute_K), and the
    cube symmetry the kernel relies on checked entry by entry: f(node) = -sum_d (c1 S1 + c2 S2)[d] w(node + d)."""
    from hercules_amd import build as hbuild
    import hercules_amd as ha
    hbuild.build()
    lib = ha.load_library()
    out = (ctypes.c_double * 16)()
    assert lib.hq_stencil_coefficients(out) == 0
    got = np.array(out[:])
    K = [np.asarray(k).reshape(8, 8, 3, 3) for k in ho.compute_K()]
    for which in range(2):
        S = np.zeros((3, 3, 3, 3, 3))
        for o in range(8):                      # the node is corner o of the element at offset -o
            for m in range(8):
                d = [-((o >> k) & 1) + ((m >> k) & 1) for k in range(3)]
                S[d[0] + 1, d[1] + 1, d[2] + 1] -= K[which][o, m]      # force = -K w
        p = got[6 * which:6 * which + 6]
        q = got[12 + 2 * which:14 + 2 * which]
        scale = np.abs(S).max()
        for d in np.ndindex(3, 3, 3):
            dd = [v - 1 for v in d]
            for a in range(3):
                for b in range(3):
                    if a == b:
                        others = [dd[k] for k in range(3) if k != a]
                        want = p[(dd[a] != 0) + 2 * sum(v != 0 for v in others)]
                    else:
                        c = 3 - a - b
                        want = q[int(dd[c] != 0)] * np.sign(dd[a]) * np.sign(dd[b])
                    assert abs(S[d][a, b] - want) <= 1e-13 * scale, (which, dd, a, b)


def test_coefficients_from_three_floats_are_solver_inits_bit_for_bit(lib):
    """hq_material_coef (the 12-byte form of an element's (c1, c2, beta) that hq_k_brick_het<PACKED> expands on the
    device) against the eTable the oracle's solver_init builds (pinned bitwise on the reference's own checkpoints):
    thousands of random materials through every branch of mu_and_lambda (psolve.c:3236-3272: the Vp/Vs cap, the
    negative-lambda fix that rewrites Vp) and both sides of the damping threshold (psolve.c:3397-3401), several
    element sizes, time steps and damping types -- c1, c2 and beta = c3 / c1 equal to the last bit."""
    rng = np.random.default_rng(77)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.material_coef.argtypes = [ctypes.c_float] * 3 + [ctypes.c_double] * 6 + [dp]
    n = 4000
    checked = {"plain": 0, "capped": 0, "fixed": 0, "zeta_thr": 0}
    for h, dt, freq, damping in ((62.5, 1e-3, 5.0, ho.DAMP_RAYLEIGH), (1.953125, 9e-5, 200.0, ho.DAMP_RAYLEIGH),
                                 (100.0, 0.02, 0.5, ho.DAMP_RAYLEIGH), (31.25, 5e-4, 10.0, ho.DAMP_NONE)):
        vs = rng.uniform(80.0, 4000.0, n).astype(np.float32)
        ratio = np.where(rng.random(n) < 0.25, rng.uniform(3.0, 12.0, n), rng.uniform(1.0, 3.0, n))
        vp = (vs * ratio).astype(np.float32)
        vp[: n // 8] = (vs[: n // 8] * rng.uniform(0.9, 1.45, n // 8)).astype(np.float32)     # Vp^2 < 2 Vs^2: negative lambda
        rho = rng.uniform(1500.0, 3000.0, n).astype(np.float32)
        edata = np.stack([np.full(n, h, np.float32), vp, vs, rho], 1).copy()
        before = edata.copy()
        lnid = np.arange(8 * n, dtype=np.int32).reshape(n, 8)          # disjoint elements: only the eTable matters here
        et, _ = ho.solver_init(lnid, edata, np.zeros(n, np.uint8), 8 * n, dt, freq, damping=damping)
        _, bbase = ho.setab(freq, damping)
        hf = float(np.float32(h))
        A = (dt * dt) * hf
        out = np.zeros(3)
        for e in range(n):
            fixed = edata[e, 1] != before[e, 1]                         # solver_init rewrote Vp
            r = -float(edata[e, 3]) if fixed else float(edata[e, 3])
            lib.material_coef(r, float(edata[e, 2]), float(edata[e, 1]), A, hf, dt, bbase, 0.05, 3.0, out.ctypes.data_as(dp))
            assert out[0] == et[e, 0] and out[1] == et[e, 1], (e, out, et[e])
            assert out[2] == (et[e, 2] / et[e, 0])
            checked["fixed" if fixed else ("capped" if before[e, 1] > before[e, 2] * 3.0 else "plain")] += 1
            checked["zeta_thr"] += int(10.0 / float(edata[e, 2]) > 0.05)
    assert min(checked.values()) > 200, checked
    out = np.ones(3)
    lib.material_coef(0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0, 0.05, 3.0, out.ctypes.data_as(dp))      # no element
    assert not out.any()
