"""The oracle built with -DSINGLE_PRECISION_SOLVER (oracle/libherc_oracle_f32.so: solver_float = float, psolve.h:60-64)
against what the REAL reference built with the same switch wrote (oracle/_ref/psolve_f32, oracle/build_ref.sh;
tests/golden/make_golden.py case_single): checkpoints of float rows, bit for bit -- tm1 / tm2 / force, the n_t rows and
the locals the reference declares with fvector_t round where the reference's statements round, everything it declares
double stays double."""
import numpy as np
import pytest

from oracle import herc_oracle as ho
from tests import helpers as H


@pytest.mark.parametrize("case,stiff", [("c1_f32", ho.STIFF_EFFECTIVE), ("c1_conv_f32", ho.STIFF_CONVENTIONAL)])
def test_single_precision_checkpoints_of_the_uniform_box_bitwise(case, stiff):
    g = H.load(case)
    assert g["ckpt_tm1"].dtype == np.float32
    p = H.c1_problem("rayleigh", real=np.float32)
    assert p["ntable"].dtype == np.float32 and p["etable"].dtype == np.float64
    # the double-precision tables rounded to float are NOT the reference's float tables: n_t is SUMMED in float
    p64 = H.c1_problem("rayleigh")
    assert np.array_equal(p["etable"], p64["etable"])
    assert not np.array_equal(p["ntable"], p64["ntable"].astype(np.float32))
    assert H.rel_linf(p["ntable"], p64["ntable"]) < 1e-6
    tm1, tm2 = np.zeros((p["N"], 3), np.float32), np.zeros((p["N"], 3), np.float32)
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done, int(step) - done, p["dt"],
                      stiff_method=stiff, loaded_lnid=g["loaded_lnid"], forces=g["forces"])
        done = int(step)
        assert np.array_equal(tm1, g["ckpt_tm2"][k])
        assert np.array_equal(tm2, g["ckpt_tm1"][k])
    assert np.abs(g["ckpt_tm1"][-1]).max() > 10.0
    # how far single precision is from double on this run (the tolerance the GPU tests state comes from here)
    g64 = H.load("c1_short" if stiff == ho.STIFF_EFFECTIVE else "c1_conv")
    k32, k64 = list(g["ckpt_steps"]).index(400), list(g64["ckpt_steps"]).index(400)
    far = H.rel_linf(g["ckpt_tm1"][k32].astype(np.float64), g64["ckpt_tm1"][k64])
    assert 1e-8 < far < (2e-5 if stiff == ho.STIFF_EFFECTIVE else 2e-3)          # measured: 7.3e-6 / 5.4e-4


def test_single_precision_two_level_octree_bitwise():
    """compute_adjust on float tables (mass distribution at init, forces and displacements every step), 800 hanging nodes."""
    p = H.c5_problem("c5_two_level_f32", real=np.float32)
    g = p["golden"]
    assert len(p["dangling"][0]) == int(g["total_dangling"]) == 800
    tm1, tm2 = np.zeros((p["N"], 3), np.float32), np.zeros((p["N"], 3), np.float32)
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done, int(step) - done, p["dt"],
                      loaded_lnid=g["loaded_lnid"], forces=g["forces"], dangling=p["dangling"])
        done = int(step)
        assert np.array_equal(tm1, g["ckpt_tm2"][k])
        assert np.array_equal(tm2, g["ckpt_tm1"][k])


def test_single_precision_on_eight_ranks_against_the_float_references_stripes():
    """The float reference on 8 MPI ranks (two-level octree: hanging nodes shared across ranks): the mass exchange on float
    n_t rows at init, four exchanges of float records and compute_adjust on floats per step (psolve.c:3498-3507, 4298-4315,
    4985-5073).  Bit for bit, as in double: the in-memory exchange adds a node's contributions messenger by messenger in
    schedule_build's list order (oracle/herc_oracle.py: octree_partition), as schedule_senddata does (psolve.c:5035-5073)."""
    pr = H.c5_np8_problem("c5_two_level_np8_f32", real=np.float32)
    g, parts = pr["golden"], pr["parts"]
    assert all(nt.dtype == np.float32 for nt in pr["nts"])
    tm1s = [np.zeros((len(p["nodes"]), 3), np.float32) for p in parts]
    tm2s = [np.zeros((len(p["nodes"]), 3), np.float32) for p in parts]
    done = 0
    for step in g["ckpt_steps"]:
        ho.multi_rank_run(parts, pr["ets"], pr["nts"], tm1s, tm2s, done, int(step) - done, pr["dt"], pr["loaded"], pr["forces"])
        done = int(step)
        stripes = [H.np8_stripe(g, step, p["rank"], len(p["nodes"])) for p in parts]
        scale = max(float(np.abs(ref1).max()) for _, ref1 in stripes)          # of the field, not of a rank's quiet corner
        for p, (ref2, ref1) in zip(parts, stripes):
            assert ref1.dtype == np.float32 and scale > 0
            assert np.array_equal(tm1s[p["rank"]], ref2) and np.array_equal(tm2s[p["rank"]], ref1)
    assert max(float(np.abs(t).max()) for t in tm2s) > 0
