"""The window logic of the full-size octree parity test (tests/test_gpu_fullsize.py: the 189 M-element basin against the
oracle through dependency cones that straddle its level interfaces), checked here against a WHOLE-mesh oracle run on a
three-level box small enough for it: a window's inner nodes must come out exactly as in the whole mesh."""
import numpy as np

from hercules_amd import host
from oracle import herc_oracle as ho
from tests import helpers as H


def test_windows_across_level_interfaces_reproduce_the_whole_mesh():
    levels = [(8, 1100.0, 600.0, 2000.0), (6, 2000.0, 1100.0, 2300.0), (4, 3600.0, 2000.0, 2500.0)]
    nx = ny = 48
    box = host.OctBox(nx, ny, 0, 0, 100.0, 0.02, 0.5, levels=levels)
    assert box.ldnnum > 0
    rng = np.random.default_rng(3)
    u1 = rng.uniform(-1, 1, (box.N, 3)) * 1e-3
    u2 = u1 * 0.999
    ho.compute_adjust(u1, 1, box.dangling)
    ho.compute_adjust(u2, 1, box.dangling)
    k = 2
    w1, w2 = u2.copy(), u1.copy()
    ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), w1, w2, 0, k, box.dt, dangling=box.dangling)
    whole1, whole2 = w2, w1
    xyz = box.node_xyz
    elem_lo = xyz[box.lnid[:, 0]].astype(np.int64)
    elem_edge = (xyz[box.lnid[:, 1], 0].astype(np.int64) - elem_lo[:, 0])
    scale = np.abs(u1).max()
    checked = 0
    # interface fine | middle at z = 8 (c = 2), middle | coarse at z = 8 + 12 = 20 (c = 4)
    for z0, c, x0 in ((8, 2, 0), (8, 2, 12), (8, 2, 28), (20, 4, 0), (20, 4, 16)):
        margin = 2 * k * c
        W = 2 * margin + 2 * c
        lo = [x0, max(0, (nx - W) // (2 * c) * c) if x0 else 0, max(0, z0 - margin - c)]
        hi = [min(nx, lo[0] + W), min(ny, lo[1] + W), z0 + margin + c]
        hi[2] = min(hi[2], int(xyz[:, 2].max()))
        win = H.octree_window(box.lnid, xyz, box.dangling, elem_lo, elem_edge, lo, hi, margin)
        assert len(win["dangling"][0]) > 0 and win["ok"].sum() > 20
        g1, g2 = H.octree_window_oracle(win, box.etable, box.ntable, u1, u2, k, box.dt)
        ok, nodes = win["ok"], win["nodes"]
        # the checked set holds hanging nodes and anchors of the interface
        hang = np.isin(nodes[ok], box.dangling[0])
        assert hang.any()
        assert np.abs(g1[ok] - whole1[nodes[ok]]).max() <= 1e-12 * scale
        assert np.abs(g2[ok] - whole2[nodes[ok]]).max() <= 1e-12 * scale
        # and the margin is not vacuous: nodes on a cut face do differ
        if (~ok).any() and (lo[0] > 0 or hi[0] < nx):
            assert np.abs(g1[~ok] - whole1[nodes[~ok]]).max() > 1e-9 * scale
        checked += int(ok.sum())
    assert checked > 500
    box.close()


def test_windows_across_lateral_interfaces_reproduce_the_whole_mesh():
    """The same for a LATERALLY refined mesh (bench.py's o4s: a sediment bowl in a layered half-space meshed by
    hqh_octree_generate, 0.93 M elements on four levels): windows centred on hanging nodes of every orientation --
    mid-edge nodes on x / y / z edges, mid-face nodes on faces normal to x / y / z -- cut out by tests/helpers.lateral_windows
    and stepped by the oracle give the whole-mesh oracle's values at their inner nodes."""
    import bench
    box, E, N, it = bench.make_octbox("o4s", 0, 1)
    u1 = it["field"]
    u2 = 0.999 * u1
    deps, mask, dist = H.hanging_kinds(box.node_xyz, box.dangling)
    assert set(mask.tolist()) == {1, 2, 3, 4, 5, 6} and set(dist.tolist()) >= {1, 2, 4}
    k = 2
    w1, w2 = u2.copy(), u1.copy()
    ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), w1, w2, 0, k, box.dt, dangling=box.dangling)
    whole1, whole2 = w2, w1
    xyz = box.node_xyz
    elem_lo = xyz[box.lnid[:, 0]].astype(np.int64)
    elem_edge = (xyz[box.lnid[:, 1], 0].astype(np.int64) - elem_lo[:, 0])
    scale = np.abs(u1).max()
    wins = H.lateral_windows(xyz, box.dangling, elem_lo, elem_edge, k, kinds={(1, 1), (2, 2), (4, 1), (3, 1), (5, 2), (6, 1), (6, 4)})
    assert len(wins) == 7
    for lo, hi, margin, centre, cand in wins:
        win = H.octree_window(box.lnid, xyz, box.dangling, elem_lo, elem_edge, lo, hi, margin, cand)
        g1, g2 = H.octree_window_oracle(win, box.etable, box.ntable, u1, u2, k, box.dt)
        ok, nodes = win["ok"], win["nodes"]
        assert centre in nodes[ok]
        assert np.abs(g1[ok] - whole1[nodes[ok]]).max() <= 1e-12 * scale
        assert np.abs(g2[ok] - whole2[nodes[ok]]).max() <= 1e-12 * scale
        assert np.abs(g1[~ok] - whole1[nodes[~ok]]).max() > 1e-9 * scale
    box.close()
