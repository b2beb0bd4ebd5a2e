"""TEST INFRASTRUCTURE, NOT PRODUCT: dependency-cone windows of an octree mesh -- oracle parity at sizes the oracle
cannot run whole.  Used by tests/ (through tests/helpers.py) and by bench.py's parity windows BEHIND its timed region."""
import numpy as np

from oracle import herc_oracle as ho



def octree_window(lnid, node_xyz, dangling, elem_lo, elem_edge, lo, hi, margin, cand=None):
    """The elements of an octree mesh that lie inside the box [lo, hi] (node coordinates, finest-element units; lo / hi
    must be multiples of the coarsest edge inside, so that no element straddles a face and every hanging node of the
    window finds its anchors in it), renumbered as a mesh of their own.
    -> dict(elems, nodes, lnid, dangling (window numbering, table order kept), ok) with ok = the window nodes at least
    `margin` inside every CUT face (a face of the window that is not a face of the domain): after k steps of
    solver_run a node depends on nodes within <= 2 k c of it (c the coarsest edge around: one element hop per step plus
    the hop from a hanging node to its anchors, compute_adjust psolve.c:5936-6039), so with margin >= 2 k c the oracle
    on the window gives the exact values of those nodes."""
    lo, hi = np.asarray(lo, np.int64), np.asarray(hi, np.int64)
    if cand is None:                                            # (cand: elements known to hold every element of the window)
        cand = np.nonzero((elem_lo[:, 0] >= lo[0]) & (elem_lo[:, 0] < hi[0]))[0]   # the x slab; y, z and the edges below
    c_lo, c_edge = elem_lo[cand], elem_edge[cand]
    inside = np.ones(len(cand), bool)
    for d in range(3):
        inside &= (c_lo[:, d] >= lo[d]) & (c_lo[:, d] + c_edge <= hi[d])
    elems = cand[np.nonzero(inside)[0]]
    assert len(elems) > 0
    assert int(elem_edge[elems].astype(np.int64).__pow__(3).sum()) == int(np.prod(hi - lo)), "the window is not filled by whole elements"
    nodes, inv = np.unique(lnid[elems], return_inverse=True)
    lnid_w = inv.reshape(-1, 8).astype(np.int32)
    ids, ptr, anchors = dangling

    def pos(v):                      # place of the mesh's nodes v among the window's (sorted) nodes, -1 where absent
        v = np.asarray(v)            # (a search, not a table over all nodes of a 100 M-element mesh per window)
        i = np.minimum(np.searchsorted(nodes, v), len(nodes) - 1)
        return np.where(nodes[i] == v, i, -1).astype(np.int64)
    sel = np.nonzero(pos(ids) >= 0)[0] if len(ids) else np.zeros(0, np.int64)
    w_ids, w_ptr, w_anc = [], [0], []
    for k in sel:
        la = pos(anchors[ptr[k]:ptr[k + 1]])
        assert (la >= 0).all(), "a hanging node of the window has an anchor outside it: window not aligned to the coarse grid"
        w_ids.append(int(pos(ids[k:k + 1])[0]))
        w_anc += [int(v) for v in la]
        w_ptr.append(len(w_anc))
    dom_hi = node_xyz.max(axis=0).astype(np.int64)
    q = node_xyz[nodes].astype(np.int64)
    ok = np.ones(len(nodes), bool)
    for d in range(3):
        if lo[d] > 0:
            ok &= q[:, d] >= lo[d] + margin
        if hi[d] < dom_hi[d]:
            ok &= q[:, d] <= hi[d] - margin
    return dict(elems=elems, nodes=nodes.astype(np.int32), lnid=lnid_w, ok=ok,
                dangling=(np.array(w_ids, np.int32), np.array(w_ptr, np.int32), np.array(w_anc, np.int32)))


def octree_window_oracle(win, etable, ntable, u1, u2, k, dt):
    """k steps of the oracle's reference loops (+ compute_adjust) on a window: (tm1, tm2) of the window's nodes,
    post-swap as hq_download / hq_gather return them."""
    o2 = u1[win["nodes"]].copy()                                        # the oracle's arrays are pre-swap
    o1 = u2 * o2 if np.isscalar(u2) else u2[win["nodes"]].copy()        # (u2 a number: u(t - dt) = u2 * u(t))
    dn = win["dangling"] if len(win["dangling"][0]) else None
    ho.solver_run(win["lnid"], etable[win["elems"]].copy(), ntable[win["nodes"]].copy(), o1, o2, 0, k, dt, dangling=dn)
    return o2, o1


def hanging_kinds(node_xyz, dangling):
    """Per hanging node: (number of anchors, axes along which its anchors differ as a 3-bit mask, distance to an anchor)
    -- 2 anchors: mid-edge node of an edge along x (1) / y (2) / z (4); 4 anchors: mid-face node of a face normal to
    z (3) / y (5) / x (6)."""
    ids, ptr, anchors = dangling
    q = node_xyz.astype(np.int64)
    first = q[anchors[ptr[:-1]]]
    last = q[anchors[ptr[1:] - 1]]
    diff = (first != last)
    mask = diff[:, 0] * 1 + diff[:, 1] * 2 + diff[:, 2] * 4
    dist = np.abs(first - q[ids]).max(axis=1)
    return np.diff(ptr), mask, dist


def lateral_windows(node_xyz, dangling, elem_lo, elem_edge, k, per_kind=1, seed=5, kinds=None, max_elems=700000):
    """Dependency-cone windows of an octree mesh centred on hanging nodes of every kind present (orientation x level),
    `per_kind` of each picked by a seeded generator: -> [(lo, hi, margin, centre node, candidate elements)].  Window faces are aligned to
    the coarsest edge of the mesh (no element straddles them); margin = 2 k c with c the coarsest edge INSIDE the
    window (octree_window's rule)."""
    ids, ptr, anchors = dangling
    deps, mask, dist = hanging_kinds(node_xyz, dangling)
    rng = np.random.default_rng(seed)
    A = int(elem_edge.max())
    far = node_xyz.max(axis=0).astype(np.int64)
    out = []
    for key in sorted(set(zip(mask.tolist(), dist.tolist()))):
        if kinds is not None and key not in kinds:
            continue
        cand = np.nonzero((mask == key[0]) & (dist == key[1]))[0]
        for pick in rng.choice(cand, min(per_kind, len(cand)), replace=False):
            q = node_xyz[ids[pick]].astype(np.int64)
            c = 2 * int(key[1])
            # ONE pass over the mesh per window: the elements that can lie in the largest window this node may get
            reach = 2 * k * A + 2 * A
            # (the slab in x over the whole mesh, then y and z on what is left of it)
            cand = np.nonzero((elem_lo[:, 0] >= q[0] - reach - A) & (elem_lo[:, 0] <= q[0] + reach))[0]
            for d in (1, 2):
                v = elem_lo[cand, d]
                cand = cand[(v >= q[d] - reach - A) & (v <= q[d] + reach)]
            c_lo = elem_lo[cand].astype(np.int64)
            c_hi = c_lo + elem_edge[cand].astype(np.int64)[:, None]
            for _ in range(3):
                half = 2 * k * c + c
                lo = np.maximum(0, (q - half) // A * A)
                hi = np.minimum(far, -((-(q + half)) // A) * A)
                inside = np.all(c_lo >= lo, axis=1) & np.all(c_hi <= hi, axis=1)
                c2 = int(elem_edge[cand][inside].max())
                if c2 == c:
                    break
                c = c2
            if inside.sum() <= max_elems:
                out.append((lo.tolist(), hi.tolist(), 2 * k * c, int(ids[pick]), cand))
    return out
