#!/usr/bin/env python3
"""Bank-conflict model of the patch kernel's LDS gathers / atomics (CPU only).

One wave-instruction reads or adds one 8-byte component of one corner node for 64 elements
(lanes).  Node rows are 24 bytes, the LDS has 64 banks of 4 bytes, a b64 access of 32 lanes is one
pass when the 32 rows are distinct modulo 32 (6 dwords x 32 = 192 = 0 mod 64); k rows of one class
cost k passes.  The model counts passes for a full interior patch (8^3 owned nodes, 9^3 elements,
10^3 local nodes) under different numberings of local nodes and orders of elements.

    python profiles/tools/lds_conflict_model.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import herc_oracle as ho  # noqa: E402  (mesh numbering only; nothing is timed or shipped)


def passes(lanes, lnid, row_of, lane_of=None):
    """lanes: element per lane (None = idle lane), in lane order."""
    tot = ideal = 0
    for w in range(0, len(lanes), 64):
        for c in range(8):
            for h in (0, 32):
                rows = {row_of[int(lnid[e][c])] for e in lanes[w + h:w + h + 32] if e is not None}
                if not rows:
                    continue
                cls = {}
                for r in rows:
                    cls[r % 32] = cls.get(r % 32, 0) + 1
                tot += max(cls.values())
                ideal += 1
    return tot, ideal


def main():
    n = 24
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(n, n, n)
    own = np.nonzero(np.all((node_ijk >= 8) & (node_ijk < 16), axis=1))[0]
    base, nown = int(own.min()), len(own)
    assert own.max() - base + 1 == nown == 512          # the Z-ordered ids of the cube are contiguous
    owned = set(range(base, base + nown))
    elems = [e for e in range(len(lnid)) if any(int(v) in owned for v in lnid[e])]
    halo = sorted({int(v) for e in elems for v in lnid[e]} - owned)
    print("patch: %d owned, %d halo, %d elements" % (nown, len(halo), len(elems)))

    cur = {g: g - base for g in owned}
    cur.update({g: nown + i for i, g in enumerate(halo)})
    print("as shipped (owned in id order | halo in id order, elements in id order):  %d passes for %d instructions" % passes(elems, lnid, cur))

    lo = np.array([7, 7, 7])
    lat = {g: int((node_ijk[g] - lo) @ np.array([1, 10, 100])) for g in list(owned) + halo}
    xfast = sorted(elems, key=lambda e: (elem_ijk[e][2], elem_ijk[e][1], elem_ijk[e][0]))
    print("lattice rows i + 10 j + 100 k, elements x fastest in consecutive lanes:   %d / %d" % passes(xfast, lnid, lat))

    # lanes on the SAME lattice as the rows: lane = ei + 10 ej + 100 ek (ei, ej < 9: every tenth lane idle)
    lanes = [None] * 900
    for e in elems:
        i, j, k = elem_ijk[e] - lo
        lanes[int(i + 10 * j + 100 * k)] = e
    print("lattice rows, lanes on the same lattice (900 lanes, 729 elements):         %d / %d" % passes(lanes, lnid, lat))




def residue_variant():
    """Can the planner alone do it (owned rows stay in id = Z order, as the kernel's thread <-> node
    map wants)?  Owned row mod 32 is the interleaved (x mod 4, y mod 4, z mod 2), so any 4 x 4 x 2
    window of elements reads 32 distinct classes at every corner IF the halo rows follow the same
    rule: the aligned 8^3 cube of elements in lanes 0..511 becomes conflict-free (128 passes for 128
    instructions with 1472 row slots, 162 with the 1024 there are).  But the 217 elements of the -1
    layer lie on three faces, where only 8 of the 32 classes occur: their 56 instructions keep ~210
    passes, and the total stays where it is.  Hence the lattice form, whose owned rows are not in id
    order."""
    n = 24
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(n, n, n)
    own = np.nonzero(np.all((node_ijk >= 8) & (node_ijk < 16), axis=1))[0]
    base, nown = int(own.min()), len(own)
    owned = set(range(base, base + nown))
    elems = [e for e in range(len(lnid)) if any(int(v) in owned for v in lnid[e])]
    halo = sorted({int(v) for e in elems for v in lnid[e]} - owned)

    def res(ijk):
        x, y, z = int(ijk[0]) % 4, int(ijk[1]) % 4, int(ijk[2]) % 2
        return (x & 1) | ((y & 1) << 1) | ((z & 1) << 2) | ((x >> 1) << 3) | ((y >> 1) << 4)

    inner = [e for e in elems if np.all((elem_ijk[e] >= 8) & (elem_ijk[e] < 16))]
    outer = [e for e in elems if e not in set(inner)]
    for nrows in (1024, 1472):
        row = {g: g - base for g in owned}
        free = {c: [r for r in range(512, nrows) if r % 32 == c] for c in range(32)}
        spill = []
        for g in sorted(halo, key=lambda g: -int((node_ijk[g] >= 16).any())):     # the +8 faces first
            c = res(node_ijk[g])
            if free[c]:
                row[g] = free[c].pop(0)
            else:
                spill.append(g)
        for g, r in zip(spill, sorted(r for c in free for r in free[c])):
            row[g] = r
        print("planner only, %4d row slots (%3d halo nodes off their class): aligned cube %d / %d, whole patch %d / %d"
              % ((nrows, len(spill)) + passes(list(inner), lnid, row) + passes(list(inner) + outer, lnid, row)))


if __name__ == "__main__":
    main()
    residue_variant()
