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


if __name__ == "__main__":
    main()
