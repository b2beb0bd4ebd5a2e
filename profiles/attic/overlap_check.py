#!/usr/bin/env python3
"""Do the exchange chain's kernels run BESIDE the interior patch launch?  ("Next round" 4.)

Input: the kernel trace (rocprofv3 --kernel-trace --output-format csv) of
    HQ_OVERLAP=1 python3 bench.py --workload c3 --inproc-parts P --steps K --warmup W
i.e. P partitions of the box stepped in one process on one GPU with the exchange chain of every partition on its own
stream, the arrangement hq_comm_init uses between GPUs.  Every partition's step is: element launch + interface stencil
launch -> event -> (exchange stream: pack, copies, hq_k_interface_update, unpack) beside (compute stream: the
interior stencil launches).  The trace does not name the partition of a kernel, so the check is the strict one: a
chain kernel counts as "beside" only if at its start EVERY partition has an interior patch launch in flight (then its
own partition's is among them).  Prints the counts and a timeline excerpt of one step.
"""
import csv
import sys


def main(path, parts):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"],
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    t0 = rows[0][0]
    patch = [r for r in rows if "hq_k_patch" in r[2]]
    # interior launches: the big ones (the interface launches ahead of the exchange are a few thousand workgroups)
    big = sorted(p[4] for p in patch)[len(patch) // 2]
    interior = [p for p in patch if p[4] >= big]
    chain = [r for r in rows if any(k in r[2] for k in ("hq_k_pack", "hq_k_unpack", "hq_k_interface_update", "hq_k_iface"))]
    hist = {}
    for c in chain:
        n = sum(1 for p in interior if p[0] <= c[0] < p[1])
        hist[n] = hist.get(n, 0) + 1
    tot = len(chain)
    print("patch launches: %d (interior, >= %d workgroups: %d), chain kernels: %d" % (len(patch), big, len(interior), tot))
    for n in sorted(hist):
        print("  chain kernels that start with %d interior launches in flight: %5d  (%.0f %%)" % (n, hist[n], 100.0 * hist[n] / tot))
    some = sum(v for n, v in hist.items() if n >= 1)
    inside = sum(1 for c in chain if any(p[0] <= c[0] and c[1] <= p[1] for p in interior))
    beside = sum(v for n, v in hist.items() if n >= parts)
    print("chain kernels that start while an interior launch is in flight: %d of %d = %.0f %%; that run from start to end "
          "inside one: %d = %.0f %%" % (some, tot, 100.0 * some / tot, inside, 100.0 * inside / tot))
    print("... while EVERY partition has one in flight (>= %d: their own is among them for certain): %d = %.0f %%"
          % (parts, beside, 100.0 * beside / tot))
    # excerpt: 40 kernels from the middle of the run, starting at an element launch (the head of a partition's step)
    mid = len(rows) // 2
    while mid < len(rows) - 40 and "hq_k_patch_seed" not in rows[mid][2]:
        mid += 1
    print("timeline excerpt (us since first kernel; queue; kernel; workgroups):")
    for r in rows[mid:mid + 40]:
        print("  %10.1f .. %10.1f  q%-3s %-28s %7d" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[3], r[2].split("(")[0][-28:], r[4]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
