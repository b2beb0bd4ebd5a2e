"""bench.py's roofline arithmetic on the counter CSVs committed under profiles/r02 (the two rocprofv3 --pmc
passes of the default bench run): HBM bytes per step = (FETCH_SIZE x 2 + WRITE_SIZE) KiB summed over the patch
kernels of a step, as MI355X_MICROARCH.md prescribes for gfx950 -- and the fraction it yields with the committed
bench line's kernel time stays a fraction (<= 1).  No GPU."""
import csv
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R02 = os.path.join(ROOT, "profiles", "r02")


def _per_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = out.setdefault(r["Kernel_Name"].split("(")[0], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return out


def test_traffic_and_fraction_from_the_committed_counter_passes():
    pmc = {"FETCH_SIZE": _per_kernel(os.path.join(R02, "pmc_default_fetch_size.csv"), "FETCH_SIZE"),
           "WRITE_SIZE": _per_kernel(os.path.join(R02, "pmc_default_write_size.csv"), "WRITE_SIZE")}
    line = json.load(open(os.path.join(R02, "bench_default_v13.json")))
    kernel = line["roofline"]["kernel"]
    total, rd, wr, steps = bench.traffic_of(pmc, kernel)
    assert steps == 4 and rd > wr > 0
    # the committed line was computed from these very passes
    assert abs(total - line["roofline"]["traffic"]) <= 1e-6 * total
    nodes = line["config"]["nodes"]
    assert total >= bench.COMPULSORY_BYTES_PER_NODE * nodes * 0.99          # nothing can move less than the compulsory bytes
    assert abs(wr - 24.0 * nodes) <= 0.01 * wr                              # u(t+dt) is written exactly once: 24 B per node
    frac = total / (line["roofline"]["kernel_ms"] * 1e-3) / 1e9 / bench.HBM_PEAK_GBS
    assert 0.0 < frac <= 1.0 and abs(frac - line["roofline"]["frac"]) < 1e-9
    assert line["roofline"]["achieved_basis"] == "measured HBM bytes"
