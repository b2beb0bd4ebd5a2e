"""One rank of an 8-way split of the 64M box ALONE on the GPU, with the exchange chain on its own stream and a
transport of zero latency: what a rank of an 8-GPU run enqueues per step and WHEN the chain's kernels run relative
to its patch and brick launches.  Transport (HQ_TRACE_TRANSPORT): `loopback` (default; hq_comm_init_loopback: the IPC
transport's kernels, peer stores and flag waits with the rank as its own peer -- all on the device, no host hop) or
`host` (hq_comm_init_host with a callback that zero-fills what it should receive: two stream synchronisations and
two PCIe hops per exchange, the round-3 trace).
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 profiles/tools/rank_alone_trace.py [rank] [steps] [workload]
    python3 profiles/tools/rank_alone_trace.py --analyse <kernel_trace.csv>
(Results of the run are meaningless: the neighbours' records are zeros.)"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(rank, steps, workload="c3"):
    import numpy as np
    import bench
    from hercules_amd import host
    nx, ny, nz, h, dt, freq = bench.WORKLOADS[workload]
    b = host.Box(nx, ny, nz, h, dt, freq, rank=rank, nranks=8)
    u = bench.seeded_field(b.node_ijk, nx, ny)
    s = b.create_solver(tm1=u, tm2=u * 0.999)

    def exchange(recvs, sends, tag):
        for _, buf in recvs:
            buf[:] = 0.0
    if os.environ.get("HQ_TRACE_TRANSPORT", "loopback") == "host":
        s.comm_init_host(exchange)
    else:
        s.comm_init_loopback()
    s.run(steps)
    s.sync()
    if os.environ.get("HQ_TRACE_TIME_STEPS"):          # wall clock per step without a profiler attached
        import time
        n = int(os.environ["HQ_TRACE_TIME_STEPS"])
        for rep in range(3):
            t0 = time.perf_counter()
            s.run(n)
            s.sync()
            print("wall clock, %d steps: %.1f us per step" % (n, (time.perf_counter() - t0) / n * 1e6))
    print("rank %d of 8 alone: %s" % (rank, s.info()))
    s.close()
    b.close()


def analyse(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Queue_Id"]))
    rows.sort()
    bricks = [i for i, r in enumerate(rows) if "hq_k_brick" in r[2]]
    chain_names = ("hq_k_pack", "hq_k_unpack", "hq_k_interface_update", "hq_k_distribute", "hq_k_ipc_wait", "hq_k_iface")
    # steps: from one brick launch to the next; skip the first few
    stats = []
    for a, b in zip(bricks[5:-1], bricks[6:]):
        b0, b1, nxt = rows[a][0], rows[a][1], rows[b][0]
        # a step's chain: the pack behind the interface patches (just ahead of the brick launch) .. the unpack
        chain = sorted(r for r in rows if any(n in r[2] for n in chain_names) and b0 - 40000 <= r[0] < nxt - 40000)
        if len(chain) < 2:
            continue
        gaps = [chain[i + 1][0] - chain[i][1] for i in range(len(chain) - 1)]
        beside = sum(1 for r in chain if b0 <= r[0] < b1)
        stats.append((b1 - b0, sum(r[1] - r[0] for r in chain), chain[-1][1] - chain[0][0], max(gaps), beside, len(chain), nxt - b0))
    n = len(stats)
    if not n:
        print("no steps found")
        return
    mean = [sum(s[k] for s in stats) / n / 1e3 for k in (0, 1, 2, 3, 6)]
    per_step = sorted(s[6] / 1e3 for s in stats)
    print("step length over %d steps: min %.1f  median %.1f  max %.1f us" % (n, per_step[0], per_step[n // 2], per_step[-1]))
    print("%d steps of %.1f us: brick launch %.1f us; the chain's %d kernels take %.1f us in all (its critical path on the "
          "device); first start -> last end %.1f us, longest gap between two of them %.1f us (the transport's round trip); "
          "%.1f of them start while the brick launch is in flight"
          % (n, mean[4], mean[0], stats[0][5], mean[1], mean[2], mean[3], sum(s[4] for s in stats) / n))
    a = bricks[len(bricks) // 2]
    t0 = rows[a][0]
    print("timeline of one step (us relative to the brick launch's start; queue; kernel):")
    for r in rows:
        if t0 - 120000 <= r[0] <= rows[a][1] + 150000:
            print("  %9.1f .. %9.1f  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[3], r[2][:40]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2])
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 30,
            sys.argv[3] if len(sys.argv) > 3 else "c3")
