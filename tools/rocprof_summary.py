#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) result into the text summaries kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.txt
    python tools/rocprof_summary.py --pmc gpurun_out/pmc/x_results.db > profiles/r01_pmc_fetch.txt

--kernel-trace --stats runs: per-kernel calls / total / average / min / max duration (from the dispatch table).
--pmc runs: per-kernel sum and per-dispatch mean of every collected counter.
"""
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = list(c.execute(
        "select k.name, count(*), sum(k.end - k.start), avg(k.end - k.start), min(k.end - k.start), "
        "max(k.end - k.start) from kernels k group by k.name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats   (durations in microseconds)")
    print("%-70s %9s %14s %11s %11s %11s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, n, tot, avg, mn, mx in rows:
        short = name.split("(")[0][-70:]
        print("%-70s %9d %14.1f %11.2f %11.2f %11.2f %6.2f%%" % (short, n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                              100.0 * tot / total))
    # Dispatches of one kernel issued on different HIP streams (the engine's cohorts) execute concurrently: the time
    # the kernel occupies the GPU is the UNION of its [start, end] intervals, not the sum of the durations.
    print("# busy = union of the dispatch intervals per kernel (equals total_us when dispatches never overlap)")
    print("%-70s %14s %14s %9s" % ("kernel", "busy_us", "total_us", "overlap"))
    for name, n, tot, *_ in rows[:6]:
        busy, last = 0, 0
        for a, b in c.execute("select start, end from kernels where name = ? order by start", (name,)):
            lo = max(a, last)
            if b > lo:
                busy += b - lo
                last = b
        print("%-70s %14.1f %14.1f %8.2fx" % (name.split("(")[0][-70:], busy / 1e3, tot / 1e3, tot / max(busy, 1)))


def chain_stats(db, tail=1.0):
    """per HIP stream (= engine cohort): how one round of the rolling self-play spends its time.  A cohort's round is the
    chain K3 launch -> tree step -> next K3 launch; the gaps are the time between a kernel's end and its successor's start.
    tail < 1: only the dispatches of the last `tail` of the traced time span (the bench's timed window: the untimed pre-roll
    before it runs with part of the slots parked, i.e. with smaller K3 launches)."""
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute("select * from kernels limit 1").description]
    key = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    print("# columns of the kernel view:", ", ".join(cols))
    if key is None:
        return
    rows = list(c.execute("select %s, name, start, end from kernels order by start" % key))
    if tail < 1.0 and rows:
        t_lo, t_hi = rows[0][2], max(r[3] for r in rows)
        cut = t_hi - tail * (t_hi - t_lo)
        rows = [r for r in rows if r[2] >= cut]
        print("# only dispatches that start in the last %.0f %% of the traced span (%d of them)" % (100 * tail, len(rows)))
    by = {}
    for q, name, a, b in rows:
        by.setdefault(q, []).append((a, b, "k3" if "cnn_forward" in name else ("tree" if "tree_step" in name else "other")))
    print("# per %s: kernel durations and the gaps inside the K3 -> tree step -> K3 chain (microseconds)" % key)
    print("%-10s %8s %10s %10s %12s %12s %12s" % (key, "rounds", "k3_avg", "tree_avg", "k3->tree", "tree->k3", "round"))
    for q, seq in sorted(by.items()):
        k3 = [b - a for a, b, t in seq if t == "k3"]
        tr = [b - a for a, b, t in seq if t == "tree"]
        g1, g2, rnd = [], [], []
        last_k3_start = None
        for (a0, b0, t0), (a1, b1, t1) in zip(seq, seq[1:]):
            if t0 == "k3" and t1 == "tree":
                g1.append(a1 - b0)
            if t0 == "tree" and t1 == "k3":
                g2.append(a1 - b0)
        starts = [a for a, b, t in seq if t == "k3"]
        rnd = [y - x for x, y in zip(starts, starts[1:])]
        if len(k3) < 10:
            continue
        med = lambda v: sorted(v)[len(v) // 2] / 1e3 if v else 0.0
        avg = lambda v: sum(v) / len(v) / 1e3 if v else 0.0
        print("%-10s %8d %10.1f %10.1f %12.1f %12.1f %12.1f   (medians: k3 %.1f tree %.1f gaps %.1f %.1f round %.1f)" %
              (q, len(k3), avg(k3), avg(tr), avg(g1), avg(g2), avg(rnd), med(k3), med(tr), med(g1), med(g2), med(rnd)))


def pmc_stats(db):
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    print("# rocprofv3 --pmc   columns available:", ", ".join(cols))
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    # a counter comes as several rows per dispatch (one per hardware instance): add them up per dispatch first
    rows = list(c.execute(
        "select kn, cn, count(*), sum(v), avg(v), avg(d) from (select %s kn, counter_name cn, dispatch_id, sum(value) v, "
        "max(duration) d from counters_collection group by 1, 2, 3) group by 1, 2 order by 1, 2" % namecol))
    print("%-60s %-28s %9s %20s %18s %12s" % ("kernel", "counter", "dispatches", "sum", "mean/dispatch", "avg_dur_us"))
    for name, ctr, n, s, a, d in rows:
        print("%-60s %-28s %9d %20.1f %18.1f %12.2f" % (name.split("(")[0][-60:], ctr, n, s, a, (d or 0) / 1e3))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--pmc" in sys.argv:
        pmc_stats(args[0])
    elif "--chains" in sys.argv:
        chain_stats(args[0])
    else:
        kernel_stats(args[0])
