#!/usr/bin/env python3
"""Build the text summaries kept under profiles/ from one tools/gpu_profile_round.sh session.
    python tools/make_profiles.py r01      (reads gpurun_out/prof_r01/, writes profiles/r01_*.txt)"""
import io
import os
import re
import sqlite3
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocprof_summary as rs  # noqa: E402


def cap(fn, *a):
    b = io.StringIO()
    with redirect_stdout(b):
        fn(*a)
    return b.getvalue()


def pmc_mean(db, kernel_like, counter):
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    r = c.execute("select avg(v), avg(d), count(*) from (select sum(value) v, max(duration) d from counters_collection where %s "
                  "like ? and counter_name = ? group by dispatch_id)" % namecol, ("%" + kernel_like + "%", counter)).fetchone()
    return r


def main():
    R = sys.argv[1] if len(sys.argv) > 1 else "r01"
    O = os.path.join(ROOT, "gpurun_out", "prof_" + R)
    P = os.path.join(ROOT, "profiles")
    K = "k_cnn_forward_wino"
    # 1. kernel stats of bench.py
    with open(os.path.join(P, R + "_bench_kernel_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --mean-plies 33.27\n")
        f.write(cap(rs.kernel_stats, os.path.join(O, "bench", "bench_results.db")))
    with open(os.path.join(P, R + "_bench_kernel_stats_bench_line.txt"), "w") as f:
        f.write("# bench.py JSON line of the same (profiled) run:\n")
        f.write(open(os.path.join(O, "bench_line.json")).read())
    # 2. K3 PMC
    with open(os.path.join(P, R + "_k3_pmc.txt"), "w") as f:
        f.write("# K3 %s: rocprofv3 --pmc <counters>, one pass per counter set, never combined with tracing.\n" % K)
        f.write("# FETCH_SIZE / WRITE_SIZE are in KB per dispatch; gfx950: FETCH_SIZE counts 16-B-per-lane streaming reads at half "
                "their bytes (MI355X_MICROARCH.md, HBM) -> x2\n")
        f.write("## passes on bench.py --steps 2 --warmup 1 (the bench's own launch shape; PMC serialises the dispatches)\n")
        for p in ("bench_fetch", "bench_write"):
            f.write("\n".join(l for l in cap(rs.pmc_stats, os.path.join(O, p, "pmc_results.db")).split("\n")
                              if "cnn" in l or l.startswith("kernel")) + "\n")
        bf = pmc_mean(os.path.join(O, "bench_fetch", "pmc_results.db"), K, "FETCH_SIZE")
        bw = pmc_mean(os.path.join(O, "bench_write", "pmc_results.db"), K, "WRITE_SIZE")
        f.write("# derived: HBM traffic per bench launch = 2 x %.1f KB + %.1f KB = %.2f MB (the 3.4 MB of Winograd-form weights once "
                "per XCD L2 per launch; positions in and outputs out are tens of KB)\n" % (bf[0], bw[0], (2 * bf[0] + bw[0]) / 1024))
        f.write("## passes on tools/k3_sweep.py 6144 (K3 alone, 6144 leaves = 1229 workgroups of 5 leaves per launch)\n")
        for p in ("k3_mfma", "k3_fetch", "k3_write"):
            f.write("### %s\n" % p)
            f.write("\n".join(l for l in cap(rs.pmc_stats, os.path.join(O, p, "pmc_results.db")).split("\n")
                              if "cnn" in l or l.startswith("kernel")) + "\n")
            for l in open(os.path.join(O, p + ".log")):
                if l.startswith("n="):
                    f.write(l)
        db = os.path.join(O, "k3_mfma", "pmc_results.db")
        gui, dur, _ = pmc_mean(db, K, "GRBM_GUI_ACTIVE")
        busy = pmc_mean(db, K, "SQ_VALU_MFMA_BUSY_CYCLES")[0]
        cyc = gui / 8.0
        n_mfma = busy / 32.0
        f.write("# derived: effective clock = GRBM_GUI_ACTIVE/8/duration = %.3f GHz; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                "(1024 SIMDs x %.0f cycles) = %.3f\n" % (cyc / (dur / 1e3) / 1e3, cyc, busy / (1024 * cyc)))
        f.write("# derived: MFMAs per launch = BUSY/32 (v_mfma_f32_16x16x4_f32, 8 passes) = %.0f = 1229 WGs x 4 waves x %.0f; executed "
                "flops = %.1f GFLOP (Winograd domain + 64/60 column padding + output column 7); algorithmic (direct convolution) = "
                "6144 x 41.62 MFLOP = 255.7 GFLOP\n" % (n_mfma, n_mfma / 1229 / 4, n_mfma * 2048 / 1e9))
        kf = pmc_mean(os.path.join(O, "k3_fetch", "pmc_results.db"), K, "FETCH_SIZE")
        kw = pmc_mean(os.path.join(O, "k3_write", "pmc_results.db"), K, "WRITE_SIZE")
        f.write("# derived: HBM traffic per launch = 2 x %.0f KB + %.0f KB = %.1f MB\n" % (kf[0], kw[0], (2 * kf[0] + kw[0]) / 1024))
    # 3. tree + env
    with open(os.path.join(P, R + "_tree_env_pmc.txt"), "w") as f:
        f.write("# tree + env kernels: rocprofv3 -- python3 tools/bench_configs.py k1 ttt nbttt pure unpack   (kernel-trace pass, FETCH_SIZE "
                "pass, WRITE_SIZE pass; KB per dispatch)\n")
        f.write(cap(rs.kernel_stats, os.path.join(O, "tree", "tree_results.db")))
        for p in ("tree_fetch", "tree_write"):
            f.write("### %s\n" % p)
            f.write(cap(rs.pmc_stats, os.path.join(O, p, "pmc_results.db")))
        f.write("### tools/bench_configs.py output of the kernel-trace run\n")
        for l in open(os.path.join(O, "tree_stats.log")):
            if not re.match(r"^[WEI]\d{8}", l) and "amdgpu.ids" not in l:
                f.write(l)
    # 4. the experimental split-fp16 K3 (tools/gpu_profile_split.sh), when its passes exist
    OS = O + "_split"
    if os.path.isdir(OS):
        KS = "k_cnn_forward_split"
        with open(os.path.join(P, R + "_k3_split.txt"), "w") as f:
            f.write("# experimental K3 %s (az_cnn_split.hip, AZ_K3_SPLIT=1 / bench.py --k3 split): fp32 products as 3 fp16 MFMA "
                    "products on split operands, fp32 accumulation\n" % KS)
            f.write("## rocprofv3 --kernel-trace --stats -- python3 bench.py --k3 split --steps 4 --warmup 1 --no-cpu-baseline "
                    "--mean-plies 33.27\n")
            f.write(cap(rs.kernel_stats, os.path.join(OS, "bench", "bench_results.db")))
            f.write("## bench.py JSON line of that run:\n")
            f.write(open(os.path.join(OS, "bench_line.json")).read())
            f.write("## PMC passes on AZ_K3_SPLIT=1 tools/k3_sweep.py 6144 (K3 alone, 1536 workgroups of 4 leaves per launch)\n")
            for p_ in ("k3_mfma", "k3_fetch", "k3_write"):
                f.write("### %s\n" % p_)
                f.write("\n".join(l for l in cap(rs.pmc_stats, os.path.join(OS, p_, "pmc_results.db")).split("\n")
                                  if "cnn" in l or l.startswith("kernel")) + "\n")
                for l in open(os.path.join(OS, p_ + ".log")):
                    if l.startswith("n="):
                        f.write(l)
            db = os.path.join(OS, "k3_mfma", "pmc_results.db")
            gui, dur, _ = pmc_mean(db, KS, "GRBM_GUI_ACTIVE")
            busy = pmc_mean(db, KS, "SQ_VALU_MFMA_BUSY_CYCLES")[0]
            cyc = gui / 8.0
            f.write("# derived: effective clock = GRBM_GUI_ACTIVE/8/duration = %.3f GHz; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                    "(1024 SIMDs x %.0f cycles) = %.3f\n" % (cyc / (dur / 1e3) / 1e3, cyc, busy / (1024 * cyc)))
            kf = pmc_mean(os.path.join(OS, "k3_fetch", "pmc_results.db"), KS, "FETCH_SIZE")
            kw = pmc_mean(os.path.join(OS, "k3_write", "pmc_results.db"), KS, "WRITE_SIZE")
            f.write("# derived: HBM traffic per launch = 2 x %.0f KB + %.0f KB = %.1f MB\n" % (kf[0], kw[0], (2 * kf[0] + kw[0]) / 1024))
    print("wrote profiles/%s_*.txt" % R)


if __name__ == "__main__":
    main()
