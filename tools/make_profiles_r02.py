#!/usr/bin/env python3
"""Turn one tools/gpu_profile_r02.sh session (gpurun_out/prof_r02/) into the text files kept under profiles/r02_*.
Runs ON THE GPU BOX right after the session (the rocprofv3 databases are too large to travel back): writes
gpurun_out/r02_txt/*.txt, which are then copied to profiles/ in the repo.
    python3 tools/make_profiles_r02.py"""
import io
import json
import os
import re
import sqlite3
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocprof_summary as rs  # noqa: E402

O = os.path.join(ROOT, "gpurun_out", "prof_r02")
T = os.path.join(ROOT, "gpurun_out", "r02_txt")
KS = "k_cnn_forward_split"


def cap(fn, *a):
    b = io.StringIO()
    with redirect_stdout(b):
        fn(*a)
    return b.getvalue()


def pmc_mean(db, kernel_like, counter):
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    return c.execute("select avg(v), avg(d), count(*) from (select sum(value) v, max(duration) d from counters_collection "
                     "where %s like ? and counter_name = ? group by dispatch_id)" % namecol,
                     ("%" + kernel_like + "%", counter)).fetchone()


def only(text, *keys):
    return "\n".join(l for l in text.split("\n") if l.startswith("kernel") or any(k in l for k in keys)) + "\n"


def logtext(name):
    p = os.path.join(O, name)
    if not os.path.exists(p):
        return "(missing %s)\n" % name
    return "".join(l for l in open(p) if not re.match(r"^[WEI]\d{8}", l) and "amdgpu.ids" not in l)


def bench():
    if not os.path.isdir(os.path.join(O, "bench")):
        return
    with open(os.path.join(T, "r02_bench_kernel_stats.txt"), "w") as f:
        cmd = "python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-leg --mean-plies 33.1 --preroll 30"
        f.write("# rocprofv3 --kernel-trace --stats -- %s   (headline kernel: split-fp16 K3, 3 positions per workgroup)\n" % cmd)
        db = os.path.join(O, "bench", "bench_results.db")
        f.write(cap(rs.kernel_stats, db))
        f.write("## one round of a cohort on its HIP stream: K3 launch -> tree step -> next K3 launch, whole trace (mostly the pre-roll, "
                "whose staggered start parks part of the slots: smaller launches)\n")
        f.write(cap(rs.chain_stats, db))
        f.write("## the same over the last 12 % of the trace = the 4 timed steps (all slots live)\n")
        f.write(cap(rs.chain_stats, db, 0.12))
        f.write("## bench.py JSON line of this (profiled) run:\n" + open(os.path.join(O, "bench_line.json")).read())
        f.write("\n# the same with the fp32 Winograd K3 (bench.py --k3 wino)\n")
        db = os.path.join(O, "bench_wino", "bench_results.db")
        f.write(cap(rs.kernel_stats, db))
        f.write(cap(rs.chain_stats, db, 0.12))
        f.write("## bench.py JSON line of that run:\n" + open(os.path.join(O, "bench_wino_line.json")).read())


def k3():
    if not os.path.isdir(os.path.join(O, "k3_mfma")):
        return
    with open(os.path.join(T, "r02_k3_pmc.txt"), "w") as f:
        f.write("# K3 %s<3, 4> (az_cnn_split.hip): rocprofv3 --pmc <counters>, one pass per counter set, never combined with "
                "tracing.\n# FETCH_SIZE / WRITE_SIZE are KB per dispatch; gfx950 counts 16-B-per-lane streaming reads at half their "
                "bytes (MI355X_MICROARCH.md, HBM) -> x2\n" % KS)
        if os.path.isdir(os.path.join(O, "bench_fetch")):
            f.write("## passes on bench.py --steps 2 --warmup 1 --preroll 8 (the bench's own launch shape; PMC serialises the dispatches)\n")
            for p in ("bench_fetch", "bench_write"):
                f.write(only(cap(rs.pmc_stats, os.path.join(O, p, "pmc_results.db")), "cnn"))
            bf = pmc_mean(os.path.join(O, "bench_fetch", "pmc_results.db"), KS, "FETCH_SIZE")
            bw = pmc_mean(os.path.join(O, "bench_write", "pmc_results.db"), KS, "WRITE_SIZE")
            f.write("# derived: HBM traffic per bench launch = 2 x %.1f KB + %.1f KB = %.2f MB\n" %
                    (bf[0], bw[0], (2 * bf[0] + bw[0]) / 1024))
        f.write("## passes on tools/k3_sweep.py 6144 (K3 alone: 6144 leaves = 2048 workgroups of 3 positions per launch)\n")
        for p in ("k3_mfma", "k3_insts", "k3_fetch", "k3_write"):
            f.write("### %s\n" % p)
            f.write(only(cap(rs.pmc_stats, os.path.join(O, p, "pmc_results.db")), "cnn"))
            f.write("".join(l for l in logtext(p + ".log").split("\n") if l.startswith("n=")) + "\n")
        db = os.path.join(O, "k3_mfma", "pmc_results.db")
        gui, dur, _ = pmc_mean(db, KS, "GRBM_GUI_ACTIVE")
        busy = pmc_mean(db, KS, "SQ_VALU_MFMA_BUSY_CYCLES")[0]
        wave = pmc_mean(db, KS, "SQ_WAVE_CYCLES")[0]
        wi = pmc_mean(db, KS, "SQ_WAIT_INST_ANY")[0]
        cyc = gui / 8.0
        f.write("# derived: effective clock = GRBM_GUI_ACTIVE/8/duration = %.3f GHz; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                "(1024 SIMDs x %.0f cycles) = %.3f; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f\n" %
                (cyc / (dur / 1e3) / 1e3, cyc, busy / (1024 * cyc), wi / wave))
        f.write("# derived: MFMAs per launch = BUSY/32 = %.0f (v_mfma_f32_32x32x16_f16) = executed %.1f GFLOP; algorithmic x 3 products = "
                "6144 x 124.86 MFLOP = 767.1 GFLOP (the rest: 128 columns for 126 cells, the heads' 16-row fragment)\n" %
                (busy / 32.0, busy / 32.0 * 32768 / 1e9))
        kf = pmc_mean(os.path.join(O, "k3_fetch", "pmc_results.db"), KS, "FETCH_SIZE")
        kw = pmc_mean(os.path.join(O, "k3_write", "pmc_results.db"), KS, "WRITE_SIZE")
        f.write("# derived: HBM traffic per launch = 2 x %.0f KB + %.0f KB = %.1f MB\n" % (kf[0], kw[0], (2 * kf[0] + kw[0]) / 1024))
    with open(os.path.join(T, "r02_k3_split_breakdown.txt"), "w") as f:
        f.write("# where the split-fp16 K3's time goes (un-profiled runs of the same session)\n")
        f.write("## tools/k3_sweep.py with AZ_K3_SPLIT_CFG = <positions per workgroup><waves>: 34 is the product configuration\n")
        f.write(logtext("k3_cfgs.log"))
        f.write("## operand-removal builds (tools/k3_variants.sh; results are garbage, timing only): NOW = no weight loads, NOB = no LDS "
                "operand reads, LOADONLY = every operand fetched as usual but the MFMAs multiply constants\n")
        f.write(logtext("k3_variants.log"))
        f.write("## in-kernel shader-clock stamps (tools/k3s_stamps.py, build -DAZ_K3S_DBG_TS)\n")
        f.write(logtext("k3_stamps.log"))
        f.write("## the same stamps and the launch time of the operand-removal builds (constant operands: no weight loads / no LDS "
                "operand reads): the workgroup takes the SAME number of cycles, the launch is shorter -> the clock is higher: the "
                "kernel runs against the chip's power limit, not against a pipe\n")
        f.write(logtext("k3_stamps_variants.log"))
        f.write("## tools/ubench/coexec.hip: one wave's MFMAs and ANOTHER wave's VALU work on the same SIMD\n")
        f.write(logtext("coexec.log"))
        f.write("## tools/ubench/fillers.hip: VALU fillers of the SAME wave between MFMAs\n")
        f.write(logtext("fillers.log"))
        f.write("## tools/ubench/mfma_power.hip: every SIMD issuing v_mfma_f32_32x32x16_f16 back to back -- the sustained clock and rate "
                "depend on the operand DATA\n")
        f.write(logtext("mfma_power.log"))
        f.write("## tools/ubench/valu_rate.hip: issue cost of the epilogue's VALU instructions with no MFMA in flight\n")
        f.write(logtext("valu_rate.log"))


def tree():
    if not os.path.isdir(os.path.join(O, "tree")):
        return
    with open(os.path.join(T, "r02_tree_env_pmc.txt"), "w") as f:
        f.write("# tree + env kernels: rocprofv3 -- python3 tools/bench_configs.py k1 ttt nbttt pure unpack   (kernel-trace pass, FETCH_SIZE "
                "pass, WRITE_SIZE pass, SQ pass; KB per dispatch)\n")
        f.write(cap(rs.kernel_stats, os.path.join(O, "tree", "tree_results.db")))
        for p in ("tree_fetch", "tree_write", "tree_sq"):
            f.write("### %s\n" % p)
            f.write(cap(rs.pmc_stats, os.path.join(O, p, "pmc_results.db")))
        f.write("### tools/bench_configs.py output of the kernel-trace run\n")
        f.write(logtext("tree_stats.log"))
        f.write("### games per wave (AZ_TREE_GPW) against the fused search's launch time: the same games on more waves\n")
        for l in logtext("tree_gpw.log").split("\n"):
            if l.startswith("=="):
                f.write(l + "\n")
            elif l.startswith("{"):
                d = json.loads(l)
                f.write("  %-24s avg_launch_ms %.3f  playouts/s (wall) %.3e  mean depth %.2f\n" %
                        (d["bench"], d["avg_launch_ms"], d["playouts_per_s"], d["mean_depth"]))


if __name__ == "__main__":
    os.makedirs(T, exist_ok=True)
    bench()
    k3()
    tree()
    print("wrote", sorted(os.listdir(T)))
