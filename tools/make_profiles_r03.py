#!/usr/bin/env python3
"""Turn one tools/gpu_profile_r03.sh session (gpurun_out/prof_r03/) into the text files kept under profiles/r03_*.
Runs ON THE GPU BOX right after the session (the rocprofv3 databases are too large to travel back): writes
gpurun_out/r03_txt/*.txt, which are then copied to profiles/ in the repo.
    python3 tools/make_profiles_r03.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_profiles_r02 as m2  # noqa: E402
import rocprof_summary as rs  # noqa: E402

O = os.path.join(ROOT, "gpurun_out", "prof_r03")
T = os.path.join(ROOT, "gpurun_out", "r03_txt")
m2.O = O
cap, pmc_mean, only, logtext = m2.cap, m2.pmc_mean, m2.only, m2.logtext
SQ = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_WAVES",
      "GRBM_GUI_ACTIVE")


def sq_row(f, label, db, kernel_like):
    print("  sq_row", label, flush=True)
    v = {c: pmc_mean(db, kernel_like, c) for c in SQ}
    if not v["SQ_WAVE_CYCLES"] or not v["SQ_WAVE_CYCLES"][0]:
        f.write("%-44s (no dispatches of %s)\n" % (label, kernel_like))
        return
    w = v["SQ_WAVE_CYCLES"][0]
    dur = v["SQ_WAVE_CYCLES"][1] / 1e3
    f.write("%-44s launches %4d  avg %9.1f us (profiled)  waves %8.0f  SQ_WAIT_ANY/WAVE_CYCLES %.3f  SQ_WAIT_INST_ANY/.. %.3f  "
            "SQ_ACTIVE_INST_ANY/.. %.3f  VALU insts per wave %.0f  clock %.2f GHz\n" % (
                label, v["SQ_WAVE_CYCLES"][2], dur, v["SQ_WAVES"][0], v["SQ_WAIT_ANY"][0] / w, v["SQ_WAIT_INST_ANY"][0] / w,
                v["SQ_ACTIVE_INST_ANY"][0] / w, v["SQ_INSTS_VALU"][0] / max(v["SQ_WAVES"][0], 1),
                v["GRBM_GUI_ACTIVE"][0] / 8.0 / max(dur, 1e-9) / 1e3))


def bench():
    if not os.path.isdir(os.path.join(O, "bench")):
        return
    with open(os.path.join(T, "r03_bench_kernel_stats.txt"), "w") as f:
        cmd = "python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-leg --mean-plies 33.1 --preroll 12"
        f.write("# rocprofv3 --kernel-trace --stats -- %s   (headline: split-fp16 K3 + evaluation cache, 4 cohorts)\n" % cmd)
        db = os.path.join(O, "bench", "bench_results.db")
        f.write(cap(rs.kernel_stats, db))
        f.write("## one round of a cohort on its HIP stream: K3 launch -> tree step -> next K3 launch, whole trace\n")
        f.write(cap(rs.chain_stats, db))
        f.write("## the same over the last 20 % of the trace = the timed steps (all slots live)\n")
        f.write(cap(rs.chain_stats, db, 0.2))
        f.write("## bench.py JSON line of this (profiled) run:\n" + open(os.path.join(O, "bench_line.json")).read())
        f.write("\n# the same with the evaluation cache off (bench.py --eval-cache -1): every non-terminal leaf goes through K3\n")
        db = os.path.join(O, "bench_off", "bench_results.db")
        f.write(cap(rs.kernel_stats, db))
        f.write(cap(rs.chain_stats, db, 0.2))
        f.write("## bench.py JSON line of that run:\n" + open(os.path.join(O, "bench_off_line.json")).read())
        db = os.path.join(O, "bench_noev", "bench_results.db")
        if os.path.exists(db):
            f.write("\n# the headline run again with bench.py --no-profile (no HIP events around the kernels): the gaps between a stream's kernels\n")
            f.write(cap(rs.chain_stats, db, 0.2))
            f.write("## bench.py JSON line of that run:\n" + open(os.path.join(O, "bench_noev_line.json")).read())
    with open(os.path.join(T, "r03_k3_pmc.txt"), "w") as f:
        f.write("# rocprofv3 --pmc passes on bench.py --steps 1 --warmup 1 --preroll 4 (evaluation cache on: K3 launches of ~260 leaves, "
                "tree steps of 512 games with up to 7 playouts per game); one pass per counter set, never combined with tracing.\n"
                "# FETCH_SIZE / WRITE_SIZE are KB per dispatch; gfx950 counts 16-B-per-lane streaming reads at half their bytes "
                "(MI355X_MICROARCH.md, HBM) -> x2 for the K3 weight stream\n")
        for p in ("bench_fetch", "bench_write"):
            f.write(only(cap(rs.pmc_stats, os.path.join(O, p, "pmc_results.db")), "cnn", "tree_step"))
        for kern, label in (("k_cnn_forward_split", "K3"), ("k_tree_step", "tree step")):
            bf = pmc_mean(os.path.join(O, "bench_fetch", "pmc_results.db"), kern, "FETCH_SIZE")
            bw = pmc_mean(os.path.join(O, "bench_write", "pmc_results.db"), kern, "WRITE_SIZE")
            f.write("# derived: %s HBM traffic per bench launch = 2 x %.1f KB + %.1f KB = %.2f MB (fetch counted once: %.2f MB)\n" %
                    (label, bf[0], bw[0], (2 * bf[0] + bw[0]) / 1024, (bf[0] + bw[0]) / 1024))
        f.write("## SQ counters of the same workload\n")
        db = os.path.join(O, "bench_sq", "pmc_results.db")
        sq_row(f, "k_cnn_forward_split<3,4> (bench)", db, "k_cnn_forward_split")
        sq_row(f, "k_tree_step<C4,512,2> (bench)", db, "k_tree_step")


def cfg3():
    if not os.path.isdir(os.path.join(O, "cfg3")):
        return
    with open(os.path.join(T, "r03_cfg3_kernel_stats.txt"), "w") as f:
        cmd = "python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-second-leg --mean-plies 33.2 --preroll 10"
        f.write("# BASELINE configs[3], one GPU's share (16 384 slots, 2^30-entry evaluation cache, 4 cohorts of 4 096 games, time budget 296 us)\n")
        f.write("# rocprofv3 --kernel-trace --stats -- %s\n" % cmd)
        db = os.path.join(O, "cfg3", "bench_results.db")
        f.write(cap(rs.kernel_stats, db))
        f.write("## one round of a cohort on its HIP stream over the last 20 % of the trace (the timed steps)\n")
        f.write(cap(rs.chain_stats, db, 0.2))
        f.write("## bench.py JSON line of this (profiled) run:\n" + open(os.path.join(O, "cfg3_line.json")).read())


def cfg4():
    if not os.path.isdir(os.path.join(O, "cfg4")):
        return
    with open(os.path.join(T, "r03_cfg4_pmc.txt"), "w") as f:
        f.write("# BASELINE configs[4]: python3 bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --mean-plies 41.1 --preroll 48\n"
                "# (NBTTT, 16 384 games x 800 playouts per launch of k_search_fused<NBTTT, 4>, 153 GB node pool)\n")
        f.write(cap(rs.kernel_stats, os.path.join(O, "cfg4", "bench_results.db")))
        f.write("## bench.py JSON line of this (profiled) run:\n" + open(os.path.join(O, "cfg4_line.json")).read())
        for p in ("cfg4_fetch", "cfg4_write"):
            f.write(only(cap(rs.pmc_stats, os.path.join(O, p, "pmc_results.db")), "search_fused"))
        bf = pmc_mean(os.path.join(O, "cfg4_fetch", "pmc_results.db"), "k_search_fused", "FETCH_SIZE")
        bw = pmc_mean(os.path.join(O, "cfg4_write", "pmc_results.db"), "k_search_fused", "WRITE_SIZE")
        f.write("# derived: HBM traffic per launch = FETCH %.1f MB (x2 if the node reads count as 16-B streaming: %.1f MB) + WRITE %.1f MB\n"
                % (bf[0] / 1024, 2 * bf[0] / 1024, bw[0] / 1024))
        sq_row(f, "k_search_fused<NBTTT,4> 16384 x 800 (bench)", os.path.join(O, "cfg4_sq", "pmc_results.db"), "k_search_fused")


def tree():
    if not os.path.exists(os.path.join(O, "tree_gpw.log")):
        return
    with open(os.path.join(T, "r03_tree_env_pmc.txt"), "w") as f:
        f.write("# fused / pure tree searches: games per wave (AZ_TREE_GPW) against the number of games, un-profiled launch times\n"
                "# (tools/gpw_sweep.sh; gpw = games per 64-lane wave: TTT / NBTTT lane groups of 16 -> 4 = full, Connect4 groups of 8 -> 8 = full)\n")
        f.write(logtext("tree_gpw.log"))
        f.write("## SQ counters per row in the saturated regime (16 384 games; rocprofv3 --pmc, one run per games-per-wave value)\n")
        for w in (4, 2):
            db = os.path.join(O, "tree_sq_gpw%d" % w, "pmc_results.db")
            if os.path.exists(db):
                sq_row(f, "TTT 16384 x 100   gpw %d" % w, db, "k_search_fused<az::TTT")
                sq_row(f, "NBTTT 16384 x 800 gpw %d" % w, db, "k_search_fused<az::NBTTT")
        for w in (8, 4):
            db = os.path.join(O, "pure_sq_gpw%d" % w, "pmc_results.db")
            if os.path.exists(db):
                sq_row(f, "pure C4 16384 x 1000 gpw %d" % w, db, "k_search_pure")
        if os.path.isdir(os.path.join(O, "tree")):
            f.write("## kernel trace + HBM passes: python3 tools/bench_configs.py k1 ttt nbttt pure unpack at 16 384 games (default games per wave)\n")
            f.write(cap(rs.kernel_stats, os.path.join(O, "tree", "tree_results.db")))
            for p in ("tree_fetch", "tree_write"):
                f.write("### %s\n" % p)
                f.write(cap(rs.pmc_stats, os.path.join(O, p, "pmc_results.db")))
            f.write("### tools/bench_configs.py output of the kernel-trace run\n")
            f.write(logtext("tree_stats.log"))


def shape():
    if not os.path.exists(os.path.join(O, "mfma_shape.log")):
        return
    with open(os.path.join(T, "r03_k3_shape.txt"), "w") as f:
        f.write("# tools/ubench/mfma_shape.hip: v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 on the same 64 x 64 wave tile, same\n"
                "# data, every SIMD issuing back to back (two passes over the table)\n")
        f.write(logtext("mfma_shape.log"))
        f.write("## tools/tree_step_alone.py: the rolling tree step with every leaf answered by the evaluation cache (no K3 beside it)\n")
        f.write(logtext("tree_alone.log"))


if __name__ == "__main__":
    os.makedirs(T, exist_ok=True)
    for part in (bench, cfg3, cfg4, tree, shape):
        print("make_profiles_r03:", part.__name__, flush=True)
        part()
    print("wrote", sorted(os.listdir(T)))
