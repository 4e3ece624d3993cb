#!/usr/bin/env python3
"""bench.py -- the self-play hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without torchrun: this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on, per GPU): Connect4 self-play,
2048 concurrent games, 500 playouts per move, the reference's real CNN checkpoint evaluated on whole leaf batches
by the fused K3 kernel (product form: fp32 products as three fp16 MFMA products on split operands, fp32 accumulation,
az_cnn_split.hip; --k3 wino: Winograd on fp32 MFMA; the other form is measured as a second leg of the same line),
Dirichlet alpha 0.7 at every expansion, random mirror augmentation, temperature 1.0 for 10 plies then 1e-3, tree reuse,
finished slots refilled with fresh games (Connect4/config.py:1-14).
A "step" = 500 rounds of every game slot, a round = CNN on the parked leaf batch -> tree step (expand/backup, and for
the games whose move is complete: move selection, env step, trajectory record, refill; then the walk to the next leaf).
In the default rolling mode (az_selfplay_rounds) the games advance individually and every round evaluates one
non-terminal leaf per live game; --mode lockstep runs one ply of every slot per step (az_selfplay_step), the form the
parity tests pin against the reference; both produce the same games (tests/test_gpu_cnn.py).  Finished trajectories
are packed, all-gathered over RCCL (N>1) and expanded into ReplayBuffer-shaped tensors inside the timed region.

Other first-class workloads (same contract, same line format; the default stays configs[2] so that the driver's N = 1
scaling point and its bench line are the same measurement):
  --config 3   BASELINE configs[3], one GPU's share: the same Connect4 CNN self-play with 16 384 slots per GPU (38.5 GB
               node pool); with --gpus N the N-GPU job of configs[3]
  --config 4   BASELINE configs[4]: nine-board tic-tac-toe self-play, 800 playouts per move, hashed evaluator (the reference
               has no usable NBTTT network), 2 random opening plies per game drawn from the game id, c_puct 1.5, tree
               reuse, node pool sized for whole games (16 384 slots = 153 GB of the 288 GB); a step = one ply of every
               slot = ONE fused-search launch (k_search_fused<NBTTT>) + the ply end; roofline = that kernel, HBM class
(AZ_BENCH_CONFIG=3|4 in the environment selects them when the command line cannot.)

value (games/s) = games that FINISH inside the timed window / wall time.  So that a window of a few game
lengths means something, an untimed pre-roll de-synchronises the slots first: the first games start staggered over one
mean game length (az_set_active) and play on until as many games have finished as there are slots, so the window sees
slots at every game age; `value_steady_state_estimate` = slot-plies per second / mean plies per finished game is
reported beside it (the two agree in steady state).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Hardware queues of this process (read by the HIP runtime when it starts, i.e. before torch is imported in main()): the
# engine's four cohort streams, torch's stream and RCCL's streams should not have to share the default 4 -- streams on
# one queue run their kernels one after the other.  Measured under torch.distributed.run: 279 ms per step with 4 queues
# (engine created first), 264 with 8; a plain single-process run is the same with either.
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

FLOP_PER_LEAF = 41.62e6          # MAC = 2 flops, stem + 3 blocks + heads
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
# K3 computes its 3x3 convolutions as Winograd F(2x2,3x3) products: the MFMA pipe EXECUTES 25.79 MFLOP per leaf
# (15 740 v_mfma_f32_16x16x4_f32 per wave x 4 waves x 2 048 flop / 5 leaves per workgroup; profiles/r01_k3_pmc.txt), while
# the roofline's `achieved` stays priced in ALGORITHMIC (direct-convolution) flops as SURVEY 8d defines it, so `frac` can
# exceed the share of the MFMA pipe that is busy (reported next to it as executed_frac).
EXEC_FLOP_PER_LEAF = 15740 * 4 * 2048 / 5.0
# HBM bytes per K3 launch of this workload from the PMC passes of tools/gpu_profile_round.sh (profiles/r01_k3_pmc.txt):
# FETCH_SIZE 13544.8 KB x 2 (gfx950 counts 16-byte streaming loads at half their bytes) + WRITE_SIZE 33.4 KB.  It is the
# 3.4 MB of Winograd-form weights fetched once per XCD L2 per launch; positions in and outputs out are ~25 KB.
K3_TRAFFIC_BYTES_PER_LAUNCH = (2 * 13544.8 + 33.4) * 1024
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
# What the fp16 matrix pipes SUSTAIN on this chip is set by its power limit and depends on the operand data
# (tools/ubench/mfma_power.hip, every SIMD issuing v_mfma_f32_32x32x16_f16 back to back; profiles/r02_k3_split_breakdown.txt):
# zeros 1.9 PFLOP/s at 1.84 GHz, random mantissas 1.46-1.50, values spread like K3's operands 1.31-1.37 at 1.27-1.33 GHz.
# roofline.peak stays the data-sheet figure; the measured ceiling is reported beside it.
SUSTAINED_F16_MFMA_TFLOPS_REAL_DATA = 1340.0
SPLIT_FLOP_PER_LEAF = 3 * FLOP_PER_LEAF   # az_cnn_split.hip: every fp32 product = 3 fp16 MFMA products (hi*hi + hi*lo + lo*hi)
# HBM bytes per launch of the split kernel in THIS workload (PMC passes on bench.py itself, profiles/r03_k3_pmc.txt:
# FETCH_SIZE 8079.5 KB x 2 + WRITE_SIZE 17.7 KB; round 2, without the evaluation cache: 8271.7 KB x 2 + 18.1 KB): the 2.0 MB of
# split weights once per XCD L2 per launch, whatever the launch's size; positions in and outputs out are ~12 KB of it
K3_SPLIT_TRAFFIC_BYTES_PER_LAUNCH = (2 * 8144.8 + 17.4) * 1024
K3_SPLIT_TRAFFIC_SOURCE = "rocprofv3 PMC passes on bench.py, profiles/r03_k3_pmc.txt"
TREE_STEP_TRAFFIC_BYTES_PER_LAUNCH = (2 * 1084.3 + 463.8) * 1024   # same passes: k_tree_step<C4, 512, 2>, 512 games per launch
DEFAULT_K3 = "split"             # the headline kernel = the library's product kernel
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# HBM bytes per k_search_fused<NBTTT> launch at 16 384 games x 800 playouts from the PMC passes of tools/gpu_profile_r03.sh
# (profiles/r03_tree_env_pmc.txt); None until measured for the shape in use
CFG4_TRAFFIC_BYTES_PER_LAUNCH = (2 * 742.6 + 3142.4) * 2**20   # FETCH_SIZE x 2 (16-byte-per-lane node reads) + WRITE_SIZE
CFG4_TRAFFIC_SOURCE = ("rocprofv3 PMC passes on bench.py --config 4 at 16 384 games x 800 playouts, profiles/r03_cfg4_pmc.txt; about half "
                       "of the algorithmic bytes: the upper tree levels are read from cache")
REF_PY_GAMES_PER_S = 0.0243      # BASELINE.md section 2 (reference Python, 1 thread, build container) -- not measured here


def _cpu_moves(orc, net, n_playout, budget_s, seed):
    """whole moves of Connect4 self-play games with the CPU oracle until the budget is used -> (moves, games, playouts)"""
    t = orc.MCTS("Connect4", c_init=1.25, n_playout=n_playout, eval_kind=orc.EVAL_CNN, net=net)
    env = orc.Env("Connect4")
    t0 = time.perf_counter()
    rng = np.random.default_rng(seed)
    moves = ply = games = 0
    while time.perf_counter() - t0 < budget_s:
        if env.done():  # next game of the sample
            env = orc.Env("Connect4")
            t.reset()
            ply = 0
            games += 1
        acts, vis = t.get_action_visits(env)
        # sample the move from the visit distribution for the first 10 plies like the GPU workload (temp 1.0)
        p = np.asarray(vis, dtype=np.float64)
        a = acts[int(rng.choice(len(acts), p=p / p.sum()))] if ply < 10 else acts[int(np.argmax(vis))]
        t.prune_root(a)
        env.step(a)
        moves += 1
        ply += 1
    return moves, games + 1, t.stat_playouts


def cpu_baseline(blob, n_playout, mean_len, budget_s=15.0, multi_budget_s=None):
    """The CPU oracle (C port of the reference algorithm: sequential single-game MCTS, one batch-1 fp32 CNN forward
    per playout) timed on this box's host on a bounded sample: whole moves of self-play games until the budget is used.
    `value` is ONE thread (what the reference's single-threaded Python loop maps to); `all_cores` runs one
    independent game sequence per core of this job's CPU share (the C calls release the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    if multi_budget_s is None:
        multi_budget_s = 0.55 * budget_s
    net = orc.CNN(blob)
    t0 = time.perf_counter()
    moves, games, sims = _cpu_moves(orc, net, n_playout, budget_s, 0)
    dt = time.perf_counter() - t0
    sims_s = sims / dt
    out = dict(value=sims_s / (n_playout * mean_len), unit="games/s", cores=1, kind="port", sims_per_s=sims_s,
               sample="%d moves x %d playouts over %d Connect4 self-play game(s) (%.1f s), oracle/az_oracle*.c, 1 thread, "
                      "no Dirichlet/flips; games/s = sims/s / (%d * %.1f plies)" % (moves, n_playout, games, dt, n_playout, mean_len))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", "16")), 16))   # the box's CPU share, not the machine
    if multi_budget_s > 0 and cores > 1:
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            res = list(ex.map(lambda i: _cpu_moves(orc, net, n_playout, multi_budget_s, 100 + i), range(cores)))
        dtm = time.perf_counter() - t0
        sims_m = sum(r[2] for r in res) / dtm
        out["all_cores"] = dict(value=sims_m / (n_playout * mean_len), unit="games/s", cores=cores, sims_per_s=sims_m,
                                sample="%d threads x independent game sequences, %d moves in %.1f s" %
                                       (cores, sum(r[0] for r in res), dtm))
    return out


def cpu_baseline_nbttt(n_playout, mean_len, eval_seed, budget_s=12.0):
    """cpu_baseline leg of --config 4: the C port's sequential search on nine-board tic-tac-toe with the same hashed
    evaluator, whole moves of self-play games from seeded 2-ply openings on ONE host core until the budget is used."""
    from oracle import oracle as orc
    t = orc.MCTS("NBTTT", c_init=1.5, n_playout=n_playout, eval_kind=orc.EVAL_HASH, eval_seed=eval_seed)
    rng = np.random.default_rng(0)

    def opening():
        b, a1 = int(rng.integers(0, 9)), int(rng.integers(0, 9))
        return orc.Env.nbttt_open(b, a1, int(rng.integers(0, 8)))
    env = opening()
    t0 = time.perf_counter()
    moves, games = 0, 1
    while time.perf_counter() - t0 < budget_s:
        if env.done():
            env = opening()
            t.reset()
            games += 1
        acts, vis = t.get_action_visits(env)
        a = acts[int(np.argmax(vis))]
        t.prune_root(a)
        env.step(a)
        moves += 1
    dt = time.perf_counter() - t0
    sims_s = t.stat_playouts / dt
    return dict(value=sims_s / (n_playout * mean_len), unit="games/s", cores=1, kind="port", sims_per_s=sims_s,
                sample="%d moves x %d playouts over %d NBTTT self-play game(s) (%.1f s), oracle/az_oracle.c, hashed evaluator, "
                       "1 thread; games/s = sims/s / (%d * %.1f plies)" % (moves, n_playout, games, dt, n_playout, mean_len))


def cpu_baseline_pure(n_playout=1000, searches=6):
    """cpu_baseline leg of the pure-MCTS measurement (tools/bench_configs.py pure): the C port of the
    reference's UCT + random-rollout search, root searches of the empty Connect4 board on one host core."""
    from oracle import oracle as orc
    t = orc.PureMCTS("Connect4", 1.0, n_playout, seed=1)
    c0 = time.perf_counter()
    for _ in range(searches):
        t.reset()
        t.get_action(orc.Env("Connect4"))
    cdt = time.perf_counter() - c0
    return {"playouts_per_s": searches * n_playout / cdt, "cores": 1, "kind": "port",
            "sample": "%d root searches x %d playouts, oracle/az_oracle_pure.c" % (searches, n_playout)}


def _box_info(torch, local):
    """what the numbers were measured on (GPU, clocks, host cores), read outside the timed region"""
    info = {"host_cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()}
    try:
        pr = torch.cuda.get_device_properties(local)
        info.update(gpu=pr.name, arch=getattr(pr, "gcnArchName", ""), cus=pr.multi_processor_count,
                    hbm_gib=round(pr.total_memory / 2**30, 1))
        if getattr(pr, "clock_rate", 0):
            info["max_sclk_mhz"] = pr.clock_rate / 1e3
    except Exception as ex:  # never let a property lookup cost the bench line
        info["gpu"] = "unknown (%s)" % type(ex).__name__
    return info


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as fresh child processes of this one, which has not
    touched the GPU (nothing here imports torch or calls HIP), one rank per GPU, rendezvous on 127.0.0.1.  Rank 0's
    stdout is this process's stdout (the JSON line); the other ranks' goes to stderr.  Non-zero exit if any rank fails."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["AZ_BENCH_SELF_LAUNCHED"] = "1"
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            for p in procs:
                if p.poll() not in (None, 0):    # one rank died: the others would wait in a collective for ever
                    raise RuntimeError("rank %d exited with %d" % (procs.index(p), p.returncode))
            time.sleep(0.2)
        rc = max(abs(p.returncode) for p in procs)
    except (RuntimeError, KeyboardInterrupt) as ex:
        sys.stderr.write("bench.py launcher: %s; stopping the other ranks\n" % (ex,))
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
        rc = 1
    return rc


def run_stub(args, rank, world):
    """--workload stub: the launcher / rendezvous / aggregation skeleton of the bench on CPU tensors over gloo, for the
    CPU test of the N-rank launcher (tests/test_host.py).  No GPU, no engine; the line says so in `data`."""
    import torch
    import torch.distributed as dist
    if os.environ.get("AZ_BENCH_STUB_FAIL_RANK") == str(rank):   # the launcher test's failing rank
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if dist.get_world_size() != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))
    dist.barrier()
    t0 = time.perf_counter()
    units = 0
    for _ in range(args.steps):
        units += 1000 + rank
    dist.barrier()
    dt = time.perf_counter() - t0 + 1e-3 * (rank + 1)
    tmax = torch.tensor([dt], dtype=torch.float64)
    work = torch.tensor([float(units)], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(work, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": "stub_units_per_sec", "value": work.item() / tmax.item(), "unit": "units/s",
                          "n_gpus": dist.get_world_size(), "steps": args.steps, "warmup": args.warmup, "data": "stub",
                          "units": work.item(), "ranks_seen": dist.get_world_size()}), flush=True)
    dist.destroy_process_group()


K3_INFO = {
    "wino": dict(kernel="k_cnn_forward_wino", dtype="f32",
                 algorithm="Winograd F(2x2,3x3) for the 3x3 convolutions on v_mfma_f32_16x16x4_f32; achieved/frac in "
                           "algorithmic (direct-convolution) flops, 41.62 MFLOP per leaf"),
    "split": dict(kernel="k_cnn_forward_split",
                  dtype="f32 via 3 x f16 MFMA products on hi/lo split operands, f32 accumulation, per-position dynamic range",
                  algorithm="direct implicit GEMM on v_mfma_f32_32x32x16_f16; every fp32 product a*w as hi*hi + hi*lo + lo*hi "
                            "of the fp16 halves (22 mantissa bits), fp32 accumulation; activations rescaled per position and "
                            "layer by a power of two so that no half leaves fp16's range"),
}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", type=int, choices=[2, 3, 4], default=int(os.environ.get("AZ_BENCH_CONFIG", "2") or 2),
                    help="BASELINE.json configs[i]: 2 = Connect4 CNN self-play, 2048 slots/GPU (default, the headline); 3 = the "
                         "same with 16384 slots/GPU; 4 = NBTTT x 800 playouts, hashed evaluator, deep trees")
    ap.add_argument("--games", type=int, default=0, help="concurrent game slots per GPU (0 = the configuration's: 2048 / 16384 / 16384)")
    ap.add_argument("--playouts", type=int, default=0, help="playouts per move (0 = the configuration's: 500 / 500 / 800)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--cohort", type=int, default=512, help="games of the untimed cohort that measures plies/game")
    ap.add_argument("--mean-plies", type=float, default=0.0, help="use this plies/game instead of playing a cohort")
    ap.add_argument("--cohorts", type=int, default=0, help="slot ranges searched on their own HIP streams (0 = engine default)")
    ap.add_argument("--no-profile", action="store_true", help="experiment: no HIP-event kernel timing in the timed region")
    ap.add_argument("--profile-all", action="store_true", help="HIP events around the tree-step launches too (kernel_time_ms)")
    ap.add_argument("--k3", choices=["wino", "split"], default=DEFAULT_K3,
                    help="leaf-evaluation kernel of the headline run.  wino: Winograd on fp32 MFMA (az_cnn_wino.hip); split: "
                         "fp32 products as three fp16 MFMA products on split operands, fp32 accumulation (az_cnn_split.hip)")
    ap.add_argument("--no-second-leg", "--no-experimental", dest="no_second_leg", action="store_true",
                    help="skip the run of the other K3 kernel that is reported beside the headline (N = 1 only)")
    ap.add_argument("--mode", choices=["rolling", "lockstep"], default="rolling",
                    help="rolling: az_selfplay_rounds (a step = 500 rounds; games advance individually, leaf batches stay "
                         "full); lockstep: az_selfplay_step (a step = one ply of every slot)")
    ap.add_argument("--preroll", type=int, default=-1,
                    help="untimed de-synchronising steps before warm-up: -1 = first games staggered over 22 steps (one mean game "
                         "duration: a slot plays ~1.5 plies per step), then until twice as many games have finished as there "
                         "are slots (at most 132 steps); 0 = none (all slots start at ply 0 together); n > 0 = staggered "
                         "start, n steps in all")
    ap.add_argument("--eval-cache", type=int, default=0,
                    help="evaluation cache of the CNN engines (include/azamd.h: az_config.eval_cache_entries): 0 = the library's "
                         "default (on, auto size), -1 = off, n = that many entries")
    ap.add_argument("--launch", action="store_true",
                    help="start the ranks as child processes of this one even for --gpus 1 (a 1-rank RCCL group)")
    ap.add_argument("--workload", choices=["selfplay", "stub"], default="selfplay", help=argparse.SUPPRESS)
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if (args.gpus > 1 or args.launch) and "RANK" not in os.environ:
        # not under torchrun: be the launcher (this process never touches the GPU)
        sys.exit(launch_ranks(args.gpus, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a %d-rank run as %d GPUs"
                         % (args.gpus, world, world, args.gpus))
    if args.workload == "stub":
        return run_stub(args, rank, world)

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)

    from alphazero_amd.collector import ReplaySink, TrajectoryGatherer
    from alphazero_amd.engine import SelfPlayEngine

    cfgi = args.config
    nb = cfgi == 4                    # nine-board tic-tac-toe on the fused tree kernel; else Connect4 + CNN
    G = args.games or {2: 2048, 3: 16384, 4: 16384}[cfgi]
    S = args.playouts or (800 if nb else 500)
    FIRST_N = 12 if nb else 10        # NBTTT/config.py: first_n_steps 12; Connect4/config.py: 10
    GAME = "NBTTT" if nb else "Connect4"
    if nb:
        args.mode = "lockstep"        # synthetic evaluators run the fused search: one launch per ply of every slot
        args.no_second_leg = True
    if cfgi == 3:
        args.no_second_leg = True     # the fp32 leg belongs to the headline configuration
    dev = torch.device("cuda", local)
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    # The engine comes first: a process has few hardware queues (GPU_MAX_HW_QUEUES) and HIP hands them to streams in
    # creation order, so the engine's stream and its three cohort streams must exist before RCCL and
    # torch.distributed create theirs -- cohort streams that share a queue serialise their K3 launches (measured
    # under torch.distributed.run with the group initialised first: 374 ms per step instead of 280).
    def make_engine(k3):
        # global game ids: rank + world * k, so the RNG streams of a job's games do not depend on how it is split
        if nb:
            return SelfPlayEngine("NBTTT", n_games=G, n_playout=S, c_puct=1.5, evaluator="hash", eval_seed=7, reuse_tree=True,
                                  move="sample", rng_seed=0, device=local, game_id_base=rank, game_id_stride=world)
        e = SelfPlayEngine("Connect4", n_games=G, n_playout=S, c_puct=1.25, evaluator="cnn", flip="random",
                           reuse_tree=True, dirichlet_alpha=0.7, rng_seed=0, move="sample", device=local,
                           game_id_base=rank, game_id_stride=world, eval_cache=args.eval_cache,
                           # engine default (3 cohorts at 2 048 slots) suits the product kernel; the fp32 Winograd kernel
                           # (5 positions per workgroup, 3 x longer launches) measured best with 4
                           n_cohorts=args.cohorts or (4 if k3 == "wino" and G >= 2048 else 0))
        e.load_net(blob, kernel=k3)
        return e

    eng = make_engine(args.k3)
    # under torch.distributed.run (RANK / MASTER_ADDR exported) the RCCL group is set up for any world size, so that a
    # one-rank launch on a one-GPU box walks the same barrier / all-reduce code as the N-GPU job
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: --gpus %d but RCCL reports %d ranks" % (args.gpus, dist.get_world_size()))
    sink = ReplaySink(GAME, capacity=1 << 20, device=dev)

    def all_min(x):
        if not use_dist:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def measure(eng, k3, steps, warmup):
        """pre-roll, warm-up and the timed window on one engine -> the numbers of one leg"""
        gather = TrajectoryGatherer(eng, device=dev)

        def barrier():
            if use_dist:
                dist.barrier()
            eng.sync()
            torch.cuda.synchronize()

        def one_step():
            if args.mode == "rolling":
                eng.selfplay_rounds(S, temp=1.0, first_n_steps=FIRST_N, refill=True)
            else:
                eng.selfplay_step(temp=1.0, first_n_steps=FIRST_N, refill=True)
            eng.sync()
            recs = gather.gather()          # pack on device -> all_gather (RCCL) -> records of every rank
            sink.store_packed(recs)         # expand into ReplayBuffer-shaped tensors on device

        # ---- untimed pre-roll: slot g's first game starts at step g % T0, i.e. the starts are spread evenly over one mean
        # game DURATION (33 plies at ~1.5 plies per slot per step: terminal playouts are finished in place, so a slot gets
        # through more than one move's playouts in 500 rounds); then two more generations of games, after which what is
        # left of the start-up oscillation of the finishing rate is below a per cent (game lengths vary by ~25 %).
        # Every rank runs the same number of steps (they hold collectives).
        pre_steps = 0
        if args.preroll != 0:
            # mean game duration in steps: rolling Connect4 22 (above; 8 with the evaluation cache, where a slot gets through
            # ~4.4 playouts per round); lockstep = one ply per step: Connect4 33, NBTTT 41
            TG = (22 if args.eval_cache < 0 else 8) if args.mode == "rolling" else (41 if nb else 33)
            T0 = TG if args.preroll < 0 else max(1, min(TG, args.preroll))
            limit = 6 * T0 if args.preroll < 0 else args.preroll
            start_step = np.arange(G) % T0
            eng.reset_stats()
            while pre_steps < limit:
                if pre_steps < T0:
                    eng.set_active((start_step <= pre_steps).astype(np.uint8))
                one_step()
                pre_steps += 1
                if args.preroll < 0 and pre_steps >= T0 and all_min(eng.stats()["games_finished"]) >= 2 * G:
                    break
        for _ in range(warmup):
            one_step()
        eng.reset_stats()
        # the fused-search workload has one launch per step: time them all
        # CNN workloads, rolling: mode 4 = blocks of 64 consecutive rounds out of every 512, inside a block every cohort's K3 launch
        # and the first cohort's tree step (an event record is a packet between two dependent kernels of a cohort's chain:
        # events around every K3 launch stretch a 199 us round to 213 us, profiles/r03_bench_count6_stats.txt).  AZ_BENCH_PROF=3
        # times every K3 launch + every fourth tree step of the first cohort, =1 everything.
        pmode = 0 if args.no_profile else (1 if args.profile_all or nb else int(os.environ.get("AZ_BENCH_PROF", "4" if args.mode == "rolling" else "3")))
        eng.set_profiling(pmode)
        stored0 = sink.total_stored
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        barrier()
        dt = time.perf_counter() - t0
        prof = eng.get_profile()
        st = eng.stats()
        eng.set_profiling(False)
        # aggregate over ranks: MAX time, SUM of work
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        work = torch.tensor([st["plies"], st["playouts"], st["games_finished"], st["leaf_evals"], st["levels"],
                             st["terminal_hits"], st["children"], st["cache_hits"]], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(work, op=dist.ReduceOp.SUM)
        r = dict(zip(("plies", "playouts", "finished", "evals", "levels", "term", "children", "hits"),
                     [float(x) for x in work.tolist()]))
        # K3 launches of this rank in the window (= tree steps): counted by the events unless they only sample (mode 4)
        launches = steps * S * eng.n_cohorts if pmode == 4 else prof["cnn"]["launches"]
        r.update(dt=float(tmax.item()), prof=prof, stored=sink.total_stored - stored0, pre_steps=pre_steps, k3=k3, launches=launches,
                 pmode=pmode)
        return r

    def roofline_tree(m):
        """--config 4: the fused search kernel, HBM class.  Algorithmic bytes per playout from the run's own
        counters (16-byte node records): select levels * (16 k + 8), backup (levels + playouts) * 8,
        expand children * 16 + evals * 4, k = children per expansion; achieved = those bytes of this rank's launches /
        the launches' summed duration (HIP events on the engine stream; one launch at a time, nothing overlaps it)."""
        f = m["prof"]["fused"]
        k = m["children"] / max(m["evals"], 1)
        by = (m["levels"] * (16.0 * k + 8.0) + (m["levels"] + m["playouts"]) * 8.0 + m["children"] * 16.0 + m["evals"] * 4.0) / world
        gbs = by / (max(f["ms"], 1e-9) * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "k_search_fused<NBTTT>", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS,
                "traffic": CFG4_TRAFFIC_BYTES_PER_LAUNCH if (G, S) == (16384, 800) else None, "traffic_source": CFG4_TRAFFIC_SOURCE,
                "avg_launch_ms": f["ms"] / max(f["launches"], 1), "launches": f["launches"],
                "algorithmic_bytes_per_launch": by / max(f["launches"], 1),
                "bytes_per_playout": by * world / max(m["playouts"], 1), "mean_children": k,
                "playouts_per_launch": m["playouts"] / world / max(f["launches"], 1),
                "note": "pointer chase over per-game arenas: the launch is bound by issue slots / dependent-load latency, not "
                        "HBM (DESIGN.md 4.1); algorithmic bytes priced against the HBM peak as SURVEY 8d asks"}

    def roofline_tree_step(m):
        """The other kernel of a CNN-workload round, k_tree_step (backup of the evaluated playout, ply ends, as many walks as fit the launch's time budget
        with cache probes): HBM class by SURVEY 8d's definition, latency-bound in fact.  Algorithmic bytes from
        the run's counters: select levels x (16 k + 8), backup (levels + playouts) x 8, expand children x 16 + 4 per
        expansion, one 64-byte cache probe per non-terminal playout and one 64-byte insert per network evaluation; duration
        = HIP events around a sample of the first cohort's launches (mode 4: those of the timed blocks, 1 in 8 n_cohorts;
        mode 3: every fourth, 1 in 4 n_cohorts)."""
        b = m["prof"]["backup"]
        if not b["launches"]:
            return None
        nonterm = m["evals"] + m["hits"]
        k = m["children"] / max(nonterm, 1)
        by = (m["levels"] * (16.0 * k + 8.0) + (m["levels"] + m["playouts"]) * 8.0 + m["children"] * 16.0 + nonterm * 4.0
              + 64.0 * nonterm + 64.0 * m["evals"]) / world
        launches_all = m["launches"]                         # one tree step per K3 launch
        per_launch = by / max(launches_all, 1)
        avg_ms = b["ms"] / b["launches"]
        gbs = per_launch / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "k_tree_step<C4>", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": TREE_STEP_TRAFFIC_BYTES_PER_LAUNCH if (G, S) == (2048, 500) else None,
                "traffic_source": "rocprofv3 PMC passes on bench.py, profiles/r03_k3_pmc.txt (FETCH_SIZE x 2 + WRITE_SIZE)",
                "avg_launch_ms": avg_ms, "sampled_launches": b["launches"], "launches": launches_all,
                "algorithmic_bytes_per_launch": per_launch, "playouts_per_launch": m["playouts"] / world / max(launches_all, 1),
                "bytes_per_playout": by * world / max(m["playouts"], 1),
                "note": "a launch lasts as long as one game's chain of dependent playouts (~14 us each beside K3) within its time budget: "
                        "latency-bound, the HBM roof is not what limits it"}

    def roofline(m, k3):
        if nb:
            return roofline_tree(m)
        cnn = m["prof"]["cnn"]
        evals_rank = m["evals"] / world
        cnn_avg_ms = cnn["ms"] / max(cnn["launches"], 1)
        leaves_per_launch = evals_rank / max(m["launches"], 1)
        timed_share = cnn["launches"] / max(m["launches"], 1)   # 1 unless the events sample (mode 4: 64 rounds in 512)
        # K3 launches of different cohorts run concurrently on their own streams, so the time the kernel occupies the
        # GPU is the UNION of the launches' [start, stop] intervals (HIP events on the launch streams), not their sum.
        # achieved = FLOPs of all launches / that busy time  (= FLOPs per launch / avg duration when the launches do
        # not overlap, i.e. --cohorts 1).
        # Sampled (mode 4): the union is taken inside the timed blocks and priced with the blocks' share of the window's leaves;
        # launches of the neighbouring, untimed rounds that overlap a block's first and last intervals are not counted, which
        # can only lower `achieved`.
        alg = FLOP_PER_LEAF * evals_rank * timed_share / (max(cnn["busy_ms"], 1e-9) * 1e-3) / 1e12
        info = K3_INFO[k3]
        if k3 == "split":
            # priced on the pipe it runs on: EXECUTED fp16 MFMA flops (3 per algorithmic flop) against the dense fp16 peak;
            # the fp32-equivalent (algorithmic) rate is reported beside it
            roof = {"bound": "mfma", "kernel": info["kernel"], "achieved": 3 * alg, "peak": PEAK_F16_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": 3 * alg / PEAK_F16_MFMA_TFLOPS, "traffic": K3_SPLIT_TRAFFIC_BYTES_PER_LAUNCH,
                    "traffic_source": K3_SPLIT_TRAFFIC_SOURCE,
                    "fp32_equivalent_tflops": alg, "fp32_equivalent_frac_of_f32_mfma_peak": alg / PEAK_F32_MFMA_TFLOPS,
                    "sustained_peak_real_data": SUSTAINED_F16_MFMA_TFLOPS_REAL_DATA,
                    "frac_of_sustained_peak_real_data": 3 * alg / SUSTAINED_F16_MFMA_TFLOPS_REAL_DATA,
                    "sustained_peak_source": "tools/ubench/mfma_power.hip, profiles/r02_k3_split_breakdown.txt: all SIMDs "
                                             "issuing back to back on K3-like operand values, 1.27-1.33 GHz (power limit)",
                    "executed_flop_per_leaf": SPLIT_FLOP_PER_LEAF, "algorithm": info["algorithm"]}
        else:
            roof = {"bound": "mfma", "kernel": info["kernel"], "achieved": alg,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": alg / PEAK_F32_MFMA_TFLOPS,
                    "traffic": K3_TRAFFIC_BYTES_PER_LAUNCH, "traffic_source": "rocprofv3 PMC pass, profiles/r01_k3_pmc.txt",
                    "algorithm": info["algorithm"], "executed_flop_per_leaf": EXEC_FLOP_PER_LEAF,
                    "executed_frac": alg * (EXEC_FLOP_PER_LEAF / FLOP_PER_LEAF) / PEAK_F32_MFMA_TFLOPS}
        roof.update({"avg_launch_ms": cnn_avg_ms, "leaves_per_launch": leaves_per_launch, "launches": m["launches"],
                     "timed_launches": cnn["launches"],
                     "timing": ("HIP events around every launch of the window" if timed_share > 0.999 else
                                "HIP events around every launch of blocks of 64 consecutive rounds out of every 512 (az_set_profiling "
                                "mode 4); busy_ms / sum_launch_ms are the blocks' own"),
                     "busy_ms": cnn["busy_ms"], "sum_launch_ms": cnn["ms"], "concurrent_cohorts": eng.n_cohorts,
                     "achieved_per_launch": (3 if k3 == "split" else 1) * FLOP_PER_LEAF * leaves_per_launch
                                            / (max(cnn_avg_ms, 1e-9) * 1e-3) / 1e12})
        return roof

    m = measure(eng, args.k3, args.steps, args.warmup)

    # mean plies per game for the steady-state estimate, untimed and unbiased: one complete cohort (every game started
    # at a reset is played to its end, no refill), so short games are not over-represented as among the first finishers.
    if args.mean_plies > 0:  # profiling runs: skip the cohort so that every CNN launch in the trace is a full batch
        mean_len = args.mean_plies
    else:
        cohort = min(G, args.cohort)
        eng.traj_fetch(clear=True)
        n_done = (eng.selfplay_rolling if args.mode == "rolling" else eng.selfplay)(cohort, temp=1.0, first_n_steps=FIRST_N)
        cg, cp = eng.traj_count()
        assert cg == n_done == cohort, (cg, n_done, cohort)
        mean_len = cp / cg

    gathered = "; trajectories all-gathered over RCCL into ReplayBuffer tensors" if world > 1 else ""
    if nb:
        workload = ("NBTTT (nine-board tic-tac-toe) self-play, %d concurrent games/GPU x %d playouts/move, hashed evaluator, 2 "
                    "random opening plies per game keyed by game id, c_puct 1.5, tree reuse, %.0f GB node pool/GPU "
                    "(BASELINE configs[4]%s)" % (G, S, eng.nodes_per_game * 16.0 * G / 1e9, gathered))
    else:
        workload = ("Connect4 CNN self-play, %d concurrent games/GPU x %d playouts/move, real AZ_Connect4_CNN_current "
                    "weights, Dirichlet 0.7, random flips, tree reuse (BASELINE configs[%d]%s)"
                    % (G, S, 2 if cfgi == 2 else 3, gathered))
    if rank == 0:
        games_s = m["finished"] / m["dt"]           # games that finish inside the window / wall
        est = (m["plies"] / m["dt"]) / mean_len
        out = {
            "metric": "selfplay_games_per_sec", "value": games_s, "unit": "games/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["dt"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 tree statistics / u64 bitboards (PUCT select, expand, backup; no network)" if nb else K3_INFO[args.k3]["dtype"],
            "data": "synthetic",
            "value_steady_state_estimate": est,
            "sims_per_sec": m["playouts"] / m["dt"],
            "config": {"workload": workload,
                       "baseline_config": cfgi, "mode": args.mode, "k3": None if nb else args.k3, "games_per_gpu": G, "playouts_per_move": S,
                       "launcher": ("bench.py (self-launched ranks)" if os.environ.get("AZ_BENCH_SELF_LAUNCHED")
                                    else ("torch.distributed.run" if use_dist else "single process")),
                       "rccl_ranks": dist.get_world_size() if use_dist else None,
                       "preroll_steps_untimed": m["pre_steps"],
                       "preroll": "first games staggered over one mean game duration, then until >= 2 x games_per_gpu games finished on every rank"
                                  if args.preroll < 0 else ("none" if args.preroll == 0 else "staggered, fixed length"),
                       "mean_plies_per_game": mean_len,
                       "games_finished_in_window": m["finished"], "slot_plies_in_window": m["plies"],
                       # playouts = leaf_evals (the network ran) + eval_cache_hits (the network's stored answer) + terminal_hits
                       "leaf_evals": m["evals"], "terminal_hits": m["term"], "eval_cache_hits": m["hits"],
                       "eval_cache_entries_per_gpu": getattr(eng, "eval_cache_entries", -1),
                       "eval_cache_hit_share_of_nonterminal_leaves": m["hits"] / max(m["hits"] + m["evals"], 1),
                       "eval_cache": None if nb else "device table keyed by the evaluated position, shared by the games of a GPU; a hit "
                                     "returns the bits K3 delivers, so the games are identical with it on or off "
                                     "(tests/test_gpu_round3.py); bench.py --eval-cache -1 switches it off",
                       "mean_tree_depth": m["levels"] / max(m["playouts"], 1),
                       "plies_stored_in_replay": m["stored"],
                       # the reference's Python was timed on configs[0]'s game (Connect4, real net); no NBTTT figure exists
                       "x_reference_python_1thread": None if nb else games_s / REF_PY_GAMES_PER_S},
            "roofline": roofline(m, args.k3),
            "roofline_tree_step": None if nb else roofline_tree_step(m),
            "kernel_time_ms": {k: v["ms"] for k, v in m["prof"].items()},      # summed over the TIMED launches only
            "kernel_time_ms_timed_launches": {k: v["launches"] for k, v in m["prof"].items()},
            "box": _box_info(torch, local),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline_nbttt(S, mean_len, 7, args.cpu_budget) if nb
                                   else cpu_baseline(blob, S, mean_len, args.cpu_budget))
            out["config"]["x_cpu_port_1core"] = games_s / out["cpu_baseline"]["value"]
            if "all_cores" in out["cpu_baseline"]:
                out["config"]["x_cpu_port_all_cores"] = games_s / out["cpu_baseline"]["all_cores"]["value"]
        if world == 1 and not args.no_second_leg:
            # further legs on the same workload, same pre-roll / warm-up / window -- each in a process of its own, started
            # after this one's engine is gone: a second engine in THIS process gets its cohort streams after torch's and
            # the first engine's and has to share hardware queues (measured: 329 instead of 375 games/s for the fp32 leg)
            #   second_leg_fp32_winograd   the strictly-fp32 K3 kernel (evaluation cache as in the headline)
            #   leg_no_eval_cache          the headline kernel with the evaluation cache off: every non-terminal leaf goes
            #                              through the network, the reference's own amount of network work (MCTS.py:146)
            import subprocess
            other = "wino" if args.k3 == "split" else "split"
            eng.close()
            eng = None
            base = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
                    "--games", str(G), "--playouts", str(S), "--mode", args.mode, "--preroll", str(args.preroll),
                    "--cohorts", str(args.cohorts), "--mean-plies", repr(mean_len), "--no-cpu-baseline", "--no-second-leg"]
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                "GROUP_RANK", "ROLE_RANK", "AZ_BENCH_SELF_LAUNCHED")}
            legs = [("second_leg_" + ("fp32_winograd" if other == "wino" else "split_f16"),
                     ["--k3", other, "--eval-cache", str(args.eval_cache)], "bench.py --k3 " + other)]
            if args.eval_cache >= 0:
                legs.append(("leg_no_eval_cache", ["--k3", args.k3, "--eval-cache", "-1"], "bench.py --eval-cache -1"))
            for key, extra, switch in legs:
                try:
                    cp = subprocess.run(base + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
                    x = json.loads([l for l in cp.stdout.splitlines() if l.startswith("{")][-1])
                    out[key] = {"value": x["value"], "unit": "games/s",
                                "value_steady_state_estimate": x["value_steady_state_estimate"], "sims_per_sec": x["sims_per_sec"],
                                "steps": x["steps"], "ms_per_step": x["ms_per_step"],
                                "games_finished_in_window": x["config"]["games_finished_in_window"],
                                "leaf_evals": x["config"]["leaf_evals"], "eval_cache_hits": x["config"]["eval_cache_hits"],
                                "dtype": x["dtype"], "roofline": x["roofline"], "switch": switch,
                                "process": "child of the headline run, started after its engine was closed"}
                except Exception as ex:   # the headline line must not depend on the informational legs
                    out[key] = {"value": 0.0, "unit": "games/s", "error": "%s: %s" % (type(ex).__name__, ex),
                                "roofline": {"kernel": K3_INFO[extra[1]]["kernel"]}, "switch": switch}
        print(json.dumps(out), flush=True)
    if eng is not None:
        eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
