#!/usr/bin/env python3
"""Side benchmarks for the BASELINE.json configurations that are not bench.py's headline line, one JSON line each:

  k1     batched Connect4 env step (K1) on device-resident positions       -> positions/s, HBM GB/s vs 8 TB/s
  ttt    configs[1]: TicTacToe 4096 games x 100 playouts, uniform priors    -> games/s, playouts/s, K2 bytes/s
  nbttt  configs[4]: NBTTT self-play, 800 playouts/move, hashed evaluator   -> games/s, playouts/s, node-pool size
  c4big  configs[3] per-GPU shape: Connect4 CNN self-play, 16384 games/GPU  -> games/s (a few plies)

Algorithmic bytes of the tree kernels follow DESIGN.md section 4 (16-byte node records):
  select   levels * (16*k + 8)      k = mean children per expanded node, read per level walked
  backup   (levels + playouts) * 8  one 8-byte {N,Q} store per node on the path
  expand   children * 16 + evals * 4
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def tree_bytes(st):
    k = st["children"] / max(st["leaf_evals"], 1)
    sel = st["levels"] * (16.0 * k + 8.0)
    bak = (st["levels"] + st["playouts"]) * 8.0
    exp = st["children"] * 16.0 + st["leaf_evals"] * 4.0
    return sel + bak + exp, k


def bench_k1(args):
    import torch
    from alphazero_amd import _abi
    L = _abi.lib()
    _abi.require_gpu()
    n = args.n
    rng = np.random.default_rng(0)
    # mid-game positions: play 12 random plies from the empty board with the kernel itself
    st = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
    out = torch.empty_like(st)
    placed = torch.empty(n, dtype=torch.uint8, device="cuda")
    winner = torch.empty(n, dtype=torch.int8, device="cuda")
    done = torch.empty(n, dtype=torch.uint8, device="cuda")
    legal = torch.empty(n, dtype=torch.int16, device="cuda")
    vp = C.c_void_p

    def step(a):
        _abi.check(L.az_env_step_batch_dev(0, n, vp(st.data_ptr()), vp(a.data_ptr()), vp(out.data_ptr()),
                                           vp(placed.data_ptr()), vp(winner.data_ptr()), vp(done.data_ptr()),
                                           vp(legal.data_ptr())))
    for _ in range(12):
        a = torch.from_numpy(rng.integers(0, 7, size=n).astype(np.int32)).cuda()
        step(a)
        torch.cuda.synchronize()
        st.copy_(out)
    a = torch.from_numpy(rng.integers(0, 7, size=n).astype(np.int32)).cuda()
    for _ in range(3):
        step(a)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = args.reps
    ev0.record()          # az_env_step_batch_dev launches on stream 0 = torch's current (default) stream
    for _ in range(reps):
        step(a)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    bytes_per = 16 + 4 + 16 + 1 + 1 + 1 + 2   # state in, action, state out, placed, winner, done, legal = 41 B
    gbs = n * bytes_per / (ms * 1e-3) / 1e9
    print(json.dumps({"bench": "k1_env_step", "kernel": "k_env_step<C4>", "positions": n, "avg_launch_ms": ms,
                      "positions_per_s": n / (ms * 1e-3), "bytes_per_position": bytes_per,
                      "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gbs / HBM_PEAK_GBS}}), flush=True)


def bench_tree(game, G, S, c_puct, evaluator, plies, first_n, label, mean_len=0.0):
    from alphazero_amd.engine import SelfPlayEngine
    eng = SelfPlayEngine(game, n_games=G, n_playout=S, c_puct=c_puct, evaluator=evaluator, eval_seed=7,
                         flip="never" if evaluator == "uniform" else "random", reuse_tree=True, move="sample",
                         rng_seed=1)
    eng.selfplay_step(temp=1.0, first_n_steps=first_n, refill=True)   # warm-up ply
    eng.sync()
    eng.reset_stats()
    eng.set_profiling(True)
    dt = 0.0
    for i in range(plies):
        t0 = time.perf_counter()
        eng.selfplay_step(temp=1.0, first_n_steps=first_n, refill=True)
        eng.sync()
        dt += time.perf_counter() - t0
        if i % 4 == 3:
            eng.traj_fetch(clear=True)   # untimed: keep the device trajectory buffer from filling up
    prof = eng.get_profile()
    st = eng.stats()
    fused_ms = prof["fused"]["ms"]
    by, k = tree_bytes(st)
    gbs = by / (fused_ms * 1e-3) / 1e9
    # plies per finished game from a complete cohort (or given: the GPW sweeps time the search launches only)
    if mean_len <= 0:
        eng.traj_fetch(clear=True)
        cohort = min(G, 1024)
        eng.selfplay(cohort, temp=1.0, first_n_steps=first_n)
        cg, cp = eng.traj_count()
        mean_len = cp / cg
    out = {"bench": label, "kernel": "k_search_fused<%s>" % game, "games": G, "playouts_per_move": S,
           "evaluator": evaluator, "plies_timed": plies, "wall_s": dt, "playouts_per_s": st["playouts"] / dt,
           "games_per_s": (st["plies"] / dt) / mean_len, "mean_plies_per_game": mean_len,
           "mean_depth": st["levels"] / max(st["playouts"], 1), "mean_children": k,
           "bytes_per_playout": by / max(st["playouts"], 1), "nodes_in_use": st["nodes_used"],
           "node_pool_bytes": eng.nodes_per_game * 16 * G,
           "avg_launch_ms": fused_ms / max(prof["fused"]["launches"], 1), "launches": prof["fused"]["launches"],
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS, "note": "latency-bound pointer chasing; bytes are algorithmic"}}
    print(json.dumps(out), flush=True)
    eng.close()


def bench_pure(args):
    """the Elo arena's opponent, pure MCTS (UCT + random rollouts), MCTSPlayer(1, 1000) of
    Pipeline.py:130 for 4096 Connect4 games at once; beside it bench.py's cpu_baseline leg for this search (C port, 1 core)."""
    from alphazero_amd.engine import SelfPlayEngine
    G, S = args.pure_games, 1000
    eng = SelfPlayEngine("Connect4", n_games=G, n_playout=S, c_puct=1.0, evaluator="rollout", flip="random",
                         reuse_tree=False, move="sample", rng_seed=3)
    eng.search()
    eng.sync()
    eng.reset_stats()
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(4):
        eng.reset_games()
        eng.search()
    eng.sync()
    dt = time.perf_counter() - t0
    st, prof = eng.stats(), eng.get_profile()
    from bench import cpu_baseline_pure   # the CPU port is timed by bench.py's cpu_baseline leg, nowhere else
    print(json.dumps({"bench": "pure_mcts_c4_%dx1000" % G, "kernel": "k_search_pure<C4>", "games": G,
                      "playouts_per_move": S, "wall_s": dt, "playouts_per_s": st["playouts"] / dt,
                      "rollout_plies_per_s": st["rollout_plies"] / dt,
                      "mean_rollout_plies": st["rollout_plies"] / max(st["leaf_evals"], 1),
                      "mean_depth": st["levels"] / max(st["playouts"], 1),
                      "avg_launch_ms": prof["fused"]["ms"] / max(prof["fused"]["launches"], 1),
                      "cpu_baseline": None if args.no_cpu else cpu_baseline_pure(S)}), flush=True)
    eng.close()


def bench_c4big(args):
    from alphazero_amd.engine import SelfPlayEngine
    G, S = args.games, 500
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    eng = SelfPlayEngine("Connect4", n_games=G, n_playout=S, c_puct=1.25, evaluator="cnn", flip="random",
                         reuse_tree=True, dirichlet_alpha=0.7, rng_seed=0, move="sample")
    eng.load_net(blob)
    eng.selfplay_step(temp=1.0, first_n_steps=10, refill=True)
    eng.sync()
    eng.reset_stats()
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.plies):
        eng.selfplay_step(temp=1.0, first_n_steps=10, refill=True)
        eng.sync()
    dt = time.perf_counter() - t0
    prof, st = eng.get_profile(), eng.stats()
    cnn = prof["cnn"]
    avg = cnn["ms"] / max(cnn["launches"], 1)
    leaves = st["leaf_evals"] / max(cnn["launches"], 1)
    tf = 41.62e6 * leaves / (avg * 1e-3) / 1e12
    print(json.dumps({"bench": "c4_cnn_selfplay_16384", "games": G, "playouts_per_move": S, "plies_timed": args.plies,
                      "wall_s": dt, "playouts_per_s": st["playouts"] / dt,
                      "games_per_s_at_33.27_plies": (st["plies"] / dt) / 33.27, "nodes_in_use": st["nodes_used"],
                      "kernel_ms": {k: v["ms"] for k, v in prof.items()},
                      "roofline": {"bound": "mfma", "kernel": "k_cnn_forward", "achieved": tf, "peak": 157.3,
                                   "unit": "TFLOP/s", "frac": tf / 157.3, "avg_launch_ms": avg,
                                   "leaves_per_launch": leaves}}), flush=True)
    eng.close()


def bench_unpack(args):
    """packed ply records -> ReplayBuffer rows (k_traj_unpack<C4>), the bulk form of buffer.store.
    Algorithmic bytes per ply: 48 in + 2 x 504 (state, next_state) + 28 (pi) + 4 + 4 + 1 = 1 093 B, all but 48 written."""
    import torch
    from alphazero_amd import _abi
    L = _abi.lib()
    _abi.require_gpu()
    n = args.unpack_n
    rng = np.random.default_rng(1)
    vp = C.c_void_p
    st = torch.zeros((n, 2), dtype=torch.int64, device="cuda")      # positions 12 random plies into a game (K1)
    out = torch.empty_like(st)
    for _ in range(12):
        a = torch.from_numpy(rng.integers(0, 7, size=n).astype(np.int32)).cuda()
        _abi.check(L.az_env_step_batch_dev(0, n, vp(st.data_ptr()), vp(a.data_ptr()), vp(out.data_ptr()), None, None,
                                           None, None))
        st.copy_(out)
    recs = torch.empty((n, 48), dtype=torch.uint8, device="cuda")
    recs[:, :16] = st.view(torch.uint8).reshape(n, 16)
    recs[:, 16:44] = torch.rand((n, 7), device="cuda").view(torch.uint8).reshape(n, 28)
    recs[:, 44:48] = torch.from_numpy(np.stack([rng.integers(-1, 2, n), rng.choice([-1, 1], n), rng.integers(0, 7, n),
                                                rng.integers(0, 2, n)], 1).astype(np.int8)).cuda().view(torch.uint8)
    state = torch.empty((n, 3, 6, 7), dtype=torch.float32, device="cuda")
    nxt = torch.empty_like(state)
    prob = torch.empty((n, 7), dtype=torch.float32, device="cuda")
    value = torch.empty((n, 1), dtype=torch.float32, device="cuda")
    winner = torch.empty((n, 1), dtype=torch.int32, device="cuda")
    done = torch.empty((n, 1), dtype=torch.uint8, device="cuda")

    def run():
        _abi.check(L.az_traj_unpack_dev(0, n, vp(recs.data_ptr()), vp(state.data_ptr()), vp(prob.data_ptr()),
                                        vp(value.data_ptr()), vp(winner.data_ptr()), vp(nxt.data_ptr()),
                                        vp(done.data_ptr())))
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()          # az_traj_unpack_dev launches on stream 0 = torch's current (default) stream
    for _ in range(args.reps):
        run()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / args.reps
    bytes_per = 48 + 2 * 504 + 28 + 4 + 4 + 1
    gbs = n * bytes_per / (ms * 1e-3) / 1e9
    # spot check against the planes kernel (env.current_state) so that a fast wrong kernel cannot pass for a result
    from alphazero_amd.engine import env_planes_batch
    ref = env_planes_batch("Connect4", st[:1024].cpu().numpy().view(np.uint64))
    assert np.array_equal(ref, state[:1024].cpu().numpy())
    # what a plain device fill of the two plane tensors reaches on this box (write-only stream of the same size)
    ev0.record()
    for _ in range(args.reps):
        state.fill_(1.0)
        nxt.fill_(1.0)
    ev1.record()
    torch.cuda.synchronize()
    fill_gbs = 2 * n * 504 / (ev0.elapsed_time(ev1) / args.reps * 1e-3) / 1e9
    print(json.dumps({"bench": "traj_unpack_c4", "kernel": "k_traj_unpack_c4_lines", "plies": n, "avg_launch_ms": ms,
                      "plies_per_s": n / (ms * 1e-3), "bytes_per_ply": bytes_per, "torch_fill_same_tensors_gbs": fill_gbs,
                      "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                                   "frac": gbs / 8000.0}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="+", choices=["k1", "ttt", "nbttt", "c4big", "pure", "unpack"])
    ap.add_argument("--unpack-n", type=int, default=1 << 21)
    ap.add_argument("--n", type=int, default=1 << 24)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--games", type=int, default=16384)
    ap.add_argument("--plies", type=int, default=3)
    ap.add_argument("--nbttt-games", type=int, default=8192)
    ap.add_argument("--ttt-games", type=int, default=4096)
    ap.add_argument("--pure-games", type=int, default=4096)
    ap.add_argument("--mean-len", type=float, default=0.0, help="skip the plies-per-game cohort of ttt / nbttt and use this")
    ap.add_argument("--no-cpu", action="store_true", help="pure: skip the CPU port's leg")
    args = ap.parse_args()
    for w in args.which:
        if w == "k1":
            bench_k1(args)
        elif w == "ttt":
            bench_tree("TTT", args.ttt_games, 100, 1.0, "uniform", 40, 1, "ttt_%dx100_uniform" % args.ttt_games, args.mean_len)
        elif w == "nbttt":
            bench_tree("NBTTT", args.nbttt_games, 800, 1.5, "hash", 6, 12, "nbttt_%dx800_hash" % args.nbttt_games, args.mean_len)
        elif w == "c4big":
            bench_c4big(args)
        elif w == "pure":
            bench_pure(args)
        elif w == "unpack":
            bench_unpack(args)


if __name__ == "__main__":
    main()
