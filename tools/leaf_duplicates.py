#!/usr/bin/env python3
"""How many leaf evaluations of a self-play game repeat an earlier one?  (measure before
building an evaluation cache.)

The network's output is a pure function of the evaluated (possibly mirrored) position, so a per-game table keyed by the
16-byte position could answer a repeated leaf with exactly the bits K3 would produce.  This tool plays the bench's workload
(Connect4, real checkpoint, 500 playouts per move, Dirichlet 0.7, random flips, temperature schedule, tree reuse) for a sample
of games through the host-evaluator path (az_select -> az_net_forward on a second engine -> az_backup), i.e. the same
search, and counts per game, over all evaluated (non-terminal) leaves:

  same_move     the exact evaluated position (orientation included) was already evaluated during THIS move's search
  same_game     ... at any earlier point of this game (a cache that lives as long as the game)
  canon_game    ... in either orientation (what a cache could reach if the network were mirror-symmetric; it is not)
  any_game      the exact evaluated position was already evaluated by ANY game of the sample (a table shared by all the games
                of a GPU; a lower bound: the sample has `games` concurrent games and no history of earlier generations)

Usage (GPU box): python3 tools/leaf_duplicates.py [games] [playouts]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mirror_key(w0, w1):
    """column mirror of a packed Connect4 position (include/azamd.h: stone bit = col*7 + h)"""
    def m(x):
        out = 0
        for c in range(7):
            out |= ((x >> (7 * c)) & 0x7F) << (7 * (6 - c))
        return out
    return m(w0 & ((1 << 49) - 1)), m(w1 & ((1 << 49) - 1)) | (w1 & (1 << 63))


def main():
    from alphazero_amd.engine import SelfPlayEngine
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    host = SelfPlayEngine("Connect4", n_games=G, n_playout=S, c_puct=1.25, evaluator="host", flip="random", reuse_tree=True,
                          dirichlet_alpha=0.7, rng_seed=0, move="sample")
    net = SelfPlayEngine("Connect4", n_games=1, n_playout=1, evaluator="cnn")
    net.load_net(blob)
    seen_game = [set() for _ in range(G)]
    canon_game = [set() for _ in range(G)]
    seen_all = set()
    tot = dict(evals=0, same_move=0, same_game=0, canon_game=0, any_game=0, terminal=0)
    per_ply = {}
    t0 = time.time()
    ply = 0
    probs = np.zeros((G, 7), np.float32)
    values = np.zeros(G, np.float32)
    while True:
        _, _, done = host.get_roots()
        live = done == 0
        if not live.any():
            break
        seen_move = [set() for _ in range(G)]
        pp = per_ply.setdefault(ply, dict(evals=0, same_move=0, same_game=0, any_game=0))
        for _ in range(S):
            st, fl = host.select()
            need = np.nonzero(((fl & 1) == 0) & ((fl & 4) == 0))[0]     # bit0 terminal, bit2 idle slot
            tot["terminal"] += int(((fl & 1) == 1).sum())
            if len(need):
                _, _, p, v = net.net_forward(st[need])
                probs[need] = p
                values[need] = v[:, 0]
                for g in need:
                    k = (int(st[g, 0]), int(st[g, 1]))
                    mk = mirror_key(*k)
                    ck = min(k, mk)
                    tot["evals"] += 1
                    pp["evals"] += 1
                    if k in seen_move[g]:
                        tot["same_move"] += 1
                        pp["same_move"] += 1
                    if k in seen_game[g]:
                        tot["same_game"] += 1
                        pp["same_game"] += 1
                    if ck in canon_game[g]:
                        tot["canon_game"] += 1
                    if k in seen_all:
                        tot["any_game"] += 1
                        pp["any_game"] += 1
                    seen_all.add(k)
                    seen_move[g].add(k)
                    seen_game[g].add(k)
                    canon_game[g].add(ck)
            host.backup(probs, values)
        a, _ = host.choose_moves(temp=1.0 if ply < 10 else 1e-3)
        host.advance(a)
        ply += 1
        if ply % 5 == 0:
            print("ply %d  %.0f s  evals %d  same_move %.3f  same_game %.3f  canon_game %.3f  any_game %.3f" % (
                ply, time.time() - t0, tot["evals"], tot["same_move"] / tot["evals"], tot["same_game"] / tot["evals"],
                tot["canon_game"] / tot["evals"], tot["any_game"] / tot["evals"]), file=sys.stderr, flush=True)
    e = max(tot["evals"], 1)
    out = dict(games=G, playouts_per_move=S, plies=ply, evaluated_leaves=tot["evals"], terminal_playouts=tot["terminal"],
               repeat_share_same_move=tot["same_move"] / e, repeat_share_same_game=tot["same_game"] / e,
               repeat_share_same_game_either_orientation=tot["canon_game"] / e,
               repeat_share_any_game_of_the_sample=tot["any_game"] / e, distinct_positions_all_games=len(seen_all),
               mean_distinct_positions_per_game=float(np.mean([len(s) for s in seen_game])),
               by_ply={str(k): dict(evals=v["evals"], same_move=v["same_move"] / max(v["evals"], 1),
                                    same_game=v["same_game"] / max(v["evals"], 1),
                                    any_game=v["any_game"] / max(v["evals"], 1)) for k, v in sorted(per_ply.items())},
               seconds=time.time() - t0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
