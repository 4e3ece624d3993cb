#!/usr/bin/env python3
"""BASELINE configs[0] through the engine: ONE Connect4 game, 500 playouts per move, real checkpoint (what the drop-in MCTS_AZ /
AlphaZeroPlayer do under play.py --self_play): milliseconds per move of az_search + move choice + advance, for the first game of
an engine and for the third (the evaluation cache then holds the earlier games' positions).  AZ_SEARCH_ROLL=0 selects the
lockstep chain (one { K3 ; tree step } round per playout).  Usage (GPU box): python3 tools/single_game_rate.py [games] [n_slots]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from alphazero_amd.engine import SelfPlayEngine
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    eng = SelfPlayEngine("Connect4", n_games=G, n_playout=500, c_puct=1.25, evaluator="cnn", flip="random", reuse_tree=True,
                         dirichlet_alpha=0.7, rng_seed=1, move="sample")
    eng.load_net(blob)
    for gi in range(games):
        eng.reset_games()
        eng.reset_stats()
        plies, t0 = 0, time.perf_counter()
        while True:
            eng.search()
            a, _ = eng.choose_moves(temp=1.0 if plies < 10 else 1e-3)
            if (a < 0).all():
                break
            eng.advance(a)
            plies += 1
            _, _, dn = eng.get_roots()
            if dn.all():
                break
        dt = time.perf_counter() - t0
        st = eng.stats()
        print("game %d (%d slots): %d plies, %.2f ms per move, %.1f us per playout; leaf_evals %d cache_hits %d terminal %d" % (
            gi, G, plies, 1e3 * dt / max(plies, 1), 1e6 * dt * G / max(st["playouts"], 1), st["leaf_evals"], st["cache_hits"],
            st["terminal_hits"]), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
