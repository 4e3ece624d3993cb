#!/usr/bin/env python3
"""cfg0 (BASELINE configs[0], the reference's own CPU-runnable case): ONE game, 500 playouts per move, real checkpoint, through
the engine's az_search (select -> K3 on one leaf -> backup, on the device).  Prints the time per move and per playout.
Usage on the GPU box: python tools/single_game_latency.py [n_playout]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from alphazero_amd.engine import SelfPlayEngine
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    for G in (1, 8, 64):
        eng = SelfPlayEngine("Connect4", n_games=G, n_playout=S, c_puct=1.25, evaluator="cnn", flip="random", reuse_tree=False)
        eng.load_net(blob)
        eng.search()                      # warm-up move
        eng.advance(eng.choose_moves(temp=1e-3)[0])
        eng.sync()
        t0 = time.perf_counter()
        moves = 0
        for _ in range(6):
            eng.search()
            a, _ = eng.choose_moves(temp=1e-3)
            eng.advance(a)
            moves += 1
        eng.sync()
        dt = (time.perf_counter() - t0) / moves
        print("%3d game(s) x %d playouts per move: %.1f ms per move, %.1f us per playout round, %.0f playouts/s" %
              (G, S, dt * 1e3, dt / S * 1e6, G * S / dt))
        eng.close()


if __name__ == "__main__":
    main()
