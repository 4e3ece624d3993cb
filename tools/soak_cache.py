#!/usr/bin/env python3
"""Soak test of the lock-free evaluation cache: the same job (rolling self-play, real network, Dirichlet noise, random flips)
with the cache off, with a table so small that thousands of concurrent games overwrite each other's entries all the time
(chunks of different writers in one entry, torn entries: the tags must refuse every one of them), and at the default size.
All three must play the same games bit for bit.  Usage on the GPU box: python3 tools/soak_cache.py [slots] [games] [tiny_entries]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from alphazero_amd.engine import SelfPlayEngine
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
    tiny = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    ref = None
    for name, cache in (("off", False), ("tiny", tiny), ("auto", 0)):
        eng = SelfPlayEngine("Connect4", n_games=G, n_playout=200, c_puct=1.25, evaluator="cnn", flip="random", reuse_tree=True,
                             dirichlet_alpha=0.7, rng_seed=77, move="sample", eval_cache=cache)
        eng.load_net(blob)
        t0 = time.time()
        n = eng.selfplay_rolling(N, temp=1.0, first_n_steps=8)
        dt = time.time() - t0
        t, s = eng.traj_fetch(), eng.stats()
        o = np.lexsort((np.arange(len(t["game_id"])), t["game_id"]))
        h = hashlib.sha256()
        for k in ("game_id", "states", "action", "player", "done", "z", "pi"):
            h.update(np.ascontiguousarray(t[k][o]).tobytes())
        sig = (h.hexdigest(), len(o), s["playouts"], s["terminal_hits"], s["levels"], s["children"])
        print("%-5s entries %11d: %d games, %d plies in %.1f s; leaf_evals %d cache_hits %d  sha %s" % (
            name, eng.eval_cache_entries, n, len(o), dt, s["leaf_evals"], s["cache_hits"], sig[0][:16]), flush=True)
        if ref is None:
            ref = sig
        assert sig == ref, (name, sig, ref)
        eng.close()
    print("identical")


if __name__ == "__main__":
    main()
