#!/usr/bin/env python3
"""Soak test of K3 inside the search (inline-asm MFMAs pad their own hazards: a sporadic wrong value would show here):
G identical games, eval mode, never-flip, first-max -> after every ply all G trees must be bit-identical, wherever their
leaves sat in the CNN batches.  Usage on the GPU box: python tools/soak_k3.py [G] [plies]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from alphazero_amd.engine import SelfPlayEngine
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    plies = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    eng = SelfPlayEngine("Connect4", n_games=G, n_playout=500, c_puct=1.25, evaluator="cnn", flip="never", reuse_tree=True)
    eng.load_net(blob)
    for ply in range(plies):
        eng.search()
        rc = eng.root_children()
        for k in ("actions", "visits"):
            assert (rc[k] == rc[k][0]).all(), (ply, k)
        assert (rc["q"].view(np.uint32) == rc["q"].view(np.uint32)[0]).all(), ply
        assert (rc["prior"].view(np.uint32) == rc["prior"].view(np.uint32)[0]).all(), ply
        a, _ = eng.choose_moves(temp=0.0)
        assert (a == a[0]).all()
        print("ply %d: %d identical trees, move %d, visits %s" % (ply, G, a[0], rc["visits"][0][:int(rc["n"][0])]), flush=True)
        if (a < 0).any():
            break
        eng.advance(a)
        _, w, dn = eng.get_roots() if hasattr(eng, "get_roots") else (None, None, None)
        if dn is not None and dn.all():
            break
    st = eng.stats()
    print("leaf evaluations:", st["leaf_evals"], "playouts:", st["playouts"])
    eng.close()


if __name__ == "__main__":
    main()
