#!/usr/bin/env python3
"""The rolling tree step without K3 beside it: 2 048 IDENTICAL Connect4 games (eval mode, no flips, first-max moves), so that
after the first pass every non-terminal leaf is answered by the evaluation cache and the K3 launches find empty leaf lists.
Prints the tree step's average launch time (HIP events) and the playouts per second it sustains alone, per in-place cap.
Usage (GPU box): AZ_ROLL_CAP=6 python3 tools/tree_step_alone.py [slots] [cohorts]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd.engine import SelfPlayEngine

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
C = int(sys.argv[2]) if len(sys.argv) > 2 else 0
blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
eng = SelfPlayEngine("Connect4", n_games=G, n_playout=500, c_puct=1.25, evaluator="cnn", flip="never", reuse_tree=True,
                     move="first_max", n_cohorts=C)
eng.load_net(blob)
for _ in range(30):      # one whole game and more: the table now holds every position these games ever evaluate
    eng.selfplay_rounds(500, temp=1.0, first_n_steps=10, refill=True)
    eng.sync()
    eng.traj_fetch(clear=True)
eng.reset_stats()
eng.set_profiling(1)
t0 = time.perf_counter()
for _ in range(10):
    eng.selfplay_rounds(500, temp=1.0, first_n_steps=10, refill=True)
    eng.sync()
    eng.traj_fetch(clear=True)
dt = time.perf_counter() - t0
st, pr = eng.stats(), eng.get_profile()
print("slots %d cohorts %d cap %s: %.1f M playouts/s, %.2f playouts per slot and round; leaf_evals %d cache_hits %d terminal %d" % (
    G, eng.n_cohorts, os.environ.get("AZ_ROLL_CAP", "3"), st["playouts"] / dt / 1e6, st["playouts"] / (5000.0 * G),
    st["leaf_evals"], st["cache_hits"], st["terminal_hits"]))
for k, v in pr.items():
    if v["launches"]:
        print("  %-7s %6d launches  avg %.1f us  busy %.1f ms of %.1f ms" % (k, v["launches"], 1e3 * v["ms"] / v["launches"], v["busy_ms"], dt * 1e3))
eng.close()
