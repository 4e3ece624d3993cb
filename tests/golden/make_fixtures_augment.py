#!/usr/bin/env python3
"""Golden vectors for the sample-time mirror augmentation, by RUNNING THE REFERENCE'S OWN functions:
environments/Connect4/utils.py:114-128 and environments/TTT/utils.py:74-89 (`instant_augment`).

Container-only (needs /root/reference).  Writes data only: tests/golden/augment.npz = seeded input batches of the
reference's field layouts and the tensors the reference returns for them."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

_refload.load_reference()
import importlib  # noqa: E402

c4u = importlib.import_module("environments.Connect4.utils")
tttu = importlib.import_module("environments.TTT.utils")


def main():
    rng = np.random.default_rng(20261111)
    out = {}
    for B in (1, 5, 64):
        c4 = (rng.random((B, 3, 6, 7), dtype=np.float32), rng.random((B, 7), dtype=np.float32),
              rng.random((B, 1), dtype=np.float32), rng.integers(-1, 2, (B, 1)).astype(np.int32),
              rng.random((B, 3, 6, 7), dtype=np.float32), rng.integers(0, 2, (B, 1)).astype(bool))
        res = c4u.instant_augment(tuple(torch.from_numpy(x) for x in c4))
        assert len(res) == 6
        for i, (x, y) in enumerate(zip(c4, res)):
            out["c4/%d/in%d" % (B, i)] = x
            out["c4/%d/out%d" % (B, i)] = y.numpy()
        ttt = (rng.random((B, 3, 3, 3), dtype=np.float32), rng.random((B, 9), dtype=np.float32),
               rng.random((B, 1), dtype=np.float32), rng.random((B, 3, 3, 3), dtype=np.float32),
               rng.integers(0, 2, (B, 1)).astype(bool), rng.integers(0, 2, (B, 9)).astype(bool))
        res = tttu.instant_augment(tuple(torch.from_numpy(x) for x in ttt))
        assert len(res) == 6
        for i, (x, y) in enumerate(zip(ttt, res)):
            out["ttt/%d/in%d" % (B, i)] = x
            out["ttt/%d/out%d" % (B, i)] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)
    print("wrote augment.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
