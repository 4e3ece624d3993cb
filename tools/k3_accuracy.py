#!/usr/bin/env python3
"""How far is each K3 kernel from the EXACT network?  The 256 positions of tests/golden/cnn.npz through the Winograd
fp32 kernel and through the split-fp16 kernel (AZ_K3_SPLIT=1), each compared with a float64 forward of the same
weights (torch, CPU; structure checked against the outputs recorded from the reference).  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import split_fp16_accuracy as ref  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def main():
    from alphazero_amd.engine import SelfPlayEngine
    blob = np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    d = np.load(os.path.join(G, "cnn.npz"))
    cells = d["cells"].reshape(-1, 6, 7)
    turn = d["turn"]
    x = torch.stack([torch.from_numpy((cells == 1).astype(np.float32)), torch.from_numpy((cells == -1).astype(np.float32)),
                     torch.from_numpy(turn.astype(np.float32))[:, None, None].expand(-1, 6, 7)], 1)
    lp64, lv64 = ref.forward(ref.load(blob), x, "exact", torch.float64)
    lp64, lv64 = lp64.numpy(), lv64.numpy()
    # packed positions: bit = col * 7 + (5 - row); word 1 carries O's stones and the turn flag (bit 63)
    packed = np.zeros((len(cells), 2), dtype=np.uint64)
    for i, b in enumerate(cells):
        xs = os_ = 0
        for r in range(6):
            for c in range(7):
                if b[r, c] == 1:
                    xs |= 1 << (c * 7 + 5 - r)
                elif b[r, c] == -1:
                    os_ |= 1 << (c * 7 + 5 - r)
        if turn[i] == -1:
            os_ |= 1 << 63
        packed[i] = (xs, os_)
    out = {}
    for name, env in (("winograd fp32", None), ("split fp16", "1")):
        if env:
            os.environ["AZ_K3_SPLIT"] = env
        e = SelfPlayEngine("Connect4", n_games=8, n_playout=10, evaluator="cnn")
        e.load_net(blob)
        os.environ.pop("AZ_K3_SPLIT", None)
        lp, lv, _, _ = e.net_forward(packed)
        e.close()
        out[name] = (lp, lv)
        print("%-14s vs float64: logp max abs %.2e  mean abs %.2e | logv max abs %.2e  mean abs %.2e | vs recorded "
              "reference fp32: logp max abs %.2e" % (name, np.abs(lp - lp64).max(), np.abs(lp - lp64).mean(),
                                                     np.abs(lv - lv64).max(), np.abs(lv - lv64).mean(),
                                                     np.abs(lp - d["logp"]).max()))
    print("recorded reference fp32 (torch CPU) vs float64: logp max abs %.2e  mean abs %.2e" %
          (np.abs(d["logp"] - lp64).max(), np.abs(d["logp"] - lp64).mean()))


if __name__ == "__main__":
    main()
