#!/usr/bin/env python3
"""In-kernel shader-clock stamps of the split-fp16 K3 (diagnostic build -DAZ_K3S_DBG_TS, tools/k3_variants.sh): where one
workgroup's cycles go, phase by phase, while every CU runs the kernel.
Usage on the GPU box:  AZAMD_LIB=dbgbuild/libazamd_AZ_K3S_DBG_TS.so python tools/k3s_stamps.py [n_leaves]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {1: "stem conv", 2: "stem bias+SiLU", 3: "stem to image"}
for b in range(3):
    NAMES.update({4 + 8 * b: "blk%d x read-back" % b, 5 + 8 * b: "blk%d conv3x3" % b, 6 + 8 * b: "blk%d bias+SiLU" % b,
                  7 + 8 * b: "blk%d t to image" % b, 8 + 8 * b: "blk%d conv1x1" % b, 9 + 8 * b: "blk%d bias+res+SiLU" % b,
                  10 + 8 * b: "blk%d x to image" % b})
NAMES.update({28: "(to heads)", 29: "heads"})


def main():
    import torch
    from alphazero_amd.engine import SelfPlayEngine
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    eng = SelfPlayEngine("Connect4", n_games=1, n_playout=1, evaluator="cnn")
    eng.load_net(blob, kernel="split")
    rng = np.random.default_rng(0)
    st = torch.from_numpy(rng.integers(0, 1 << 40, size=(n, 2), dtype=np.int64) & 0x0FDFBF7EFDFB).cuda()
    st[:, 1] = st[:, 1] & ~st[:, 0]
    probs = torch.empty((n, 7), dtype=torch.float32, device="cuda")
    value = torch.empty((n,), dtype=torch.float32, device="cuda")
    vp = C.c_void_p
    for _ in range(3):
        eng.L.az_net_forward_dev(eng.h, n, vp(st.data_ptr()), vp(probs.data_ptr()), vp(value.data_ptr()))
    eng.sync()
    ts = np.zeros((4, 64), dtype=np.uint64)
    f = eng.L.az_dbg_split_stamps
    f.argtypes, f.restype = [C.c_void_p], C.c_int
    assert f(ts.ctypes.data_as(C.c_void_p)) == 0
    t = ts.astype(np.int64)
    idx = sorted(NAMES)
    print("cycles per phase, waves 0..3 of workgroup 7 (n = %d leaves)" % n)
    prev = 0
    tot = {}
    for i in idx:
        d = t[:, i] - t[:, prev]
        print("%-22s %s" % (NAMES[i], "  ".join("%8d" % x for x in d)))
        key = NAMES[i].split(" ", 1)[1] if NAMES[i].startswith("blk") else NAMES[i]
        tot[key] = tot.get(key, 0) + int(d[0])
        prev = i
    print("total %d cycles (wave 0)" % (t[0, 29] - t[0, 0]))
    print("entry -> first stamp (LDS image cleared, input planes): %s" % "  ".join("%8d" % x for x in t[:, 0] - t[:, 30]))
    print("heads -> end (Linear, log-softmax, outputs; waves 0..2): %s" % "  ".join("%8d" % x for x in t[:, 31] - t[:, 29]))
    if t[0, 32] > 0:   # inside the epilogue of blk1's t (layer 3): SiLU + maximum | barrier | scale, split, store | barrier
        for a, b, name in ((14, 32, "conv3x3 end -> maxima published"), (32, 33, "barrier 1"), (33, 34, "scale, split, store"),
                           (34, 15, "barrier 2")):
            print("blk1 t: %-32s %s" % (name, "  ".join("%8d" % x for x in t[:, b] - t[:, a])))
    print("entry -> end: %s" % "  ".join("%8d" % x for x in t[:, 31] - t[:, 30]))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print("  %-20s %8d  %5.1f %%" % (k, v, 100.0 * v / (t[0, 29] - t[0, 0])))
    eng.close()


if __name__ == "__main__":
    main()
