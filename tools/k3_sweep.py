#!/usr/bin/env python3
"""K3 micro-benchmark: time az_net_forward_dev (the fused CNN kernel alone) over a sweep of batch sizes.
Shows how the launch time steps with the number of workgroups (occupancy rounds on 256 CUs) -- the input for choosing
samples per workgroup.  Usage on the GPU box: python tools/k3_sweep.py [n ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from alphazero_amd.engine import SelfPlayEngine
    ns = [int(x) for x in sys.argv[1:]] or [96, 384, 768, 1152, 1536, 1920, 2033, 2048, 2304, 3072, 4096, 6144, 8192,
                                            16384, 65536]
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    eng = SelfPlayEngine("Connect4", n_games=1, n_playout=1, evaluator="cnn")
    eng.load_net(blob)
    rng = np.random.default_rng(0)
    nmax = max(ns)
    # random (not necessarily reachable) positions: disjoint X/O stones below the column sentinels
    cells = rng.integers(0, 3, size=(nmax, 42))
    x = np.zeros(nmax, dtype=np.uint64)
    o = np.zeros(nmax, dtype=np.uint64)
    for i in range(42):
        bit = np.uint64(1) << np.uint64((i % 7) * 7 + i // 7)
        x |= np.where(cells[:, i] == 1, bit, np.uint64(0))
        o |= np.where(cells[:, i] == 2, bit, np.uint64(0))
    st = torch.from_numpy(np.stack([x, o], 1).view(np.int64)).cuda()
    probs = torch.empty((nmax, 7), dtype=torch.float32, device="cuda")
    value = torch.empty((nmax,), dtype=torch.float32, device="cuda")
    vp = C.c_void_p
    L = eng.L
    for n in ns:
        reps = 30 if n <= 16384 else 8
        for _ in range(3):
            L.az_net_forward_dev(eng.h, n, vp(st.data_ptr()), vp(probs.data_ptr()), vp(value.data_ptr()))
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            L.az_net_forward_dev(eng.h, n, vp(st.data_ptr()), vp(probs.data_ptr()), vp(value.data_ptr()))
        eng.sync()
        dt = (time.perf_counter() - t0) / reps
        tf = n * 41.62e6 / dt / 1e12
        print("n=%6d  %9.1f us  %7.2f TFLOP/s  frac=%.3f  leaves/s=%.3e" %
              (n, dt * 1e6, tf, tf / 157.3, n / dt), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
