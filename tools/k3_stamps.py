#!/usr/bin/env python3
"""Reads the in-kernel shader-clock stamps of a -DAZ_K3_DBG_TS build of the Winograd K3 kernel (wave 0 of workgroup 0).
Usage on the GPU box: AZAMD_LIB=dbgbuild/libazamd_AZ_K3_DBG_TS.so python tools/k3_stamps.py [n]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from alphazero_amd.engine import SelfPlayEngine
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")
    eng = SelfPlayEngine("Connect4", n_games=1, n_playout=1, evaluator="cnn")
    eng.load_net(blob)
    st = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
    probs = torch.empty((n, 7), dtype=torch.float32, device="cuda")
    value = torch.empty((n,), dtype=torch.float32, device="cuda")
    vp = C.c_void_p
    for _ in range(3):
        eng.L.az_net_forward_dev(eng.h, n, vp(st.data_ptr()), vp(probs.data_ptr()), vp(value.data_ptr()))
    eng.sync()
    ts = np.zeros(512, dtype=np.uint64)
    rc = eng.L.az_dbg_read_stamps(ts.ctypes.data_as(C.c_void_p), 512)
    assert rc == 0, rc
    t = ts.astype(np.int64)
    print("stem..blocks start:", 0)
    for blk in range(3):
        npass = int(os.environ.get("K3_NPASS", "4"))
        for mt in range(npass):
            a, b = t[1 + (blk * 8 + mt) * 4], t[1 + (blk * 8 + mt) * 4 + 1]
            nxt = t[1 + (blk * 8 + mt + 1) * 4] if mt < npass - 1 else t[200 + blk * 3]
            nm = 4096 // npass
            if blk == 0 or mt == 0:
                print("blk %d pass %d: loop %7d cyc (%.1f/MFMA)  tail(1x1+xform+prologue) %7d cyc (%.1f/MFMA)" %
                      (blk, mt, b - a, (b - a) / nm, nxt - b, (nxt - b) / (nm / 4.0)))
        print("blk %d: barrier wait %d, epilogue %d" % (blk, t[201 + blk * 3] - t[200 + blk * 3],
                                                      t[202 + blk * 3] - t[201 + blk * 3]))
    print("blocks total %d cyc; first pass start - WTS0 = %d" % (t[220] - t[0], t[1] - t[0]))
    print("kernel %d cyc: planes+stem+geometry %d, blocks %d, heads+linear+softmax %d" %
          (t[231] - t[230], t[0] - t[230], t[220] - t[0], t[231] - t[220]))
    eng.close()


if __name__ == "__main__":
    main()
