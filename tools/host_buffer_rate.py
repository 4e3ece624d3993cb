import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from alphazero_amd.engine import SelfPlayEngine
blob = np.fromfile("tests/golden/weights/AZ_Connect4_CNN_current.f32", dtype="<f4")
e = SelfPlayEngine("Connect4", n_games=8, n_playout=8, evaluator="cnn")
e.load_net(blob)
rng = np.random.default_rng(0)
for n in (1, 256, 4096, 65536):
    st = (rng.integers(0, 1 << 40, size=(n, 2), dtype=np.int64) & 0x0FDFBF7EFDFB).astype(np.uint64)
    st[:, 1] &= ~st[:, 0]
    for _ in range(3):
        e.net_forward(st)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        e.net_forward(st)
    dt = (time.perf_counter() - t0) / reps
    line = "az_net_forward (host buffers in and out, allocates) n=%6d: %9.1f us  %.3e leaves/s" % (n, dt * 1e6, n / dt)
    if n <= 4096:
        for _ in range(3):
            e.net_forward_static(st)
        t0 = time.perf_counter()
        for _ in range(reps):
            e.net_forward_static(st)
        ds = (time.perf_counter() - t0) / reps
        line += " | az_net_forward_static: %9.1f us  %.3e leaves/s" % (ds * 1e6, n / ds)
    print(line)
from alphazero_amd.engine import env_status_batch, env_step_batch
one = np.zeros((1, 2), np.uint64)
calls = (("az_env_status_batch n=1 (env.valid_move / done)", lambda: env_status_batch("Connect4", one)),
         ("az_env_step_batch   n=1 (env.step)", lambda: env_step_batch("Connect4", one, np.array([3], np.int32))))
for name, fn in calls + calls:
    for _ in range(5):
        fn()
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    print("%-52s %7.1f us per call" % (name, (time.perf_counter() - t0) / 200 * 1e6))
e.close()
