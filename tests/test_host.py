"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol include/azamd.h
declares, the product refuses to run without a GPU (no CPU fallback), and the multi-rank trajectory gather works
over gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "azamd.h")


def _declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(az_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from alphazero_amd import _abi
    lib = ctypes.CDLL(_abi.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), "libazamd.so does not export %s" % name
    # the ctypes table mirrors the header one to one
    assert sorted(_abi.SIGNATURES) == declared
    assert lib.az_version() >= 100
    for game, (words, acts) in enumerate([(2, 7), (1, 9), (4, 9)]):
        assert lib.az_state_words(game) == words and lib.az_n_actions(game) == acts
    assert [lib.az_traj_record_bytes(g) for g in range(3)] == [48, 48, 72]


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU every compute entry point must fail loudly (AZ_ERR_NO_DEVICE), never compute."""
    from alphazero_amd import _abi, engine
    if _abi.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_abi.AzError):
        engine.SelfPlayEngine("TTT", n_games=1, n_playout=4)
    with pytest.raises(_abi.AzError):
        engine.env_step_batch("TTT", np.zeros((1, 1), np.uint64), [0])
    L = _abi.lib()
    st = np.zeros(2, np.uint64)
    rc = L.az_env_status_batch(0, 0, 1, _abi.ptr(st), None, None, None, None)
    assert rc == -2 and b"no usable HIP device" in L.az_last_error()
    cfg = _abi.az_config()
    cfg.game, cfg.n_games, cfg.n_playout = 0, 1, 1
    h = ctypes.c_void_p()
    assert L.az_engine_create(ctypes.byref(cfg), ctypes.byref(h)) == -2 and not h.value


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under alphazero_amd/ may reference it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "alphazero_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"\boracle\b", txt) and not f.endswith((".hip", ".hpp")):
                    bad.append(f)
                if re.search(r"az_oracle|orc_", txt):
                    bad.append(f)
    assert not bad, bad


GLOO_WORKER = r"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, %r)
from alphazero_amd.collector import gather_records
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
# ragged: rank 0 contributes 5 records, rank 1 contributes 0 then 3
for rnd, counts in enumerate([(5, 0), (2, 3), (0, 0)]):
    n = counts[rank]
    local = (torch.arange(n * 48, dtype=torch.int64).reshape(n, 48) %% 251 + rank * 3 + rnd).to(torch.uint8)
    out = gather_records(local)
    want = []
    for r, c in enumerate(counts):
        want.append((torch.arange(c * 48, dtype=torch.int64).reshape(c, 48) %% 251 + r * 3 + rnd).to(torch.uint8))
    want = torch.cat(want) if sum(counts) else local
    assert out.shape == want.shape and torch.equal(out, want), (rank, rnd, out.shape)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_trajectory_gather_world_size_2_gloo(tmp_path):
    """N>1 path of bench.py: counts exchange + padded all_gather of packed records, two ranks, CPU (gloo)."""
    script = tmp_path / "w.py"
    script.write_text(GLOO_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def _bench(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_bench_launcher_starts_n_ranks_stub_workload():
    """`python bench.py --gpus 2` outside torchrun must start 2 ranks itself (the driver's SCALE command).  CPU stand-in:
    the stub workload walks launcher -> rendezvous on 127.0.0.1 -> barrier -> MAX/SUM all-reduce -> rank 0's JSON line
    over gloo; n_gpus is what the process group reports."""
    import json
    p = _bench(["--gpus", "2", "--workload", "stub", "--steps", "5", "--warmup", "0"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["units"] == 5 * 1000 + 5 * 1001 and d["steps"] == 5


def test_bench_refuses_rank_count_mismatch():
    """a 1-rank environment must never be reported as an N-GPU run"""
    p = _bench(["--gpus", "2", "--workload", "stub"], {"RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                                                       "MASTER_PORT": "29537"})
    assert p.returncode != 0 and b"refusing" in p.stderr
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_bench_launcher_fails_when_a_rank_fails():
    p = _bench(["--gpus", "2", "--workload", "stub", "--steps", "1"], {"AZ_BENCH_STUB_FAIL_RANK": "1"})
    assert p.returncode != 0
