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
