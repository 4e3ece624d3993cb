#!/usr/bin/env python3
"""Record which rows the REFERENCE's ReplayBuffer (ReplayBuffer.py:10-185) hands out, for tests/golden/replay.npz.

CONTAINER-ONLY (needs /root/reference; nothing on the GPU box runs this).  The reference class is loaded from its file
under a private module name and driven on CPU tensors: seeded synthetic plies are stored one by one, then
`dataloader`, `sample_balanced` and `sample` are called after seeding torch / numpy, and the `value` column of what they
return -- the running number of the ply that was stored there -- is saved as the list of selected rows.

    python tests/golden/make_fixtures_replay.py
"""
import importlib.util
import json
import os

import numpy as np
import torch

REF = os.environ.get("AZ_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

# (capacity, plies stored, replay_ratio, balance_done_value, P(done), outcomes present, torch seed)
CASES = [
    (64, 40, 0.1, True, 0.2, (-1, 0, 1), 1),        # <= 1000 rows: everything is handed out
    (64, 100, 0.1, True, 0.2, (-1, 1), 2),          # ring wrapped
    (4000, 3000, 0.1, True, 0.15, (-1, 0, 1), 3),   # budget = 300, 60 game-ending plies over three outcomes
    (4000, 3000, 0.25, False, 0.15, (-1, 0, 1), 4),  # no balancing
    (4000, 2500, 0.1, True, 0.0, (1,), 5),          # no game-ending ply at all
    (4000, 2500, 0.1, True, 0.004, (-1, 0, 1), 6),  # fewer game-ending plies than their quota
    (2000, 1500, 0.1, True, 1.0, (0, 1), 7),        # only game-ending plies
]


def synthetic_plies(n, p_done, outcomes, seed):
    """the rows both sides store: ply k has value k, a state/prob made from k, a random outcome and done flag"""
    rng = np.random.default_rng(seed)
    done = rng.random(n) < p_done
    winner = rng.choice(np.asarray(outcomes, dtype=np.int32), size=n)
    return done, winner


def store_all(buf, n, done, winner):
    for k in range(n):
        s = np.full((1, 1, 2), k, dtype=np.float32)
        buf.store(s, np.asarray([k, -k], dtype=np.float32), float(k), int(winner[k]), s + 0.5, bool(done[k]))


def main():
    spec = importlib.util.spec_from_file_location("az_reference_replaybuffer", os.path.join(REF, "ReplayBuffer.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out, meta = {}, []
    for i, (cap, n, ratio, bal, p_done, outcomes, seed) in enumerate(CASES):
        done, winner = synthetic_plies(n, p_done, outcomes, 1000 + seed)
        buf = ref.ReplayBuffer(1, cap, 2, 1, 2, replay_ratio=ratio, device="cpu", balance_done_value=bal)
        store_all(buf, n, done, winner)
        m = dict(capacity=cap, n=n, replay_ratio=ratio, balance=bal, p_done=p_done, outcomes=list(outcomes), seed=seed,
                 len=len(buf), ptr=int(buf._ptr), is_full=bool(buf.is_full()))
        torch.manual_seed(seed)
        out["case%d_dataloader" % i] = buf.dataloader(32).dataset.tensors[2].reshape(-1).numpy().astype(np.int64)
        torch.manual_seed(seed + 100)
        dl = buf.sample_balanced(32)
        out["case%d_balanced" % i] = dl.dataset.tensors[2].reshape(-1).numpy().astype(np.int64)
        m["balanced_batch"] = int(dl.batch_size)
        np.random.seed(seed)
        out["case%d_sample" % i] = buf.sample(50)[2].reshape(-1).numpy().astype(np.int64)
        meta.append(m)
    out["meta"] = np.frombuffer(json.dumps(dict(cases=meta, torch=torch.__version__, numpy=np.__version__,
                                                source="ReplayBuffer.py:10-185 of the reference, CPU tensors")).encode(),
                                dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "replay.npz"), **out)
    for m in meta:
        print(m)


if __name__ == "__main__":
    main()
