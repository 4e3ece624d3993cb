#!/usr/bin/env python3
"""Golden vectors for the PURE Monte-Carlo tree search (the Elo arena's opponent): outputs of the reference itself.

CONTAINER-ONLY (imports /root/reference through _refload; listed in .gpurunignore).  Writes tests/golden/pure_mcts.npz and
tests/golden/pure_fixtures.json.

The reference's MCTS class (MCTS.py:90-135) is driven unmodified with utils.policy_value_fn (utils.py:43-46).  Harness
rules (the reference never seeds np.random):
  * np.random.rand() in env.random_flip (one draw per playout)  -> a recorded bit stream (as in make_fixtures.py);
  * utils.rollout_policy_fn (np.random.rand per valid move, utils.py:49-52) -> a position-hashed choice:
    idx = (mix64(fold(words, seed) + 0x524F4C4C) >> 33) % n_valid, returned as one-hot "probabilities", so that the
    arg-max taken by evaluate_rollout (utils.py:32) plays that move.  Same rule in oracle/az_oracle_pure.c and az_pure.hip.
Recorded per ply: root children (actions in insertion order, visits, Q as float64 bit patterns), root N and Q, the
action MCTS.get_action returns; and a table of TreeNode.UCT values (MCTS.py:57-63) as float64 bit patterns.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402  (reuses the loader, pack_planes, mix64, RngPatch, NBTTT harness)

ref = mf.ref
M64 = mf.M64


def rollout_pick(game, env, seed, nv):
    h = seed
    for w in mf.pack_planes(game, env.current_state()):
        h = mf.mix64(h ^ w)
    return (mf.mix64((h + 0x524F4C4C) & M64) >> 33) % nv


class HashedRollout:
    def __init__(self, game, seed):
        self.game, self.seed = game, seed

    def __call__(self, env):  # stands in for utils.rollout_policy_fn
        valid = env.valid_move()
        probs = np.zeros(len(valid))
        probs[rollout_pick(self.game, env, self.seed, len(valid))] = 1.0
        return list(zip(valid, probs))

    def __enter__(self):
        self._orig = ref.utils.rollout_policy_fn
        ref.utils.rollout_policy_fn = self
        return self

    def __exit__(self, *exc):
        ref.utils.rollout_policy_fn = self._orig


def gen_case(game, NB, seed, S, c_puct, flips, reuse, plies, opening=None):
    mcts = ref.MCTS.MCTS(ref.utils.policy_value_fn, c_puct, S)
    env = mf.new_env(game, NB, opening)
    rec = dict(actions=[], visits=[], q_bits=[], n_children=[], root_N=[], root_Q_bits=[], action=[])
    with HashedRollout(game, seed), mf.RngPatch(flips) as rp:
        for ply in range(plies):
            if env.done():
                break
            action, node = mcts.get_action(env)           # MCTS.py:125-128, unmodified
            root = mcts.root
            acts = list(root.children.keys())
            pad = 9 - len(acts)
            rec["actions"].append(acts + [-1] * pad)
            rec["visits"].append([root.children[a].n_visits for a in acts] + [0] * pad)
            rec["q_bits"].append(np.array([np.float64(root.children[a].Q).view(np.uint64) for a in acts]
                                         + [np.uint64(0)] * pad, dtype=np.uint64))
            rec["n_children"].append(len(acts))
            rec["root_N"].append(root.n_visits)
            rec["root_Q_bits"].append(np.uint64(np.float64(root.Q).view(np.uint64)))
            rec["action"].append(int(action))
            mcts.prune_root(int(action) if reuse else -1)  # MCTSPlayer (player.py:79-82) ends up with a fresh tree
            env.step(int(action))
        n_rand = rp.pos
    out = {k: np.array(v, dtype=np.uint64) if k.endswith("bits") else np.array(v) for k, v in rec.items()}
    out["winner"] = np.array(env.winPlayer() if env.done() else 99)
    out["n_rand"] = np.array(n_rand)
    return out


def gen_uct(rng):
    """TreeNode.UCT (MCTS.py:57-63) on random node statistics, float64 bit patterns."""
    n = 4000
    Np = rng.integers(1, 4000, size=n)
    nv = np.minimum(rng.integers(0, 400, size=n), Np)
    c = rng.choice([1.0, 1.25, 4.0], size=n)
    Q = np.where(nv > 0, np.round(rng.uniform(-1, 1, size=n) * nv) / np.maximum(nv, 1), 0.0)
    bits = []
    for i in range(n):
        parent = ref.MCTS.TreeNode(None, 1)
        parent.n_visits = int(Np[i])
        node = ref.MCTS.TreeNode(parent, 1)
        node.n_visits = int(nv[i])
        node.Q = float(Q[i])
        bits.append(np.float64(node.UCT(float(c[i]), 500)).view(np.uint64))
    return dict(uct_Np=Np, uct_n=nv, uct_c=c, uct_Q=Q, uct_bits=np.array(bits, dtype=np.uint64))


def main():
    rng = np.random.default_rng(20261111)
    NB, _, _ = mf.load_nbttt()
    cases, meta = {}, {}

    def add(name, **kw):
        r = gen_case(NB=NB, **kw)
        for k, v in r.items():
            cases[name + "__" + k] = v
        if kw.get("flips") is not None:
            cases[name + "__flips"] = np.array(kw["flips"], dtype=np.uint8)
        meta[name] = dict(game=kw["game"], seed=kw["seed"], S=kw["S"], c_puct=kw["c_puct"], reuse=kw["reuse"],
                          has_flips=kw.get("flips") is not None, opening=kw.get("opening"), plies=len(r["action"]))
        print(name, "plies", len(r["action"]), "winner", int(r["winner"]))

    def bits(n):
        return [int(x) for x in rng.integers(0, 2, size=n)]

    add("c4_s200_c1_flips_fresh", game="Connect4", seed=11, S=200, c_puct=1.0, flips=bits(200 * 44), reuse=False, plies=44)
    add("c4_s300_c4_flips_reuse", game="Connect4", seed=12, S=300, c_puct=4.0, flips=bits(300 * 44), reuse=True, plies=44)
    add("c4_s1000_c1_noflip_fresh", game="Connect4", seed=13, S=1000, c_puct=1.0, flips=None, reuse=False, plies=6)
    add("ttt_s150_c1_flips_reuse", game="TTT", seed=21, S=150, c_puct=1.0, flips=bits(150 * 9), reuse=True, plies=9)
    add("ttt_s100_c4_flips_fresh", game="TTT", seed=22, S=100, c_puct=4.0, flips=bits(100 * 9), reuse=False, plies=9)
    add("nbttt_s200_c15_reuse", game="NBTTT", seed=31, S=200, c_puct=1.5, flips=None, reuse=True, plies=30,
        opening=[3, 4, 1])
    cases.update(gen_uct(rng))
    np.savez_compressed(os.path.join(HERE, "pure_mcts.npz"), **cases)
    with open(os.path.join(HERE, "pure_fixtures.json"), "w") as f:
        json.dump(dict(numpy=np.__version__, cases=meta,
                       note="outputs of the reference's MCTS/utils (pure MCTS); see make_fixtures_pure.py"), f, indent=1,
                  sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
