#!/usr/bin/env python3
"""Soak test of az_search's forms: whole games (search, sampled move, advance, until every game is over; several games per
engine so that the evaluation cache is warm and full of speculative entries) in the lockstep chain, the rolling form, and
the rolling form with speculative rows -- each in a process of its own (the switches are read once) -- must leave the
same SHA-256 over every root's children (actions, visits, Q bits, prior bits) and every move played.
Usage on the GPU box: python3 tools/soak_search.py [slots] [games_per_engine] [playouts]"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
from alphazero_amd.engine import SelfPlayEngine
G, games, S = %d, %d, %d
blob = np.fromfile(%r, dtype='<f4')
e = SelfPlayEngine('Connect4', n_games=G, n_playout=S, c_puct=1.25, evaluator='cnn', flip='random', reuse_tree=True,
                   dirichlet_alpha=0.7, rng_seed=23, move='sample')
e.load_net(blob)
h = hashlib.sha256()
plies = 0
for gi in range(games):
    e.reset_games()
    while True:
        _, _, dn = e.get_roots()
        if dn.all():
            break
        e.search()
        rc = e.root_children()
        a, _ = e.choose_moves(temp=1.0 if plies %% 50 < 8 else 1e-3)
        for k in ('actions', 'visits', 'q', 'prior', 'n', 'root_visits'):
            h.update(np.ascontiguousarray(rc[k]).tobytes())
        h.update(a.tobytes())
        e.advance(a)
        plies += 1
st = e.stats()
print(h.hexdigest(), plies, st['playouts'], st['leaf_evals'], st['cache_hits'], st['terminal_hits'])
"""


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    games = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    code = CODE % (ROOT, G, games, S, os.path.join(ROOT, "tests", "golden", "weights", "AZ_Connect4_CNN_current.f32"))
    ref = None
    for name, env in (("chain", dict(AZ_SEARCH_ROLL="0")), ("rolling", dict(AZ_SEARCH_ROLL="1", AZ_SPEC="0")),
                      ("rolling + speculation", dict(AZ_SEARCH_ROLL="1", AZ_SPEC_MAX_G="100000"))):
        base = {k: v for k, v in os.environ.items() if k not in ("AZ_SEARCH_ROLL", "AZ_SPEC", "AZ_SPEC_MAX_G")}
        p = subprocess.run([sys.executable, "-c", code], env=dict(base, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if p.returncode != 0:
            print(name, "FAILED", p.stderr.decode()[-2000:])
            sys.exit(1)
        out = p.stdout.decode().strip().splitlines()[-1].split()
        print("%-22s sha %s  plies %s playouts %s leaf_evals %s cache_hits %s terminal %s" % (name, out[0][:16], *out[1:]), flush=True)
        key = (out[0], out[1], out[2], out[5])
        if ref is None:
            ref = key
        if key != ref:
            print("MISMATCH", name)
            sys.exit(1)
    print("identical")


if __name__ == "__main__":
    main()
