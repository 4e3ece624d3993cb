"""Round-3 parity coverage on the GPU, all through the C ABI:
  * split invariance ("RNG streams keyed by global game id so results are independent of the GPU count"): one
    engine with game ids base 0 / stride 1 against 2 and 4 engines with base r / stride W, trajectories equal bit for bit
    by game id -- the multi-GPU partitioning pinned on one GPU;
  * rolling == lockstep once at the shape bench.py times (2 048 slots x 500 playouts, real net);
  * az_set_active in the middle of rolling self-play: a parked slot's game, tree and trajectory do not move, and the games
    come out the same as without parking;
  * az_net_load on a live engine with a blob that is refused: the engine keeps the weights it had;
  * AZ_TREE_GPW values that are not a power of two: every game is still searched;
  * the evaluation cache and the rolling tree step's time budget change which work runs when, never what a game does.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def E():
    from alphazero_amd import engine
    return engine


@pytest.fixture(scope="module")
def blob():
    return np.fromfile(os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32"), dtype="<f4")


def by_game(t):
    out = {}
    for i in np.argsort(t["game_id"], kind="stable"):
        out.setdefault(int(t["game_id"][i]), []).append(i)
    return {k: np.array(v) for k, v in out.items()}


def assert_same_games(ta, tb, ids=None):
    """two trajectory fetches hold the same games: states, actions, movers, done flags, z, and pi bit patterns, by game id"""
    ga, gb = by_game(ta), by_game(tb)
    if ids is None:
        assert sorted(ga) == sorted(gb)
        ids = sorted(ga)
    for gid in ids:
        ia, ib = ga[gid], gb[gid]
        assert len(ia) == len(ib), gid
        for k in ("states", "action", "player", "done", "z"):
            assert np.array_equal(ta[k][ia], tb[k][ib]), (gid, k)
        assert np.array_equal(ta["pi"][ia].view(np.uint32), tb["pi"][ib].view(np.uint32)), gid


def merge(ts):
    return {k: np.concatenate([t[k] for t in ts]) for k in ts[0]}


# ------------------------------------------------------------------------------------------------ split invariance
@pytest.mark.parametrize("rolling", [True, False])
def test_game_set_independent_of_the_split(E, blob, rolling):
    """The set of games a job plays does not depend on how its slots are split over GPUs: engine A
    = the whole job on one GPU (64 slots, ids 0, 1, 2, ...), engines B_r = the ranks of a W-GPU job (64 / W slots each, ids
    r, r + W, r + 2W, ...), W = 2 and 4; production settings (real net, Dirichlet 0.7 at every expansion, random flips,
    sampled moves, tree reuse, refill).  Union of the ranks' trajectories == A's, bit for bit, game id by game id."""
    Gn, S, M = 64, 40, 256

    def run(slots, target, base, stride):
        eng = E.SelfPlayEngine("Connect4", n_games=slots, n_playout=S, c_puct=1.25, evaluator="cnn", flip="random",
                               reuse_tree=True, dirichlet_alpha=0.7, rng_seed=11, move="sample",
                               game_id_base=base, game_id_stride=stride)
        eng.load_net(blob)
        n = (eng.selfplay_rolling if rolling else eng.selfplay)(target, temp=1.0, first_n_steps=6)
        assert n == target
        t = eng.traj_fetch()
        eng.close()
        return t

    whole = run(Gn, M, 0, 1)
    assert sorted(by_game(whole)) == list(range(M))
    for W in (2, 4):
        parts = [run(Gn // W, M // W, r, W) for r in range(W)]
        for r, t in enumerate(parts):
            assert sorted(by_game(t)) == list(range(r, M, W)), (W, r)
        assert_same_games(whole, merge(parts))
    # and a different seed does give different games (the comparison above is not vacuous)
    other = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, c_puct=1.25, evaluator="cnn", flip="random",
                             reuse_tree=True, dirichlet_alpha=0.7, rng_seed=12, move="sample")
    other.load_net(blob)
    other.selfplay_rolling(Gn, temp=1.0, first_n_steps=6)
    t2 = other.traj_fetch()
    other.close()
    ga, gb = by_game(whole), by_game(t2)
    assert sum(len(ga[i]) != len(gb[i]) or not np.array_equal(whole["action"][ga[i]], t2["action"][gb[i]])
               for i in range(Gn)) > Gn // 2


def test_nbttt_openings_keyed_by_game_id(E):
    """configs[4]: the two random opening plies of a nine-board game (NBTTT/env.py:15-23) are drawn from the game id, so
    the ranks of a split job start exactly the games the unsplit job starts."""
    Gn = 64
    a = E.SelfPlayEngine("NBTTT", n_games=Gn, n_playout=8, c_puct=1.5, evaluator="hash", eval_seed=3, rng_seed=5)
    whole = a.get_roots()[0]
    a.close()
    assert len({tuple(r) for r in whole}) > Gn // 2
    for W in (2, 4):
        for r in range(W):
            b = E.SelfPlayEngine("NBTTT", n_games=Gn // W, n_playout=8, c_puct=1.5, evaluator="hash", eval_seed=3,
                                 rng_seed=5, game_id_base=r, game_id_stride=W)
            assert np.array_equal(b.get_roots()[0], whole[r::W]), (W, r)
            b.close()


# ------------------------------------------------------------------------------------------------ bench shape
def test_rolling_equals_lockstep_at_bench_shape(E, blob):
    """The mode bench.py times (rolling rounds) against the mode the oracle tests pin (lockstep plies) at the size that is
    timed: 2 048 slots x 500 playouts, real net, eval mode, random flips, sampled moves, tree reuse, refill; every slot's
    first game and 256 refilled ones, compared by game id (states, actions, movers, z, pi bits)."""
    Gn, S, N = 2048, 500, 2048 + 256

    def run(rolling):
        eng = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, c_puct=1.25, evaluator="cnn", flip="random",
                               reuse_tree=True, rng_seed=21, move="sample")
        assert eng.n_cohorts == 4
        eng.load_net(blob)
        n = (eng.selfplay_rolling if rolling else eng.selfplay)(N, temp=1.0, first_n_steps=10)
        assert n == N
        t, st = eng.traj_fetch(), eng.stats()
        eng.close()
        return t, st

    tl, sl = run(False)
    tr, sr = run(True)
    assert sorted(by_game(tl)) == list(range(N))
    assert_same_games(tl, tr)
    for k in ("playouts", "terminal_hits", "plies", "games_finished", "levels", "children"):
        assert sl[k] == sr[k], (k, sl[k], sr[k])
    for st in (sl, sr):
        assert st["leaf_evals"] + st["cache_hits"] + st["terminal_hits"] == st["playouts"] == st["plies"] * S
    print("network evaluations lockstep %d rolling %d, cache hits %d / %d" % (sl["leaf_evals"], sr["leaf_evals"], sl["cache_hits"], sr["cache_hits"]))


# ------------------------------------------------------------------------------------------------ parking
def test_parking_slots_in_the_middle_of_rolling_selfplay(E, blob):
    """az_set_active between az_selfplay_rounds calls: parked slots hold a delivered leaf at that moment (rolling mode keeps
    one playout of every live game in flight).  While parked their root position, root statistics and children do not
    move and nothing of theirs is recorded; after they are resumed the job produces, game id by game id, exactly the games
    of a run that never parked anything."""
    Gn, S, M = 48, 30, 96      # 96 games x <= 42 plies fit the engine's trajectory buffer (2 x slots x 42 rows)

    def make():
        eng = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, c_puct=1.25, evaluator="cnn", flip="random",
                               reuse_tree=True, dirichlet_alpha=0.7, rng_seed=31, move="sample", n_cohorts=2)
        eng.load_net(blob)
        eng.set_game_target(M)
        return eng

    def finish(eng):
        for _ in range(400):
            eng.selfplay_rounds(S, temp=1.0, first_n_steps=6, refill=True)
            eng.sync()
            if eng.traj_count()[0] >= M:
                break
        assert eng.traj_count()[0] == M
        t = eng.traj_fetch()
        eng.close()
        return t

    plain = finish(make())

    eng = make()
    eng.selfplay_rounds(S * 3 + 17, temp=1.0, first_n_steps=6, refill=True)     # stop in the middle of a move
    eng.sync()
    parked = np.arange(Gn) % 3 == 0
    eng.set_active((~parked).astype(np.uint8))
    roots0, rc0 = eng.get_roots()[0].copy(), eng.root_children()
    eng.selfplay_rounds(S * 4 + 5, temp=1.0, first_n_steps=6, refill=True)
    eng.sync()
    roots1, rc1 = eng.get_roots()[0], eng.root_children()
    assert np.array_equal(roots0[parked], roots1[parked])
    for k in ("actions", "visits", "n", "root_visits"):
        assert np.array_equal(rc0[k][parked], rc1[k][parked]), k
    assert np.array_equal(rc0["q"][parked].view(np.uint32), rc1["q"][parked].view(np.uint32))
    assert not np.array_equal(roots0[~parked], roots1[~parked])                  # the others did play on
    eng.set_active(np.ones(Gn, np.uint8))
    assert_same_games(plain, finish(eng))


# ------------------------------------------------------------------------------------------------ hot reload, refused blob
@pytest.mark.parametrize("kernel", ["split", "wino"])
def test_refused_blob_leaves_the_live_network_in_place(E, blob, kernel):
    """Pipeline.py:87 hot reload with a blob from a diverged training step (a NaN / inf, or the wrong size): az_net_load
    refuses it AND the engine goes on with the weights it had -- forward outputs bit-identical, searches still run."""
    from alphazero_amd._abi import AzError
    rng = np.random.default_rng(0)
    d = np.load(os.path.join(G, "cnn.npz"))
    from oracle import oracle as orc
    packed = np.array([orc.Env.from_cells("Connect4", c).packed() for c in d["cells"][:64]])
    eng = E.SelfPlayEngine("Connect4", n_games=32, n_playout=24, c_puct=1.25, evaluator="cnn", flip="never")
    eng.load_net(blob, kernel=kernel)
    before = eng.net_forward(packed)
    for what in ("nan", "inf", "short"):
        bad = blob.copy()
        if what == "short":
            bad = bad[:-5]
        else:
            bad[int(rng.integers(0, bad.size))] = np.nan if what == "nan" else np.inf
        with pytest.raises(AzError):
            eng.load_net(bad)
        after = eng.net_forward(packed)
        for x, y in zip(before, after):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), what
        assert eng.k3_kernel == kernel
    eng.search()
    rc = eng.root_children()
    assert (rc["root_visits"] == 24).all() and (rc["visits"].sum(1) == 23).all()
    eng.close()


# ------------------------------------------------------------------------------------------------ AZ_TREE_GPW
def test_fused_search_covers_every_game_for_any_gpw():
    """AZ_TREE_GPW (a documented measurement switch) is read once per process, so each value runs in a process of its own:
    values that are not a power of two used to size the grid for more games per workgroup than the kernel launched had,
    and the last games were never searched.  Every game must reach root N = S for 1..8 (Connect4) and the pure search."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from alphazero_amd.engine import SelfPlayEngine\n"
        "for game, ev, G in (('Connect4', 'hash', 1000), ('TTT', 'uniform', 333), ('Connect4', 'rollout', 200)):\n"
        "    e = SelfPlayEngine(game, n_games=G, n_playout=20, evaluator=ev, eval_seed=1)\n"
        "    e.search()\n"
        "    rc = e.root_children()\n"
        "    assert (rc['root_visits'] == 20).all(), (game, ev, int((rc['root_visits'] != 20).sum()))\n"
        "    e.close()\n" % ROOT)
    for gpw in (1, 2, 3, 5, 6, 7, 8, 64):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AZ_TREE_GPW=str(gpw)),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, (gpw, p.stderr.decode()[-1500:])


# ------------------------------------------------------------------------------------------------ in-place playout budget
def test_rolling_time_budget_changes_nothing_in_any_game():
    """The rolling tree step stops starting in-place playouts once its launch is AZ_ROLL_BUDGET_US old (read once per process,
    so each value runs in a process of its own): which playout runs in which round depends on the clock, what a game does must
    not.  Budget off with the old count of 6, the default, a budget so short that every round runs one playout per game, and
    one so long that only the safety count (here 3 and 40) ends a round: the same games bit for bit, the same counters."""
    import hashlib
    import json
    import subprocess
    import sys
    code = (
        "import sys, json, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from alphazero_amd.engine import SelfPlayEngine\n"
        "blob = np.fromfile(%r, dtype='<f4')\n"
        "e = SelfPlayEngine('Connect4', n_games=160, n_playout=48, c_puct=1.25, evaluator='cnn', flip='random', reuse_tree=True,\n"
        "                   dirichlet_alpha=0.7, rng_seed=5, move='sample', n_cohorts=2)\n"
        "e.load_net(blob)\n"
        "assert e.selfplay_rolling(320, temp=1.0, first_n_steps=6) == 320\n"
        "t, s = e.traj_fetch(), e.stats()\n"
        "o = np.lexsort((np.arange(len(t['game_id'])), t['game_id']))\n"
        "h = hashlib.sha256()\n"
        "for k in ('game_id', 'states', 'action', 'player', 'done', 'z', 'pi'):\n"
        "    h.update(np.ascontiguousarray(t[k][o]).tobytes())\n"
        "print(json.dumps(dict(sha=h.hexdigest(), rows=int(len(o)), **{k: int(s[k]) for k in ('playouts', 'terminal_hits', 'levels',\n"
        "      'children', 'plies', 'games_finished')}, nonterminal=int(s['leaf_evals'] + s['cache_hits']))))\n"
        % (ROOT, os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32")))
    res = {}
    for name, env in (("count6", dict(AZ_ROLL_BUDGET_US="0", AZ_ROLL_CAP="6")), ("default", {}),
                      ("tiny", dict(AZ_ROLL_BUDGET_US="0.05")), ("count3", dict(AZ_ROLL_BUDGET_US="100000", AZ_ROLL_CAP="3")),
                      ("count40", dict(AZ_ROLL_BUDGET_US="100000", AZ_ROLL_CAP="40"))):
        e = {k: v for k, v in os.environ.items() if k not in ("AZ_ROLL_BUDGET_US", "AZ_ROLL_CAP")}
        p = subprocess.run([sys.executable, "-c", code], env=dict(e, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, (name, p.stderr.decode()[-1500:])
        res[name] = json.loads(p.stdout.decode().strip().splitlines()[-1])
    for name in res:
        assert res[name] == res["count6"], (name, res[name], res["count6"])
    assert res["default"]["rows"] > 320 * 7


# ------------------------------------------------------------------------------------------------ az_search, rolling form
def test_az_search_rolling_form_grows_the_same_trees():
    """az_search on an AZ_EVAL_CNN engine of up to 512 games runs on the rolling tree step (terminal playouts and cache hits
    finished in place, a round only for leaves the network has to see) and takes a waiting leaf's unvisited siblings and its
    children along into the K3 launch (speculative rows: their answers go into the evaluation cache); larger engines keep
    the lockstep chain.  AZ_SEARCH_ROLL=0|1 forces one form for every size, AZ_SPEC=0 switches the speculation off (read once
    per process): all must grow the same trees -- actions, visit counts, Q and prior bit patterns of every root, and the
    moves played -- for 1, 5, 200 and 600 games (600: two cohorts), eval and train mode."""
    import hashlib
    import subprocess
    import sys
    code = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from alphazero_amd.engine import SelfPlayEngine\n"
        "blob = np.fromfile(%r, dtype='<f4')\n"
        "h = hashlib.sha256()\n"
        "for G, S, noise, plies in ((1, 300, 0.0, 12), (5, 120, 0.7, 10), (200, 64, 0.7, 6), (600, 40, 0.0, 4)):\n"
        "    e = SelfPlayEngine('Connect4', n_games=G, n_playout=S, c_puct=1.25, evaluator='cnn', flip='random', reuse_tree=True,\n"
        "                       dirichlet_alpha=noise, rng_seed=11, move='sample')\n"
        "    e.load_net(blob)\n"
        "    for ply in range(plies):\n"
        "        e.search()\n"
        "        rc = e.root_children()\n"
        "        assert (rc['root_visits'] == (S if ply == 0 else rc['root_visits'])).all()\n"
        "        a, _ = e.choose_moves(temp=1.0)\n"
        "        for k in ('actions', 'visits', 'q', 'prior', 'n', 'root_visits'):\n"
        "            h.update(np.ascontiguousarray(rc[k]).tobytes())\n"
        "        h.update(a.tobytes())\n"
        "        e.advance(a)\n"
        "    st = e.stats()\n"
        "    h.update(np.array([st[k] for k in ('playouts', 'terminal_hits', 'levels', 'children')], dtype=np.int64).tobytes())\n"
        "    assert st['playouts'] == G * S * plies or G > 1, st\n"
        "    e.close()\n"
        "print(h.hexdigest())\n" % (ROOT, os.path.join(G, "weights", "AZ_Connect4_CNN_current.f32")))
    res = {}
    for name, env in (("chain", dict(AZ_SEARCH_ROLL="0")), ("rolling", dict(AZ_SEARCH_ROLL="1", AZ_SPEC="0")),
                      ("speculating", dict(AZ_SEARCH_ROLL="1", AZ_SPEC_MAX_G="4096")), ("default", {})):
        e = {k: v for k, v in os.environ.items() if k not in ("AZ_SEARCH_ROLL", "AZ_SPEC", "AZ_SPEC_MAX_G")}
        p = subprocess.run([sys.executable, "-c", code], env=dict(e, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
        assert p.returncode == 0, (name, p.stderr.decode()[-1500:])
        res[name] = p.stdout.decode().strip().splitlines()[-1]
    assert res["chain"] == res["rolling"] == res["speculating"] == res["default"], res


# ------------------------------------------------------------------------------------------------ evaluation cache
@pytest.mark.parametrize("rolling", [True, False])
@pytest.mark.parametrize("noise", [0.0, 0.7])
def test_evaluation_cache_changes_nothing_but_the_work(E, blob, rolling, noise):
    """az_config.eval_cache_entries: a leaf whose position the network has already evaluated is answered from the table with
    the bits K3 delivers, so a job's games are the same bit for bit with the cache off, on at its default size, and on with a
    table so small (256 entries) that entries are overwritten all the time; only leaf_evals moves to cache_hits."""
    Gn, S, N = 96, 40, 192
    out = {}
    for name, cache in (("off", False), ("auto", 0), ("tiny", 256)):
        eng = E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, c_puct=1.25, evaluator="cnn", flip="random",
                               reuse_tree=True, dirichlet_alpha=noise, rng_seed=41, move="sample", n_cohorts=3, eval_cache=cache)
        assert eng.eval_cache_entries == {"off": -1, "auto": 1 << 24, "tiny": 256}[name]     # 96 x 128 Ki rounded up to a power of two
        eng.load_net(blob)
        n = (eng.selfplay_rolling if rolling else eng.selfplay)(N, temp=1.0, first_n_steps=6)
        assert n == N
        out[name] = (eng.traj_fetch(), eng.stats())
        eng.close()
    t0, s0 = out["off"]
    assert s0["cache_hits"] == 0 and s0["leaf_evals"] + s0["terminal_hits"] == s0["playouts"]
    for name in ("auto", "tiny"):
        t, s = out[name]
        assert_same_games(t0, t)
        for k in ("playouts", "terminal_hits", "plies", "games_finished", "levels", "children"):
            assert s[k] == s0[k], (name, k)
        assert s["leaf_evals"] + s["cache_hits"] == s0["leaf_evals"] and s["cache_hits"] > 0, name
    assert out["auto"][1]["cache_hits"] > out["tiny"][1]["cache_hits"] > 0
    assert out["auto"][1]["cache_hits"] > 0.3 * s0["leaf_evals"]        # games from one network share their openings
    print("leaf evaluations without cache %d; cache hits auto %d, 256 entries %d"
          % (s0["leaf_evals"], out["auto"][1]["cache_hits"], out["tiny"][1]["cache_hits"]))


def test_auto_sized_cache_yields_to_the_memory_that_is_left(E, blob):
    """An auto-sized table (eval_cache_entries = 0) takes at most half of the device memory that is free when the engine is created;
    a size the caller names is allocated as named or the engine is refused, loudly."""
    import torch
    from alphazero_amd._abi import AzError
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(max(free - (12 << 30), 0), dtype=torch.uint8, device="cuda")    # leave ~12 GB
    try:
        left, _ = torch.cuda.mem_get_info()
        eng = E.SelfPlayEngine("Connect4", n_games=2048, n_playout=8, c_puct=1.25, evaluator="cnn", flip="random",
                               reuse_tree=True, rng_seed=3, move="sample", nodes_per_game=4096)
        n = eng.eval_cache_entries
        assert (1 << 16) <= n < (1 << 28) and n * 64 <= left // 2 + (1 << 20), (n, left)     # 2 048 slots would ask for 2^28 = 17 GB
        eng.load_net(blob)
        assert eng.selfplay_rolling(64, temp=1.0, first_n_steps=4) == 64
        assert eng.stats()["cache_hits"] > 0
        eng.close()
        with pytest.raises(AzError):
            E.SelfPlayEngine("Connect4", n_games=64, n_playout=8, evaluator="cnn", eval_cache=1 << 29)    # 34 GB: not there
    finally:
        del hog
        torch.cuda.empty_cache()


def test_evaluation_cache_is_emptied_by_a_hot_reload(E, blob):
    """Pipeline.py:87: after az_net_load on a live engine no leaf may be answered with the previous weights' output.  An engine
    that has filled its cache under checkpoint A and is then given checkpoint B must search exactly like a cache-less engine
    that only ever saw B (same roots, same recorded flips), and again like A's after reloading A."""
    blob_b = np.fromfile(os.path.join(G, "weights", "AZ2_Connect4_CNN_best.f32"), dtype="<f4")
    Gn, S = 64, 60
    rng = np.random.default_rng(9)
    flips = rng.integers(0, 2, size=(Gn, S * 2)).astype(np.uint8)

    def searched(eng):
        eng.reset_games()
        eng.set_flip_stream(flips)
        res = []
        for _ in range(2):
            eng.search()
            rc = eng.root_children()
            a, _ = eng.choose_moves(temp=0.0)
            res.append((rc["actions"].copy(), rc["visits"].copy(), rc["q"].view(np.uint32).copy(),
                        rc["prior"].view(np.uint32).copy(), a.copy()))
            eng.advance(a)
        return res

    def same(x, y):
        return all(np.array_equal(u, v) for p, q in zip(x, y) for u, v in zip(p, q))

    mk = lambda cache: E.SelfPlayEngine("Connect4", n_games=Gn, n_playout=S, c_puct=1.25, evaluator="cnn", flip="stream",
                                        reuse_tree=True, move="first_max", eval_cache=cache)
    plain = mk(False)
    plain.load_net(blob)
    ref_a = searched(plain)
    plain.load_net(blob_b)
    ref_b = searched(plain)
    plain.close()
    assert not same(ref_a, ref_b)
    eng = mk(0)
    eng.load_net(blob)
    assert same(searched(eng), ref_a)
    assert same(searched(eng), ref_a) and eng.stats()["cache_hits"] > Gn * S      # second pass: nearly every leaf is a hit
    eng.load_net(blob_b)
    h0 = eng.stats()["cache_hits"]
    assert same(searched(eng), ref_b)
    # identical games: only the first one to reach a position misses; none of A's entries may answer
    eng.load_net(blob)
    assert same(searched(eng), ref_a)
    assert eng.stats()["cache_hits"] > h0
    eng.close()
