"""Batched evaluation arenas: G games between two players in lockstep on the device.

Reference: Pipeline.update_elo (Pipeline.py:125-135) plays the AlphaZero player against `MCTSPlayer(1, 1000)` and
Pipeline.select_best_player (Pipeline.py:137-163) plays `NetworkPlayer(net, False)` against the best net, one game at a
time through Game.start_play (game.py:12-33): per move `player.get_action(env)`, `player.reset_player()` (fresh tree
for every move), `env.step(action)`.  Here every game of the batch is at the same ply, so the side to move handles all
of them at once and the K1 kernel steps the positions.
"""
import numpy as np

from .engine import env_status_batch, env_step_batch


class SearchSide:
    """a searching player (AlphaZeroPlayer in eval mode, or MCTSPlayer): `engine` re-rooted at the current positions
    with fresh trees (reset_player, game.py:21), n_playout playouts, then the move of temp 0 (player.py:106-111:
    uniform among the most visited moves with move="sample", the first of them with move="first_max")."""

    def __init__(self, engine):
        self.engine = engine

    def moves(self, states, status):
        self.engine.set_roots(states)
        self.engine.search()
        actions, _ = self.engine.choose_moves(temp=0.0)
        return actions


class NetworkSide:
    """the raw-policy player NetworkPlayer (player.py:31-59): one network call per move; the policy is restricted to
    the valid moves and renormalised (player.py:47-50); deterministic -> arg-max, else sampled (np.random.choice in the
    reference, a numpy Generator here)."""

    def __init__(self, engine_with_net, deterministic=False, seed=0):
        self.engine, self.deterministic = engine_with_net, deterministic
        self.rng = np.random.default_rng(seed)

    def moves(self, states, status):
        _, _, probs, _ = self.engine.net_forward(states)
        A = probs.shape[1]
        legal = ((status["legal"][:, None] >> np.arange(A)[None, :]) & 1).astype(bool)
        p = np.where(legal, probs, np.float32(0)).astype(np.float32)
        s = p.sum(1, keepdims=True, dtype=np.float32)
        p = p / np.where(s > 0, s, 1)
        if self.deterministic:
            return p.argmax(1).astype(np.int32)
        u = self.rng.random(len(p))[:, None]
        return np.minimum((np.cumsum(p, axis=1) < u).sum(1), A - 1).astype(np.int32)


def play_sides(game, n_games, words, side_x, side_o, start_states=None, max_plies=None):
    """side_x moves first (X, turn +1).  Returns (winners int8[G] in {+1, -1, 0}, plies int32[G])."""
    states = np.zeros((n_games, words), dtype=np.uint64) if start_states is None else np.array(start_states, np.uint64)
    st = env_status_batch(game, states)
    plies = np.zeros(n_games, dtype=np.int32)
    ply = 0
    while not st["done"].all() and (max_plies is None or ply < max_plies):
        side = side_x if ply % 2 == 0 else side_o
        live = st["done"] == 0
        actions = side.moves(states, st)
        legal = (st["legal"] >> np.clip(actions, 0, 15)) & 1
        if not ((legal == 1) | ~live).all():
            raise RuntimeError("arena: a side returned an illegal move")
        actions = np.where(live, actions, 0).astype(np.int32)   # finished games: dummy move, result discarded below
        out = env_step_batch(game, states, actions)
        states = np.where(live[:, None], out["states"], states)
        plies += live
        st = env_status_batch(game, states)
        ply += 1
    return st["winner"].astype(np.int8), plies


def play_arena(engine_x, engine_o, max_plies=None):
    """Pipeline.update_elo's pairing for a batch: two searching engines (same game, same n_games)."""
    assert engine_x.game == engine_o.game and engine_x.G == engine_o.G
    engine_x.reset_games()
    start = engine_x.get_roots()[0]
    return play_sides(engine_x.game, engine_x.G, engine_x.W, SearchSide(engine_x), SearchSide(engine_o), start, max_plies)


def win_rate(winners_as_x, winners_as_o):
    """select_best_player's score of the candidate (Pipeline.py:145-158): 1 per win, 0.5 per draw, over all games."""
    n = len(winners_as_x) + len(winners_as_o)
    w = (winners_as_x == 1).sum() + (winners_as_o == -1).sum()
    d = (winners_as_x == 0).sum() + (winners_as_o == 0).sum()
    return float(w + 0.5 * d) / n
