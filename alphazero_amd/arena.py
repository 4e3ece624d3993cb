"""Batched evaluation arena: G games between two searching players in lockstep on the device.

Reference: Pipeline.update_elo (Pipeline.py:125-135) plays the AlphaZero player against `MCTSPlayer(1, 1000)` one game
at a time through Game.start_play (game.py:12-33): per move `player.get_action(env)` (temp 0), `player.reset_player()`
(fresh tree for every move), `env.step(action)`.  Here every game of the batch is at the same ply, so the player to move
searches all of them at once: its engine is re-rooted at the current positions (fresh trees), runs its playouts,
chooses, and the K1 kernel steps the positions.
"""
import numpy as np

from .engine import env_status_batch, env_step_batch


def play_arena(engine_x, engine_o, max_plies=None):
    """engine_x moves first (X, turn +1), engine_o second; both must have the same game and n_games and be created with
    move="sample" (uniform among the most visited moves at temp 0, like player.py:106-111) or "first_max".
    Returns (winners int8[G] in {+1, -1, 0}, plies int32[G])."""
    assert engine_x.game == engine_o.game and engine_x.G == engine_o.G
    G, W = engine_x.G, engine_x.W
    game = engine_x.game
    states = np.zeros((G, W), dtype=np.uint64)
    engine_x.reset_games()
    states[:] = engine_x.get_roots()[0]
    st = env_status_batch(game, states)
    plies = np.zeros(G, dtype=np.int32)
    ply = 0
    while not st["done"].all() and (max_plies is None or ply < max_plies):
        eng = engine_x if ply % 2 == 0 else engine_o
        eng.set_roots(states)                 # fresh trees at the current positions (reset_player, game.py:21)
        eng.search()
        actions, _ = eng.choose_moves(temp=0.0)
        live = st["done"] == 0
        actions = np.where(live, actions, 0).astype(np.int32)   # finished games: dummy move, result discarded below
        out = env_step_batch(game, states, actions)
        states = np.where(live[:, None], out["states"], states)
        plies += live
        st = env_status_batch(game, states)
        ply += 1
    return st["winner"].astype(np.int8), plies
