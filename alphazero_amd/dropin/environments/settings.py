"""Hyper-parameters of the three games, as the engine consumes them.  Values are the reference's
(environments/Connect4/config.py:1-21, TTT/config.py:1-22, NBTTT/config.py:1-24); only the ones on the self-play
path matter here, the rest are carried so that drivers reading them keep working."""

_COMMON = dict(init_elo=1500, num_eval=50, pure_mcts_n_playout=1000, temp=1.0)

GAMES = {
    "Connect4": dict(
        training=dict(_COMMON, lr=1e-3, n_playout=80, first_n_steps=10, c_puct=1.25, buffer_size=200000,
                      batch_size=256, dirichlet_alpha=0.7, win_rate_threshold=0.65),
        env=dict(row=6, col=7), network=dict(in_dim=3, h_dim=128, out_dim=7)),
    "TTT": dict(
        training=dict(_COMMON, lr=3e-3, n_playout=100, first_n_steps=1, c_puct=1.0, buffer_size=10000, batch_size=64,
                      play_batch_size=1, dirichlet_alpha=0.1, win_rate_threshold=0.65),
        env=dict(row=3, col=3), network=dict(in_dim=3, h_dim=128, out_dim=9)),
    "NBTTT": dict(
        training=dict(_COMMON, lr=1e-3, n_playout=100, first_n_steps=12, c_puct=1.5, buffer_size=100000,
                      batch_size=128, discount=0.98, play_batch_size=1, epochs=1, dirichlet_alpha=0.1,
                      win_rate_threshold=0.6),
        env=dict(row=3, col=3), network=dict(in_dim=28, h_dim=128, out_dim=9)),
}


def configs(name):
    g = GAMES[name]
    return dict(g["training"]), dict(g["network"]), dict(g["env"])
