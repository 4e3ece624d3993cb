"""Game loop with the reference's interface (game.py:8-58) plus its batched sibling.

start_play / start_self_play drive ONE game through player.get_action exactly like the reference (so play.py and
Pipeline.data_collector run unchanged).  start_self_play_batch plays many games in lockstep on the device
(az_selfplay) and yields the same 6-tuples per game."""
import numpy as np


def _seed(rng_seed):
    """device RNG seed of a batched self-play call: the caller's, else one draw from the global np.random state"""
    return int(np.random.randint(1 << 31)) if rng_seed is None else int(rng_seed)


class Game:
    def __init__(self, env):
        self.env = env

    def start_play(self, player1, player2, show=1):
        self.env.reset()
        if show:
            self.env.show()
        while True:
            player = player1 if self.env.turn == 1 else player2
            action, probs, *_ = player.get_action(self.env)
            player.reset_player()
            self.env.step(action)
            if show:
                self.env.show()
            if self.env.done():
                winner = self.env.winPlayer()
                if show:
                    print("Game end. Winner is %s" % ("X" if winner == 1 else "O") if winner != 0 else "Game end. Draw")
                return winner

    def start_self_play(self, player, temp=1, first_n_steps=5):
        self.env.reset()
        states, mcts_probs, current_players, next_states = [], [], [], []
        steps = 0
        while True:
            temperature = 1e-3 if steps >= first_n_steps else temp
            action, probs = player.get_action(self.env, temperature)
            steps += 1
            states.append(self.env.current_state())
            mcts_probs.append(probs)
            current_players.append(self.env.turn)
            self.env.step(action)
            next_states.append(self.env.current_state())
            if self.env.done():
                winner = self.env.winPlayer()
                cp = np.array(current_players)
                winner_z = np.zeros(len(cp))
                if winner != 0:
                    winner_z[cp == winner] = 1
                    winner_z[cp != winner] = -1
                dones = [False] * len(cp)
                dones[-1] = True
                return winner, zip(states, mcts_probs, winner_z, winner_z, next_states, dones)

    def collect_self_play(self, player, buffer, n_games, temp=1, first_n_steps=5, n_slots=None, rng_seed=None):
        """Pipeline.data_collector (Pipeline.py:53-63) in one call: play n_games in lockstep on the device with player's
        search settings and write every ply into `buffer` (the drop-in ReplayBuffer, or the reference's, with its tensors on the GPU)
        with collector.store_many -- trajectories never visit the host.  Under torch.distributed every rank plays
        n_games and every rank's buffer receives the plies of ALL ranks (RCCL all-gather of packed records).
        Returns the list of episode lengths of the games stored (data_collector's episode_len is their mean).
        rng_seed=None draws the device RNG seed from np.random like every random choice of the reference
        (Pipeline.py:53-63 never seeds), so successive calls play different games; pass a number to replay one."""
        import torch
        from alphazero_amd.collector import TrajectoryGatherer, store_many
        from alphazero_amd.engine import SelfPlayEngine
        m = player.mcts
        if not hasattr(m.policy, "export_flat"):
            raise NotImplementedError("batched self-play needs the device evaluator (the drop-in CNN)")
        game = self.env.GAME
        dev_index = getattr(m.policy, "device_index", 0)
        ddp = torch.distributed.is_available() and torch.distributed.is_initialized()
        rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if ddp else (0, 1)
        G = int(n_slots or min(n_games, 2048))
        eng = SelfPlayEngine(game, n_games=G, n_playout=m.n_playout, c_puct=m.c_init, evaluator="cnn", flip="random",
                             reuse_tree=bool(player.is_selfplay), dirichlet_alpha=None if m.deterministic else m.alpha,
                             rng_seed=_seed(rng_seed), move="sample", device=dev_index, game_id_base=rank,
                             game_id_stride=world)
        try:
            eng.load_net(m.policy.export_flat())
            eng.selfplay_rolling(n_games, temp=float(temp), first_n_steps=int(first_n_steps))
            # Games enter the trajectory buffer in the order in which they finish, and which of two games finishes first is a
            # matter of timing (a tree step ends by the clock).  A caller that names a seed asks for a replay: its rows come
            # in game-id order, whatever the timing was (one rank; under torch.distributed every rank's block is its own).
            order = None
            if rng_seed is not None and world == 1:
                gid = eng.traj_fetch(clear=False)["game_id"]
                order = torch.from_numpy(np.argsort(gid, kind="stable").astype(np.int64)).to(torch.device("cuda", dev_index))
            recs = TrajectoryGatherer(eng, torch.device("cuda", dev_index)).gather()
            if order is not None and order.numel() == recs.shape[0]:
                recs = recs.index_select(0, order)
        finally:
            eng.close()
        if hasattr(buffer, "store_many"):    # the drop-in ReplayBuffer keeps its row counter itself
            buffer.store_many(recs, game)
        else:                                # the reference's own ReplayBuffer object
            store_many(buffer, recs, game)
        done_col = eng.W * 8 + eng.A * 4 + 3
        ends = torch.nonzero(recs[:, done_col]).flatten().cpu().numpy()
        return [int(x) for x in np.diff(np.concatenate([[-1], ends]))]

    def start_self_play_batch(self, player, n_games, temp=1, first_n_steps=5, n_slots=None, rng_seed=None):
        """n_games self-play games at once on the device (rolling rounds, include/azamd.h) with player's search settings (c_puct, n_playout, alpha,
        train/eval mode) and network.  Returns a list of (winner, [6-tuples]) in completion order, each 6-tuple as
        game.py:58 builds it: (state (1,C,H,W) f32, pi (A,) f32, z, z, next_state, done)."""
        from alphazero_amd.engine import SelfPlayEngine, env_planes_batch, env_step_batch
        m = player.mcts
        game = self.env.GAME
        on_device = hasattr(m.policy, "export_flat")
        if not on_device:
            raise NotImplementedError("batched self-play needs the device evaluator (the drop-in CNN)")
        G = int(n_slots or min(n_games, 2048))
        eng = SelfPlayEngine(game, n_games=G, n_playout=m.n_playout, c_puct=m.c_init, evaluator="cnn", flip="random",
                             reuse_tree=bool(player.is_selfplay), dirichlet_alpha=None if m.deterministic else m.alpha,
                             rng_seed=_seed(rng_seed), move="sample", device=getattr(m.policy, "device_index", 0))
        try:
            eng.load_net(m.policy.export_flat())
            eng.selfplay_rolling(n_games, temp=float(temp), first_n_steps=int(first_n_steps))
            t = eng.traj_fetch(clear=True)
            if rng_seed is not None:   # a replay: game-id order instead of completion order (which is a matter of timing)
                o = np.argsort(t["game_id"], kind="stable")
                t = {k: v[o] for k, v in t.items()}
        finally:
            eng.close()
        planes = env_planes_batch(game, t["states"])
        nxt = env_step_batch(game, t["states"], t["action"].astype(np.int32))
        next_planes = env_planes_batch(game, nxt["states"])
        out, lo = [], 0
        ends = np.flatnonzero(t["done"])
        for hi in ends:
            hi = int(hi) + 1
            mover_last = int(t["player"][hi - 1])
            z_last = float(t["z"][hi - 1])
            winner = 0 if z_last == 0 else (mover_last if z_last > 0 else -mover_last)
            tuples = [(planes[i:i + 1], t["pi"][i], float(t["z"][i]), float(t["z"][i]), next_planes[i:i + 1],
                       bool(t["done"][i])) for i in range(lo, hi)]
            out.append((winner, tuples))
            lo = hi
        return out
