// az_play_dev.hpp -- K4 per-game device functions: move selection, advancing a game, trajectory recording, refill.
// One lane works on one game.  Called by the K4 kernels (az_play.hip, once per lockstep ply) and, in rolling self-play,
// by the tree step itself at the playout that completes a move (az_tree.hip).
#pragma once
#include "az_internal.hpp"
#include "az_play.hpp"

namespace az {

// numpy's float64 add-reduce order for a short contiguous array (n < 8: plain loop; else 8 partial sums).
__device__ inline double np_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double res = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    for (int i = 8; i < n; i++) res += a[i];
    return res;
}

__device__ inline double rng_uniform53(u64 seed, u64 gid, u64 ctr, u32 stream) {
    u64 h = mix64(seed ^ mix64(gid * 0x9E3779B97F4A7C15ull + ctr) ^ ((u64)stream << 32));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// AlphaZeroPlayer.get_action after the search (player.py:104-111) for game g (active): pi[A] (zero-filled here) =
// softmax(ln N / max(T,1e-8)) as float32, returns the move (first-max or sampled) or -1 when the root has no children.
template <class G>
__device__ inline int choose_game(const Dev& d, int g, double T, int move_mode, float* pi, i64 pi_stride) {
    const uint4* nodes = d.nodes + (i64)g * d.cap;
    uint4 rr = nodes[d.root[g]];
    const int k = (int)link_nchild(rr.w);
    const u32 first = link_first(rr.w);
    if (k == 0) {
        for (int a = 0; a < G::A; a++) pi[a * pi_stride] = 0.f;
        return -1;
    }
    // every loop runs over the static action count with an `i < k` predicate, so that the per-child arrays live in
    // registers (this code also runs inside the tree step of rolling self-play)
    int acts[G::A], vis[G::A];
    double x[G::A], ex[G::A];
    const double t = T > 1e-8 ? T : 1e-8;
    double mx = -INFINITY;
    int maxv = 0;
    uint4 c[G::A];
#pragma unroll
    for (int i = 0; i < G::A; i++) c[i] = nodes[first + (i < k ? i : 0)];
#pragma unroll
    for (int i = 0; i < G::A; i++) {
        acts[i] = (int)link_action(c[i].w);
        vis[i] = (int)c[i].x;
        x[i] = 0.0;
        if (i < k) {
            if (vis[i] > maxv) maxv = vis[i];
            x[i] = log((double)vis[i]) / t;  // np.log(visits) / max(temp, 1e-8)
            if (x[i] > mx) mx = x[i];
        }
    }
#pragma unroll
    for (int i = 0; i < G::A; i++) ex[i] = i < k ? exp(x[i] - mx) : 0.0;
    double s;   // numpy's float64 add-reduce order (np_sum)
    if (G::A < 8 || k < 8) {
        s = 0.0;
#pragma unroll
        for (int i = 0; i < G::A; i++)
            if (i < k) s += ex[i];
    } else {
        s = ((ex[0] + ex[1]) + (ex[2] + ex[3])) + ((ex[4] + ex[5 < G::A ? 5 : 0]) + (ex[6 < G::A ? 6 : 0] + ex[7 < G::A ? 7 : 0]));
#pragma unroll
        for (int i = 8; i < G::A; i++)
            if (i < k) s += ex[i];
    }
    // action_probs[list(actions)] = visit_dist
#pragma unroll
    for (int a = 0; a < G::A; a++) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < G::A; i++)
            if (i < k && acts[i] == a) v = (float)(ex[i] / s);
        pi[a * pi_stride] = v;
    }
    int best = 0;
    if (move_mode == AZ_MOVE_FIRST_MAX) {
        if (T == 0) {
            best = -1;
#pragma unroll
            for (int i = 0; i < G::A; i++)
                if (best < 0 && i < k && vis[i] == maxv) best = i;
        } else {
            double bd = -1.0;
#pragma unroll
            for (int i = 0; i < G::A; i++) {
                double dd = ex[i] / s;
                if (i < k && dd > bd) { bd = dd; best = i; }
            }
        }
    } else {
        double u = rng_uniform53(d.rng_seed, d.game_id[g], (u64)d.ply[g], 7u);
        if (T == 0) {  // player.py:106-108: uniform over the arg-max set
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < G::A; i++) cnt += (i < k && vis[i] == maxv);
            int pick = (int)(u * cnt);
            if (pick >= cnt) pick = cnt - 1;
            best = -1;
#pragma unroll
            for (int i = 0; i < G::A; i++)
                if (best < 0 && i < k && vis[i] == maxv) {
                    if (pick == 0) best = i;
                    pick--;
                }
        } else {  // np.random.choice(actions, p=visit_dist): inverse CDF
            double acc = 0.0;
            best = -1;
#pragma unroll
            for (int i = 0; i < G::A; i++)
                if (best < 0 && i < k) {
                    acc += ex[i] / s;
                    if (u < acc) best = i;
                }
            if (best < 0) best = k - 1;
        }
    }
    int act = acts[0];
#pragma unroll
    for (int i = 1; i < G::A; i++)
        if (i == best) act = acts[i];
    return act;
}

template <class G>
__device__ inline typename G::State opening_state(const Dev& d, u64 gid) {
    typename G::State s = G::initial();
    if constexpr (G::GAME == AZ_GAME_NBTTT) {  // NBTTT/env.py:15-23: two random opening plies
        u64 h = mix64(d.rng_seed ^ mix64(gid + 0x5851F42D4C957F2Dull));
        int board = (int)(h % 9), a1 = (int)((h >> 16) % 9);
        s.w[3] = (u64)board;
        G::step(s, a1);
        u32 legal = G::legal(s) & ~(1u << board);
        int m = __popc(legal), pick = (int)((h >> 32) % (u64)m);
        for (int a = 0; a < 9; a++)
            if ((legal >> a) & 1u) {
                if (pick == 0) { G::step(s, a); break; }
                pick--;
            }
    }
    return s;
}

// ---- wave-aggregated reservations on a global counter.  Whole cohorts of games end on the same ply (4 096 TicTacToe games, all
// 5 to 9 plies long): with one compare-and-swap loop per lane those lanes serialise on one address (8.7 ms for a ply end that
// takes 26 us otherwise); here the lanes of a wave that arrive together add up their needs and ONE lane runs the loop.

// Every calling lane reserves n units of *ctr without ever moving it past `limit`.  Returns the lane's first unit, or ~0 when
// its n units do not fit (the counter is then untouched by this lane).
__device__ inline unsigned long long wave_reserve(unsigned long long* ctr, u32 n, unsigned long long limit) {
    const u64 m = __ballot(1);
    const int lane = (int)__lane_id(), leader = __ffsll((long long)m) - 1;
    unsigned long long pre = 0, tot = 0;
    for (u64 mm = m; mm; mm &= mm - 1) {
        const int src = __ffsll((long long)mm) - 1;
        const u32 v = (u32)__shfl((int)n, src);
        if (src < lane) pre += v;
        tot += v;
    }
    u32 lo = ~0u, hi = ~0u;
    if (lane == leader) {
        unsigned long long off = atomicAdd(ctr, 0ull);
        while (off + tot <= limit) {
            const unsigned long long prev = atomicCAS(ctr, off, off + tot);
            if (prev == off) {
                lo = (u32)off;
                hi = (u32)(off >> 32);
                break;
            }
            off = prev;
        }
    }
    lo = (u32)__shfl((int)lo, leader);
    hi = (u32)__shfl((int)hi, leader);
    const unsigned long long base = ((unsigned long long)hi << 32) | lo;
    if (base != ~0ull) return base + pre;
    // the wave's units do not all fit any more: lane by lane, so that the ones that still fit are not lost
    unsigned long long off = atomicAdd(ctr, 0ull);
    while (off + n <= limit) {
        const unsigned long long prev = atomicCAS(ctr, off, off + (unsigned long long)n);
        if (prev == off) return off;
        off = prev;
    }
    return ~0ull;
}

// Every calling lane takes one index below `target` from *ctr, in lane order; ~0 for the lanes that find the target reached.
__device__ inline unsigned long long wave_take_one(unsigned long long* ctr, unsigned long long target) {
    const u64 m = __ballot(1);
    const int lane = (int)__lane_id(), leader = __ffsll((long long)m) - 1;
    const u32 cnt = (u32)__popcll(m), rank = (u32)__popcll(m & ((1ull << lane) - 1ull));
    u32 lo = 0, hi = 0, grant = 0;
    if (lane == leader) {
        unsigned long long idx = atomicAdd(ctr, 0ull);
        while (idx < target) {
            const unsigned long long g = target - idx < cnt ? target - idx : (unsigned long long)cnt;
            const unsigned long long prev = atomicCAS(ctr, idx, idx + g);
            if (prev == idx) {
                grant = (u32)g;
                lo = (u32)idx;
                hi = (u32)(idx >> 32);
                break;
            }
            idx = prev;
        }
    }
    lo = (u32)__shfl((int)lo, leader);
    hi = (u32)__shfl((int)hi, leader);
    grant = (u32)__shfl((int)grant, leader);
    if (rank >= grant) return ~0ull;
    return (((unsigned long long)hi << 32) | lo) + rank;
}

// env.step(a) + prune_root(a)/reset_player() + trajectory bookkeeping for the active game g (a >= 0).
template <class G>
__device__ inline void advance_game(const Dev& d, const Play& p, int g, int a, const float* pi, i64 pi_stride, int reuse,
                                    int record) {
    typename G::State s;
    for (int i = 0; i < G::W; i++) s.w[i] = d.root_state[(i64)g * G::W + i];
    int ply = d.ply[g];
    if (record && ply < G::MAX_PLIES) {  // game.py:45-47: state, probs, current player BEFORE the move
        i64 o = (i64)g * G::MAX_PLIES + ply;
        for (int i = 0; i < G::W; i++) p.tj_state[o * G::W + i] = s.w[i];
        for (int i = 0; i < G::A; i++) p.tj_pi[o * G::A + i] = pi[i * pi_stride];
        p.tj_action[o] = (i8)a;
        p.tj_player[o] = (i8)G::turn(s);
    }
    G::step(s, a);
    for (int i = 0; i < G::W; i++) d.root_state[(i64)g * G::W + i] = s.w[i];
    ply++;
    d.ply[g] = ply;
    uint4* nodes = d.nodes + (i64)g * d.cap;
    bool kept = false;
    if (reuse) {  // MCTS.py:131-133
        uint4 rr = nodes[d.root[g]];
        u32 k = link_nchild(rr.w), first = link_first(rr.w);
        for (u32 i = 0; i < k; i++)
            if ((int)link_action(nodes[first + i].w) == a) {
                d.root[g] = first + i;
                kept = true;
                break;
            }
    }
    if (!kept) {  // MCTS.py:134-135
        nodes[0] = make_uint4(0u, 0u, __float_as_uint(1.0f), 0u);
        d.root[g] = 0u;
        d.n_alloc[g] = 1u;
    }
    int w = G::winner(s);
    bool dn = (w != 0) || G::full(s);
    d.winner[g] = (i8)w;
    d.done[g] = dn ? 1 : 0;
    atomicAdd(&d.stats[ST_PLIES], 1ull);
    if (dn) {
        d.active[g] = 0;
        atomicAdd(&d.stats[ST_GAMES], 1ull);
        if (record) {  // game.py:50-58: commit the finished game with z from each mover's perspective
            int n = ply < G::MAX_PLIES ? ply : G::MAX_PLIES;
            // reserve n rows below out_cap: out_count only ever covers rows that get written, so a game that does not fit is
            // dropped without leaving a hole and later, shorter games may still fit
            const unsigned long long off = wave_reserve(p.out_count, (u32)n, (unsigned long long)p.out_cap);
            const bool fits = off != ~0ull;
            if (fits) {
                for (int j = 0; j < n; j++) {
                    i64 o = (i64)g * G::MAX_PLIES + j, q = (i64)off + j;
                    for (int i = 0; i < G::W; i++) p.out_state[q * G::W + i] = p.tj_state[o * G::W + i];
                    for (int i = 0; i < G::A; i++) p.out_pi[q * G::A + i] = p.tj_pi[o * G::A + i];
                    int pl = p.tj_player[o];
                    p.out_z[q] = (w == 0) ? 0.f : ((pl == w) ? 1.f : -1.f);
                    p.out_player[q] = (i8)pl;
                    p.out_action[q] = p.tj_action[o];
                    p.out_done[q] = (j == n - 1) ? 1 : 0;
                    p.out_gid[q] = d.game_id[g];
                }
                atomicAdd(p.out_games, 1ull);
            } else {
                atomicOr(d.err, 4u);  // trajectory buffer full: game dropped
            }
        }
    }
}

// Restart the finished slot g with a fresh game while fewer than `target` games have been started.
template <class G>
__device__ inline void refill_game(const Dev& d, const Play& p, int g, unsigned long long target, u64 gid_base) {
    if (d.active[g] || !d.done[g]) return;
    // take the next game index while it is below the target (the counter never runs past the target, so the target may be
    // raised between launches and exactly `target` games are ever started)
    const unsigned long long idx = wave_take_one(p.games_started, target);
    if (idx == ~0ull) return;
    u64 gid = gid_base + d.gid_stride * idx;
    typename G::State s = opening_state<G>(d, gid);
    for (int i = 0; i < G::W; i++) d.root_state[(i64)g * G::W + i] = s.w[i];
    d.nodes[(i64)g * d.cap] = make_uint4(0u, 0u, __float_as_uint(1.0f), 0u);
    d.root[g] = 0u;
    d.n_alloc[g] = 1u;
    d.ply[g] = 0;
    d.sim_ctr[g] = 0;
    d.move_sims[g] = 0;
    d.game_id[g] = gid;
    d.winner[g] = 0;
    d.done[g] = 0;
    d.active[g] = 1;
    d.leaf_flags[g] = 0u;
}

}  // namespace az
