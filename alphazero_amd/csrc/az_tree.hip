// az_tree.hip -- K2: batched PUCT select / expand / backup over per-game node arenas in HBM (gfx950).
//
// Replaces the reference's Python tree for G games at once:
//   TreeNode            MCTS.py:10-19     -> 16-byte node record (az_internal.hpp), children contiguous in insertion order
//   TreeNode.PUCT       MCTS.py:47-55     -> puct_score(): the exact float32 rounding sequence, no FMA contraction
//   TreeNode.select     MCTS.py:79-80     -> one lane per child, sub-wave butterfly arg-max, ties -> lowest insertion index
//   select_leaf_node    MCTS.py:108-113   -> walk(): one coalesced 16-byte load per lane per level, env.step on bitboards
//   MCTS_AZ.playout     MCTS.py:139-164   -> leaf_prepare() + finish(): flip, mask + renormalise priors, expand, terminal value
//   TreeNode.expand     MCTS.py:71-77     -> bump allocation of k child records, one lane per child
//   TreeNode.update     MCTS.py:82-87     -> backup from the path staged in LDS: float32 running mean, alternating sign
//
// Work decomposition: one sub-wave GROUP of LG lanes (8 for Connect4, 16 for TTT/NBTTT) owns one game; a 256-thread
// block holds 256/LG games.  Groups never talk to each other, a game is only ever touched by its own group, and a
// group lives inside one wavefront, so all tree reads/writes of a game are wavefront-ordered.
//
// This file must be compiled with -ffp-contract=off and correctly rounded f32 divide/sqrt (see Makefile).
#include <stdlib.h>
#include <string.h>

#include "az_internal.hpp"
#include "az_play_dev.hpp"

namespace az {

__device__ inline float as_f(u32 x) { return __uint_as_float(x); }
__device__ inline u32 as_u(float x) { return __float_as_uint(x); }

// MCTS.py:47-55.  tab = {f32(c_init + ln((1+Np+c_base)/c_base)), f32(sqrt(Np))}, built on the host in float64.
__device__ inline float puct_score(u32 n, float Q, float P, float2 tab) {
    float pr = 0.75f * P;   // (1 - eps) * prior
    float nz = 0.25f * P;   // eps * noise   (noise == prior in eval mode; train mode stores the mix in P)
    float p = pr + nz;
    if (n == 0) return INFINITY;  // first-play urgency: u = inf  (MCTS.py:50-51)
    float t = tab.x * p;
    t = t * tab.y;
    float u = t / (float)(1u + n);
    return -Q + u;
}

__device__ inline float2 tab_lookup(const Dev& d, u32 Np) {
    if (Np < (u32)d.tab_n) return d.tab[Np];
    atomicOr(d.err, 2u);  // table too short: fall back to device libm (not covered by the bit-exact contract)
    double A = d.c_init + log((double)(1u + Np + (u32)d.c_base) / (double)d.c_base);
    return make_float2((float)A, (float)sqrt((double)Np));
}

template <class G>
__device__ inline typename G::State load_root_state(const Dev& d, int g) {
    typename G::State s;
#pragma unroll
    for (int i = 0; i < G::W; i++) s.w[i] = d.root_state[(i64)g * G::W + i];
    return s;
}

// Synthetic evaluators standing in for CNN.predict (Network.py:78-85) in tree-only runs.
//   UNIFORM: probs = f32(1/A) for every action, value 0 (BASELINE config 2).
//   HASH   : h = fold of splitmix64 over the packed words of the EVALUATED position; raw_a = ((mix(h+a+1) >> 40) + 1) / 2^24,
//            probs = raw / left-to-right f32 sum(raw), value = (mix(h+64) >> 40) / 2^23 - 1.  Specified in DESIGN.md so that any
//            implementation (the reference through a fake policy_value_fn, a CPU checker, this kernel) computes identical bits.
template <class G>
__device__ inline void eval_synthetic(int kind, u64 seed, const typename G::State& aug, float* probs, float& value, int gl) {
    if (kind == AZ_EVAL_UNIFORM) {
        const float p = 1.0f / (float)G::A;
#pragma unroll
        for (int a = 0; a < G::A; a++) probs[a] = p;
        value = 0.f;
    } else {
        u64 h = seed;
#pragma unroll
        for (int i = 0; i < G::W; i++) h = mix64(h ^ aug.w[i]);
        // lane a of the group hashes action a (every lane used to hash all A of them); the raw values are then handed round
        // so that every lane adds them up in the same left-to-right order as before
        const float mine = (float)((mix64(h + (u64)gl + 1) >> 40) + 1) * (1.0f / 16777216.0f);
        float raw[G::A];
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < G::A; a++) {
            raw[a] = __shfl(mine, a, G::LG);
            s = s + raw[a];
        }
#pragma unroll
        for (int a = 0; a < G::A; a++) probs[a] = raw[a] / s;
        value = (float)(mix64(h + 64) >> 40) * (1.0f / 8388608.0f) - 1.0f;
    }
}

// Counter-based uniform in [0,1) keyed by (seed, game id, playout counter, stream).
__device__ inline float rng_uniform(u64 seed, u64 gid, u64 ctr, u32 stream) {
    u64 h = mix64(seed ^ mix64(gid * 0x9E3779B97F4A7C15ull + ctr) ^ ((u64)stream << 32));
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}

// Gamma(alpha, 1) sample, Marsaglia-Tsang with the alpha<1 boost; statistical parity only (np.random.dirichlet,
// MCTS.py:156).  Each child lane draws its own variate from its own counter stream.  The sampler runs at EVERY expansion of a
// train-mode search (MCTS.py:155-157), i.e. once per non-terminal playout, inside the tree step's dependent chain: its
// logarithms, cosine, square root and power are the hardware's approximations (v_log_f32, v_cos_f32, v_sqrt_f32, v_exp_f32:
// ~1 ulp-class errors that a statistical contract does not see) instead of libm's correctly rounded routines, which were
// several hundred instructions per expansion.
__device__ inline float fast_ln(float x) { return __builtin_amdgcn_logf(x) * 0.69314718f; }   // v_log_f32 is log2
__device__ inline float rng_gamma(float alpha, u64 seed, u64 gid, u64 ctr, u32 lane) {
    float a = alpha < 1.f ? alpha + 1.f : alpha;
    float dd = a - 1.0f / 3.0f, c = __builtin_amdgcn_rsqf(9.0f * dd), x = dd;
    for (u32 it = 0; it < 32; it++) {
        float u1 = rng_uniform(seed, gid, ctr, 0x100u + lane * 128u + it * 3u);
        float u2 = rng_uniform(seed, gid, ctr, 0x101u + lane * 128u + it * 3u);
        float u3 = rng_uniform(seed, gid, ctr, 0x102u + lane * 128u + it * 3u);
        // Box-Muller; v_cos_f32 takes its argument in revolutions: cos(2 pi u2)
        float n = __builtin_amdgcn_sqrtf(-2.f * fast_ln(u1 + 5.9604645e-8f)) * __builtin_amdgcn_cosf(u2);
        float v = 1.f + c * n;
        if (v <= 0.f) continue;
        v = v * v * v;
        if (fast_ln(u3 + 5.9604645e-8f) < 0.5f * n * n + dd - dd * v + dd * fast_ln(v)) {
            x = dd * v;
            break;
        }
    }
    if (alpha < 1.f) {
        float u = rng_uniform(seed, gid, ctr, 0x1000u + lane);
        x *= __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(u + 5.9604645e-8f) * __builtin_amdgcn_rcpf(alpha));   // u^(1/alpha)
    }
    return x;
}

// Lane exchange inside a group through DPP (data-parallel primitives: a VALU operand modifier, ~8 cycles) instead of
// ds_bpermute (an LDS round trip, ~120 cycles on the walk's dependent chain).  STEP 0: lane ^ 1, 1: lane ^ 2 (quad
// permutes), 2: mirror inside each half-row of 8 lanes, 3: mirror inside the row of 16.  After steps 0..log2(LG)-1 every
// lane has met, directly or through a partner's running result, every other lane of its group of LG = 8 or 16 lanes.
template <int STEP>
__device__ inline u32 dpp_xchg(u32 v) {
    constexpr int ctrl = STEP == 0 ? 0xB1 : STEP == 1 ? 0x4E : STEP == 2 ? 0x141 : 0x140;
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, true);
}
template <int STEP>
__device__ inline void argmax_step(float& score, int& idx) {
    const float os = as_f(dpp_xchg<STEP>(as_u(score)));
    const int oi = (int)dpp_xchg<STEP>((u32)idx);
    // `|` and `&`, not `||` and `&&`: three compares and two selects, no branches on the exec mask inside the walk's chain
    const bool take = (os > score) | ((os == score) & (oi < idx));
    score = take ? os : score;
    idx = take ? oi : idx;
}
// first maximum over the group (ties -> lowest lane = insertion order, MCTS.py:79-80); every lane gets the result
template <int LG>
__device__ inline int group_argmax_first(float score, int gl) {
    static_assert(LG == 8 || LG == 16, "lane groups of 8 or 16");
    int idx = gl;
    argmax_step<0>(score, idx);
    argmax_step<1>(score, idx);
    argmax_step<2>(score, idx);
    if constexpr (LG == 16) argmax_step<3>(score, idx);
    return idx;
}

// the value held by the ONE lane of the group for which `mine` is set, in every lane of the group: three (four) DPP steps of
// an OR butterfly instead of a ds_bpermute round trip through the LDS
template <int LG>
__device__ inline u32 group_pick(u32 v, bool mine) {
    v = mine ? v : 0u;
    v |= dpp_xchg<0>(v);
    v |= dpp_xchg<1>(v);
    v |= dpp_xchg<2>(v);
    if constexpr (LG == 16) v |= dpp_xchg<3>(v);
    return v;
}

template <int LG>
__device__ inline float group_sum(float v) {
#pragma unroll
    for (int m = LG / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, LG);
    return v;
}

struct Leaf {
    u32 node;      // arena index of the leaf
    int depth;     // levels walked (path has depth+1 entries)
    u32 link;      // the leaf's link word (no children yet; action nibble set)
    u32 flags;     // bit0 terminal, bit1 flipped, bit2 unflipped board mirror-symmetric
    float tvalue;  // terminal value
    int hint;      // G::move_hint of the position before the last step of the walk (meaningless at depth 0)
};

// MCTS.py:108-113 (select_leaf_node) for one game.  Stages (index, N, Q) of every node on the path in LDS.
// What the last level of a walk looked at, for the speculation of a parked leaf's siblings (below): the position the last move
// was made from, that node's child count, the lane that won the arg-max, and this lane's own child record.
template <class G>
struct WalkTail {
    typename G::State before;
    u32 k, n, link;
    int best;
};
template <class G, bool TAIL = false>
__device__ inline Leaf walk(const Dev& d, const uint4* nodes, typename G::State& s, u32 root, u32 rN, float rQ,
                            u32 rlink, int gl, u32* p_idx, u32* p_N, float* p_Q, WalkTail<G>* tail = nullptr) {
    constexpr int LG = G::LG;
    u32 node = root, Np = rN, link = rlink;
    int depth = 0, hint = 0;
    if (gl == 0) {
        p_idx[0] = root;
        p_N[0] = rN;
        p_Q[0] = rQ;
    }
    while (link_nchild(link) > 0 && depth < G::MAXD - 1) {
        u32 k = link_nchild(link), first = link_first(link);
        float2 tb = tab_lookup(d, Np);
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        if ((u32)gl < k) rec = nodes[first + gl];  // one coalesced 16-byte load per child lane
        // all four words are needed HERE: left to itself the compiler fetches the prior only once it has seen n != 0 (PUCT
        // returns +inf before it looks at the prior), i.e. a second memory round trip on every level of the walk
        asm volatile("" : "+v"(rec.x), "+v"(rec.y), "+v"(rec.z), "+v"(rec.w));
        const float score = (u32)gl < k ? puct_score(rec.x, as_f(rec.y), as_f(rec.z), tb) : -INFINITY;
        const int best = group_argmax_first<LG>(score, gl);
        const bool mine = gl == best;
        const u32 cN = group_pick<LG>(rec.x, mine), cQ = group_pick<LG>(rec.y, mine), cl = group_pick<LG>(rec.w, mine);
        node = first + (u32)best;
        depth++;
        if (gl == 0) {
            p_idx[depth] = node;
            p_N[depth] = cN;
            p_Q[depth] = as_f(cQ);
        }
        hint = G::move_hint(s);
        if constexpr (TAIL) {
            tail->before = s;
            tail->k = k;
            tail->n = rec.x;
            tail->link = rec.w;
            tail->best = best;
        }
        G::step(s, (int)link_action(cl));  // env.step(action), redundantly on every lane of the group
        Np = cN;
        link = cl;
    }
    Leaf lf;
    lf.hint = hint;
    lf.node = node;
    lf.link = link;
    lf.depth = depth;
    lf.flags = 0;
    lf.tvalue = 0.f;
    return lf;
}

// MCTS.py:142 + :154-163 -- draw the flip bit (always, like np.random.rand()), classify the leaf, build the
// position the evaluator sees.
template <class G>
__device__ inline void leaf_prepare(const Dev& d, int g, const typename G::State& s, typename G::State& aug, Leaf& lf,
                                    u32& fpos, u64 ctr) {
    bool flipped = false;
    if constexpr (G::CAN_FLIP) {
        if (d.flip_mode == AZ_FLIP_STREAM) {
            flipped = (fpos < (u64)d.flip_n) ? (d.flip_bits[(i64)g * d.flip_n + fpos] != 0) : false;
            fpos++;
        } else if (d.flip_mode == AZ_FLIP_RANDOM) {
            flipped = rng_uniform(d.rng_seed, d.game_id[g], ctr, 1u) < 0.5f;
        }
    }
    // The root of a live game holds no line, and a walk only ever steps from non-terminal nodes: a line at the leaf was made
    // by the last move, in the sub-board it went into (G::winner_moved; the full scan G::winner stays with the K1 API).
    int w = lf.depth > 0 ? G::winner_moved(s, lf.hint) : 0;
    bool term = (w != 0) || G::full(s);
    aug = flipped ? G::flip(s) : s;
    lf.flags = (term ? 1u : 0u) | (flipped ? 2u : 0u) | (G::symmetric(s) ? 4u : 0u);
    lf.tvalue = (w == 0) ? 0.f : ((w == G::turn(s)) ? 1.f : -1.f);  // MCTS.py:159-163
}

// MCTS.py:147-157 (mask, renormalise, remap, expand) + :164 (update).  probs[] are over ALL actions of the
// evaluated (possibly mirrored) position; returns the updated root (N, Q, link) through references.
template <class G>
__device__ inline void finish(const Dev& d, int g, uint4* nodes, const typename G::State& aug, const Leaf& lf,
                              const float* probs, float value, int gl, u32& nalloc, u32 root, u32& rN, float& rQ,
                              u32& rlink, const u32* p_idx, const u32* p_N, const float* p_Q, u64 ctr, u32& st_children) {
    constexpr int LG = G::LG;
    const bool term = lf.flags & 1u, flipped = lf.flags & 2u, sym = lf.flags & 4u;
    float v = lf.tvalue;
    if (!term) {
        v = value;
        u32 legal = G::legal(aug);  // env_aug.valid_move()
        int nv = 0, myact = -1;
        float s = 0.f, myp = 0.f;
#pragma unroll
        for (int a = 0; a < G::A; a++) {
            if ((legal >> a) & 1u) {
                float pa = probs[a];
                s = s + pa;  // sum(probs): left-to-right float32 over the evaluated env's ascending legal actions
                if (nv == gl) {
                    myp = pa;
                    myact = a;
                }
                nv++;
            }
        }
        float prior = myp / s;  // probs /= sum(probs)
        if (d.noise_alpha > 0.f) {  // train(): Dirichlet at every expansion, mixed 0.75/0.25 (MCTS.py:48-49,155-157)
            float gsample = (gl < nv) ? rng_gamma(d.noise_alpha, d.rng_seed, d.game_id[g], ctr, (u32)gl) : 0.f;
            float tot = group_sum<LG>(gsample);
            float noise = tot > 0.f ? gsample / tot : 1.0f / (float)nv;
            prior = 0.75f * prior + 0.25f * noise;
        }
        if (flipped) myact = G::flip_action(sym, myact);  // MCTS.py:152-153
        u32 first = nalloc;
        if ((i64)first + nv <= d.cap && first + (u32)nv <= LINK_FIRST_MASK) {
            if (gl < nv) nodes[first + gl] = make_uint4(0u, 0u, as_u(prior), (u32)myact << 28);
            u32 nl = (lf.link & 0xF0000000u) | first | ((u32)nv << 24);  // keep the leaf's own action nibble
            if (gl == 0) reinterpret_cast<u32*>(&nodes[lf.node])[3] = nl;
            if (lf.node == root) rlink = nl;
            nalloc = first + (u32)nv;
            st_children += (u32)nv;
        } else {
            atomicOr(d.err, 1u);  // arena overflow: the leaf stays a leaf
        }
    }
    // TreeNode.update (MCTS.py:82-87): n += 1; Q += (v - Q) / n, sign alternating towards the root.
    const int depth = lf.depth;
    for (int dd = gl; dd <= depth; dd += LG) {
        u32 idx = p_idx[dd], N = p_N[dd];
        float Q = p_Q[dd];
        float val = ((depth - dd) & 1) ? -v : v;
        u32 N1 = N + 1u;
        float dq = val - Q;
        dq = dq / (float)N1;
        float Q1 = Q + dq;
        *reinterpret_cast<uint2*>(&nodes[idx]) = make_uint2(N1, as_u(Q1));
    }
    {  // the root's new statistics, kept in registers by every lane for the next playout
        float val = (depth & 1) ? -v : v;
        u32 N1 = rN + 1u;
        float dq = val - rQ;
        dq = dq / (float)N1;
        rQ = rQ + dq;
        rN = N1;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Evaluation cache (AZ_EVAL_CNN).  CNN.predict (Network.py:78-85) is a pure function of the evaluated (possibly mirrored)
// position, and K3's result for a position does not depend on where the position sits in a batch (tests/test_gpu_cnn.py:
// bit-identical), so a table keyed by the 16-byte position can answer a repeated leaf with exactly the bits K3 would produce:
// every game's trajectory is the same, bit for bit, with the cache on, off, hit or missed.  Self-play games from one network
// share most of their early and a good part of their later positions (measured with tools/leaf_duplicates.py on the bench's
// workload: 62 % of the leaves of 1 024 concurrent games had been evaluated before by some game, 14 % by the same game), so
// the table is shared by all the games of the engine.  The reference evaluates every leaf anew (MCTS.py:146).
//
// Direct-mapped, 64-byte entries, newest wins.  An entry is four 16-byte chunks of {3 data words, tag}:
//     data = w0.lo w0.hi w1.lo | w1.hi p0 p1 | p2 p3 p4 | p5 p6 value            (position words, CNN probs over all 7 actions, value)
//     tag_i = hi32(kh) ^ fold(chunk i's three data words) + i,   kh = mix64(w0 ^ mix64(w1 ^ epoch))
// Writers (any group of any wave, no locks, no atomics) store the four chunks; a reader accepts an entry only if every chunk
// carries the tag that ITS OWN data and the probed position give and the stored position equals the probed one.  A chunk torn
// between two writers, or chunks of two different writers in one entry, fail their tags (2^-32 per chunk to pass by chance,
// and the full position is compared on top); two writers of the same position write identical bytes.  `epoch` changes with
// every az_net_load: entries of the previous weights stop validating at once, nothing has to be cleared.
// ------------------------------------------------------------------------------------------------------------
__device__ inline u64 cache_hash(const Dev& d, u64 w0, u64 w1) { return mix64(w0 ^ mix64(w1 ^ d.cache_epoch)); }
__device__ inline u32 cache_tag(u64 kh, u32 a, u32 b, u32 c, u32 i) {
    const u32 f = (a * 0x9E3779B1u) ^ (((b << 11) | (b >> 21)) * 0x85EBCA77u) ^ (((c << 23) | (c >> 9)) * 0xC2B2AE3Du);
    return ((u32)(kh >> 32) ^ f) + i + 1u;
}
// probe for position (w0, w1): true and probs[7] / value filled on a hit.  Every lane of the group asks for the same 64 bytes.
__device__ inline bool cache_probe(const Dev& d, u64 w0, u64 w1, float* probs, float& value) {
    const u64 kh = cache_hash(d, w0, w1);
    const uint4* e = d.cache + (kh & d.cache_mask) * 4;
    const uint4 c0 = e[0], c1 = e[1], c2 = e[2], c3 = e[3];
    bool ok = c0.x == (u32)w0 && c0.y == (u32)(w0 >> 32) && c0.z == (u32)w1 && c1.x == (u32)(w1 >> 32);
    ok = ok && c0.w == cache_tag(kh, c0.x, c0.y, c0.z, 0u) && c1.w == cache_tag(kh, c1.x, c1.y, c1.z, 1u) &&
         c2.w == cache_tag(kh, c2.x, c2.y, c2.z, 2u) && c3.w == cache_tag(kh, c3.x, c3.y, c3.z, 3u);
    probs[0] = as_f(c1.y);
    probs[1] = as_f(c1.z);
    probs[2] = as_f(c2.x);
    probs[3] = as_f(c2.y);
    probs[4] = as_f(c2.z);
    probs[5] = as_f(c3.x);
    probs[6] = as_f(c3.y);
    value = as_f(c3.z);
    return ok;
}
// store K3's answer for position (w0, w1): lane i < 4 of the group writes chunk i
__device__ inline void cache_insert(const Dev& d, u64 w0, u64 w1, const float* probs, float value, int gl) {
    if (gl >= 4) return;
    const u64 kh = cache_hash(d, w0, w1);
    u32 a, b, c;
    if (gl == 0) a = (u32)w0, b = (u32)(w0 >> 32), c = (u32)w1;
    else if (gl == 1) a = (u32)(w1 >> 32), b = as_u(probs[0]), c = as_u(probs[1]);
    else if (gl == 2) a = as_u(probs[2]), b = as_u(probs[3]), c = as_u(probs[4]);
    else a = as_u(probs[5]), b = as_u(probs[6]), c = as_u(value);
    d.cache[(kh & d.cache_mask) * 4 + gl] = make_uint4(a, b, c, cache_tag(kh, a, b, c, (u32)gl));
}

// the same by ONE lane (speculative rows: every lane has a row of its own)
__device__ inline void cache_insert_lane(const Dev& d, u64 w0, u64 w1, const float* probs, float value) {
    const u64 kh = cache_hash(d, w0, w1);
    uint4* e = d.cache + (kh & d.cache_mask) * 4;
    const u32 x[12] = {(u32)w0, (u32)(w0 >> 32), (u32)w1, (u32)(w1 >> 32), as_u(probs[0]), as_u(probs[1]),
                       as_u(probs[2]), as_u(probs[3]), as_u(probs[4]), as_u(probs[5]), as_u(probs[6]), as_u(value)};
#pragma unroll
    for (u32 i = 0; i < 4; i++) e[i] = make_uint4(x[3 * i], x[3 * i + 1], x[3 * i + 2], cache_tag(kh, x[3 * i], x[3 * i + 1], x[3 * i + 2], i));
}

// ------------------------------------------------------------------------------------------------------------
// Speculation (az_search of a handful of games: the drop-in player's single game).  One game keeps one workgroup of K3 busy
// and 255 CUs idle, and a search's next new leaves are no secret: unvisited children are tried in insertion order before any
// revisit (first-play urgency, MCTS.py:50-51,80), so when a leaf has to wait for the network its unvisited SIBLINGS are for
// certain the next new leaves of its parent, and its own CHILDREN the next new leaves below it.  Their positions -- in both
// orientations, the flip bit of a future playout is not known -- ride along in the same K3 launch on rows of their own
// (Dev::spec_rows rows behind the slots' rows: kind x lane x orientation), and the next tree step stores the answers in the
// evaluation cache, where the later playouts find them.  Only the table is written: no game's tree, counters or random streams
// are touched, so the search is the same bit for bit with speculation on or off (K3's result for a position does not depend
// on the launch it was evaluated in).  Positions that are terminal or already in the table are skipped.
// ------------------------------------------------------------------------------------------------------------
template <class G>
__device__ inline void spec_flush(const Dev& d, int g, int gl) {   // K3 has answered the rows marked by the previous tree step
    if constexpr (G::GAME == AZ_GAME_C4) {
        for (int r = gl; r < d.spec_rows; r += G::LG) {
            const i64 m = (i64)g * d.spec_rows + r;
            const u8 mark = d.spec_mark[m];
            if (!mark) continue;
            if (mark == d.spec_tag) {   // (else: marked under weights that have been replaced since)
                const i64 v = (i64)d.G + m;
                float pr[G::A];
#pragma unroll
                for (int a = 0; a < G::A; a++) pr[a] = d.eval_probs[v * G::A + a];
                cache_insert_lane(d, d.leaf_state[v * G::W], d.leaf_state[v * G::W + 1], pr, d.eval_value[v]);
            }
            d.spec_mark[m] = 0;
        }
    }
}
template <class G>
__device__ inline void spec_row(const Dev& d, int g, int gl, int kind, const typename G::State& from, int action) {
    if constexpr (G::GAME == AZ_GAME_C4) {
        typename G::State t = from;
        const int hint = G::move_hint(t);
        G::step(t, action);
        if (G::winner_moved(t, hint) != 0 || G::full(t)) return;   // terminal leaves never reach the network
        const bool sym = G::symmetric(t);
        for (int o = 0; o < (sym ? 1 : 2); o++) {                   // the mirror image of a symmetric board is the board
            const typename G::State aug = o ? G::flip(t) : t;
            float pr[G::A], val;
            if (cache_probe(d, aug.w[0], aug.w[1], pr, val)) continue;
            const i64 m = (i64)g * d.spec_rows + (kind * G::LG + gl) * 2 + o, v = (i64)d.G + m;
            if (d.spec_mark[m]) continue;   // taken (cannot happen with one parked leaf per game and round; the list must never overflow)
#pragma unroll
            for (int i = 0; i < G::W; i++) d.leaf_state[v * G::W + i] = aug.w[i];
            d.spec_mark[m] = d.spec_tag;
            const u32 pos = atomicAdd(d.eval_count, 1u);
            d.eval_list[pos] = (u32)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Fused search: n_sims complete playouts per game in ONE launch (evaluator computed in-kernel).
// ------------------------------------------------------------------------------------------------------------
// GPW = games per wavefront: 64 / LG fills every lane; fewer games per wave (down to one: the other lanes exit at once) put the
// same games on MORE waves (a measurement hook: see pick_gpw).  Results do not depend on it (a game is touched only by its
// own lane group).
template <class G, int GPW>
__global__ __launch_bounds__(256) void k_search_fused(Dev d, int n_sims) {
    constexpr int LG = G::LG, GPB = 4 * GPW;
    static_assert(GPW * LG <= 64, "lane groups of one wave");
    __shared__ u32 s_idx[GPB][G::MAXD];
    __shared__ u32 s_N[GPB][G::MAXD];
    __shared__ float s_Q[GPB][G::MAXD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= GPW * LG) return;
    const int grp = wave * GPW + lane / LG, gl = lane % LG;
    const int g = blockIdx.x * GPB + grp;
    if (g >= d.G) return;
    if (!d.active[g]) return;
    uint4* nodes = d.nodes + (i64)g * d.cap;
    const typename G::State rs = load_root_state<G>(d, g);
    const u32 root = d.root[g];
    u32 nalloc = d.n_alloc[g];
    uint4 rr = nodes[root];
    u32 rN = rr.x, rlink = rr.w;
    float rQ = as_f(rr.y);
    u32 fpos = d.flip_pos[g];
    u64 ctr = d.sim_ctr[g];
    u32 st_levels = 0, st_evals = 0, st_term = 0, st_children = 0;
    for (int sim = 0; sim < n_sims; sim++) {
        typename G::State s = rs;  // env.copy()
        Leaf lf = walk<G>(d, nodes, s, root, rN, rQ, rlink, gl, s_idx[grp], s_N[grp], s_Q[grp]);
        typename G::State aug;
        leaf_prepare<G>(d, g, s, aug, lf, fpos, ctr);
        float probs[G::A], value = 0.f;
        if (!(lf.flags & 1u)) {
            eval_synthetic<G>(d.evaluator, d.eval_seed, aug, probs, value, gl);
            st_evals++;
        } else {
#pragma unroll
            for (int a = 0; a < G::A; a++) probs[a] = 0.f;
            st_term++;
        }
        finish<G>(d, g, nodes, aug, lf, probs, value, gl, nalloc, root, rN, rQ, rlink, s_idx[grp], s_N[grp], s_Q[grp],
                  ctr, st_children);
        st_levels += (u32)lf.depth;
        ctr++;
    }
    if (gl == 0) {
        d.n_alloc[g] = nalloc;
        d.flip_pos[g] = fpos;
        d.sim_ctr[g] = ctr;
        atomicAdd(&d.stats[ST_PLAYOUTS], (unsigned long long)n_sims);
        atomicAdd(&d.stats[ST_LEVELS], (unsigned long long)st_levels);
        atomicAdd(&d.stats[ST_EVALS], (unsigned long long)st_evals);
        atomicAdd(&d.stats[ST_TERMINAL], (unsigned long long)st_term);
        atomicAdd(&d.stats[ST_CHILDREN], (unsigned long long)st_children);
    }
}

// Statistics of the split-search kernels: counted per game in registers and added to the global counters ONCE per launch and
// wave.  (One atomicAdd per playout and counter, as round 2 had it, is a few thousand same-address atomics per launch; they
// serialise in L2 and -- with several playouts per game and launch since the evaluation cache -- were most of the tree step's
// duration: the step got SLOWER when its games were spread over more waves.)
struct Cnt {
    u32 playouts = 0, levels = 0, evals = 0, term = 0, hits = 0, children = 0;
};
__device__ inline void flush_counts(const Dev& d, const Cnt& c, int gl) {
    const u64 act = __ballot(1), leaders = __ballot(gl == 0);
    const int first = __ffsll((long long)act) - 1;
    const u32 v[6] = {c.playouts, c.levels, c.evals, c.term, c.hits, c.children};
    const int idx[6] = {ST_PLAYOUTS, ST_LEVELS, ST_EVALS, ST_TERMINAL, ST_CACHE_HITS, ST_CHILDREN};
#pragma unroll
    for (int i = 0; i < 6; i++) {
        u32 sum = 0;
        for (u64 mm = leaders; mm; mm &= mm - 1) sum += (u32)__shfl((int)v[i], __ffsll((long long)mm) - 1);
        if ((int)__lane_id() == first && sum) atomicAdd(&d.stats[idx[i]], (unsigned long long)sum);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Split search: select -> (evaluator kernel or host) -> backup, one playout per game per round.
// ------------------------------------------------------------------------------------------------------------
// Bodies of the two halves; `p_*` are this group's LDS path rows.
template <class G>
__device__ inline void select_body(const Dev& d, int g, int gl, u32* p_idx, u32* p_N, float* p_Q, Cnt& cnt) {
    constexpr int LG = G::LG;
    if (!d.active[g]) {
        if (gl == 0) d.leaf_flags[g] = 0u;
        return;
    }
    const uint4* nodes = d.nodes + (i64)g * d.cap;
    typename G::State s = load_root_state<G>(d, g);
    const u32 root = d.root[g];
    uint4 rr = nodes[root];
    Leaf lf = walk<G>(d, nodes, s, root, rr.x, as_f(rr.y), rr.w, gl, p_idx, p_N, p_Q);
    typename G::State aug;
    u32 fpos = d.flip_pos[g];
    u64 ctr = d.sim_ctr[g];
    leaf_prepare<G>(d, g, s, aug, lf, fpos, ctr);
    float cprobs[G::A], cvalue = 0.f;
    bool hit = false;
    if constexpr (G::GAME == AZ_GAME_C4) {
        if (d.cache && !(lf.flags & 1u)) hit = cache_probe(d, aug.w[0], aug.w[1], cprobs, cvalue);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int dd = gl; dd <= lf.depth; dd += LG) {  // path: LDS -> HBM, coalesced
        d.path_idx[(i64)g * G::MAXD + dd] = p_idx[dd];
        d.path_N[(i64)g * G::MAXD + dd] = p_N[dd];
        d.path_Q[(i64)g * G::MAXD + dd] = p_Q[dd];
    }
    if (gl == 0) {
#pragma unroll
        for (int i = 0; i < G::W; i++) d.leaf_state[(i64)g * G::W + i] = aug.w[i];
        d.leaf_depth[g] = lf.depth;
        d.leaf_flags[g] = lf.flags | 8u | (hit ? 16u : 0u);
        d.leaf_tvalue[g] = lf.tvalue;
        d.flip_pos[g] = fpos;
        if (hit) {   // the network's answer to this position is in the table: hand it to the backup as if K3 had written it
#pragma unroll
            for (int a = 0; a < G::A; a++) d.eval_probs[(i64)g * G::A + a] = cprobs[a];
            d.eval_value[g] = cvalue;
        } else if (!(lf.flags & 1u)) {
            u32 pos = atomicAdd(d.eval_count, 1u);  // compaction: only non-terminal leaves reach the network
            d.eval_list[pos] = (u32)g;
        }
        cnt.levels += (u32)lf.depth;
    }
}

template <class G>
__device__ inline bool backup_body(const Dev& d, int g, int gl, u32* p_idx, u32* p_N, float* p_Q, Cnt& cnt) {
    constexpr int LG = G::LG;
    const u32 flags = d.leaf_flags[g];
    if (!(flags & 8u)) return false;
    // a slot parked (az_set_active) while it holds a delivered leaf keeps the leaf, its path and the evaluator's answer (all
    // indexed by slot, nothing else writes them) until it is resumed: parking never advances ply, tree or trajectory
    if (!d.active[g]) return false;
    uint4* nodes = d.nodes + (i64)g * d.cap;
    Leaf lf;
    lf.hint = 0;
    lf.depth = d.leaf_depth[g];
    lf.flags = flags & 7u;
    lf.tvalue = d.leaf_tvalue[g];
    for (int dd = gl; dd <= lf.depth; dd += LG) {  // path: HBM -> LDS
        p_idx[dd] = d.path_idx[(i64)g * G::MAXD + dd];
        p_N[dd] = d.path_N[(i64)g * G::MAXD + dd];
        p_Q[dd] = d.path_Q[(i64)g * G::MAXD + dd];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    lf.node = p_idx[lf.depth];
    lf.link = nodes[lf.node].w;
    typename G::State aug;
#pragma unroll
    for (int i = 0; i < G::W; i++) aug.w[i] = d.leaf_state[(i64)g * G::W + i];
    float probs[G::A], value = 0.f;
    if (!(flags & 1u)) {
#pragma unroll
        for (int a = 0; a < G::A; a++) probs[a] = d.eval_probs[(i64)g * G::A + a];
        value = d.eval_value[g];
        if constexpr (G::GAME == AZ_GAME_C4) {
            if (d.cache && !(flags & 16u)) cache_insert(d, aug.w[0], aug.w[1], probs, value, gl);   // K3's answer, fresh
        }
    } else {
#pragma unroll
        for (int a = 0; a < G::A; a++) probs[a] = 0.f;
    }
    const u32 root = d.root[g];
    u32 nalloc = d.n_alloc[g];
    u32 rN = p_N[0], rlink = nodes[root].w;
    float rQ = p_Q[0];
    u32 st_children = 0;
    u64 ctr = d.sim_ctr[g];
    finish<G>(d, g, nodes, aug, lf, probs, value, gl, nalloc, root, rN, rQ, rlink, p_idx, p_N, p_Q, ctr, st_children);
    if (gl == 0) {
        d.n_alloc[g] = nalloc;
        d.sim_ctr[g] = ctr + 1;
        d.leaf_flags[g] = 0u;
        cnt.playouts++;
        if (flags & 1u) cnt.term++;
        else if (flags & 16u) cnt.hits++;
        else cnt.evals++;
        cnt.children += st_children;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// Rolling self-play (az_selfplay_rounds): the games do not move in lockstep.  Every round each live game delivers
// exactly ONE non-terminal leaf to the network: a playout that ends on a terminal node is finished on the spot and the
// game walks again, and the playout that completes a move (the S-th) is followed at once, by the game's own lane group,
// by move selection, env.step + prune_root, trajectory recording and -- when the game is over -- the slot's refill
// (az_play_dev.hpp: the same functions the lockstep K4 kernels call).  Every game still runs its S playouts per move
// one after the other with its own counter-keyed random streams, so each game's trajectory is the one lockstep
// self-play produces; what changes is that leaf batches stay full although 20-60 % of late-game playouts are terminal.
// ------------------------------------------------------------------------------------------------------------
struct Roll {
    int on;        // 0: lockstep tree step
    int cap;       // terminal playouts a game may finish in place per round
    u32 budget;    // 0, or: no further in-place playout is started once the launch is this many 10 ns ticks old
    int search;    // az_search on the rolling step: a game stops after its S playouts (no ply end).  2: the first selection
    u32* search_cnt;   // search: [0] games that take part, [1] games whose S playouts are complete
    int spec_width;    // speculation: how many of a waiting leaf's next siblings, and of its first children, ride along
    int spec_kinds;    // bit 0: siblings, bit 1: children
    int S;         // playouts per move
    int first_n;   // game.py:41 temperature schedule (<0: temp for every ply)
    int move_mode, reuse;
    double temp;
    unsigned long long target;   // games the refill may start in total (0: no refill)
    u64 gid_base;
    Play p;
};

// the move of game g is complete (its S-th playout has been backed up): AlphaZeroPlayer.get_action's tail + Game's ply
// (player.py:104-116, game.py:41-58) + the slot's refill, by the game's own lane group
template <class G>
__device__ inline void roll_ply_end(const Dev& d, const Roll& r, int g, int gl) {
    if (gl == 0) {
        float pi[G::A];
        const double T = (r.first_n >= 0 && d.ply[g] >= r.first_n) ? 1e-3 : r.temp;   // game.py:41
        const int a = choose_game<G>(d, g, T, r.move_mode, pi, 1);
        if (a >= 0) advance_game<G>(d, r.p, g, a, pi, 1, r.reuse, 1);
        if (r.target) refill_game<G>(d, r.p, g, r.target, r.gid_base);   // only acts on a finished slot
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// one playout of game g's current move has just been backed up: count it; at the S-th, end the ply.
template <class G>
__device__ inline void roll_playout_done(const Dev& d, const Roll& r, int g, int gl) {
    const int ms = d.move_sims[g] + 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (r.search) {   // az_search: the S-th playout completes the game's search; nothing else happens
        if (gl == 0) {
            d.move_sims[g] = ms;
            if (ms >= r.S) atomicAdd(&r.search_cnt[1], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        return;
    }
    if (gl == 0) d.move_sims[g] = ms >= r.S ? 0 : ms;
    if (ms >= r.S) roll_ply_end<G>(d, r, g, gl);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// select for rolling self-play: walk until a leaf is found that needs the network.  Playouts that end on a terminal node, or
// on a position whose evaluation is in the cache, are completed in place (at most r.cap of them per round: a game whose
// search has converged on a forced line ends ~all of its playouts on terminal nodes, and the launch lasts as long as its
// slowest game).  The game's state -- root position, root statistics, allocation mark, counters -- lives in registers from
// one playout to the next (as in the fused kernel) and goes back to memory at a ply end and at the end of the round: a
// playout's dependent chain is then the walk itself plus one table probe.
template <class G, bool SPEC = false>
__device__ inline void select_body_roll(const Dev& d, const Roll& r, int g, int gl, u32* p_idx, u32* p_N, float* p_Q, Cnt& cnt,
                                        u64 t_start) {
    constexpr int LG = G::LG;
    // an idle slot (parked, or its game is over and was not refilled) is left exactly as it is -- a parked leaf stays parked
    if (!d.active[g]) return;
    if constexpr (SPEC) {
        spec_flush<G>(d, g, gl);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    uint4* nodes = d.nodes + (i64)g * d.cap;
    typename G::State rs = load_root_state<G>(d, g);
    u32 root = d.root[g], nalloc = d.n_alloc[g], fpos = d.flip_pos[g];
    u64 ctr = d.sim_ctr[g];
    int ms = d.move_sims[g];
    if (r.search) {
        if (r.search == 2 && gl == 0) atomicAdd(&r.search_cnt[0], 1u);   // the first selection counts the games that take part
        if (ms >= r.S) return;                                            // this game has had its S playouts
    }
    uint4 rr = nodes[root];
    u32 rN = rr.x, rlink = rr.w;
    float rQ = as_f(rr.y);
    bool parked = false;
    for (int it = 0; it <= r.cap; it++) {
        // the launch lasts as long as its slowest game: past the time budget no game starts another in-place playout (which
        // playouts run in which round changes nothing in any game: each game's playouts follow one another all the same)
        if (it > 0 && r.budget && __builtin_amdgcn_s_memrealtime() - t_start > (u64)r.budget) break;
        typename G::State s = rs;   // env.copy()
        WalkTail<G> tail;
        Leaf lf = walk<G, SPEC>(d, nodes, s, root, rN, rQ, rlink, gl, p_idx, p_N, p_Q, &tail);
        typename G::State aug;
        leaf_prepare<G>(d, g, s, aug, lf, fpos, ctr);
        float probs[G::A], cvalue = 0.f;
        bool hit = false;
        if constexpr (G::GAME == AZ_GAME_C4) {
            if (d.cache && !(lf.flags & 1u)) hit = cache_probe(d, aug.w[0], aug.w[1], probs, cvalue);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cnt.levels += (u32)lf.depth;
        if (!(lf.flags & 1u) && !hit) {  // a leaf for the network: park the playout exactly like select_body
            for (int dd = gl; dd <= lf.depth; dd += LG) {
                d.path_idx[(i64)g * G::MAXD + dd] = p_idx[dd];
                d.path_N[(i64)g * G::MAXD + dd] = p_N[dd];
                d.path_Q[(i64)g * G::MAXD + dd] = p_Q[dd];
            }
            if (gl == 0) {
#pragma unroll
                for (int i = 0; i < G::W; i++) d.leaf_state[(i64)g * G::W + i] = aug.w[i];
                d.leaf_depth[g] = lf.depth;
                d.leaf_flags[g] = lf.flags | 8u;
                d.leaf_tvalue[g] = lf.tvalue;
                u32 pos = atomicAdd(d.eval_count, 1u);
                d.eval_list[pos] = (u32)g;
            }
            if constexpr (SPEC) {   // the leaf's unvisited siblings and its own children ride along (Speculation, above)
                // ... the next r.spec_width of each, in the order in which the search will ask for them: children are visited in
                // insertion order (the unvisited ones are the tail of the parent's array; the leaf's own children will be inserted
                // in ascending order of the EVALUATED position's actions, i.e. descending when this playout's board is mirrored)
                if ((r.spec_kinds & 1) && lf.depth > 0 && (u32)gl < tail.k && gl > tail.best && gl <= tail.best + r.spec_width && tail.n == 0u)
                    spec_row<G>(d, g, gl, 0, tail.before, (int)link_action(tail.link));
                const u32 legal = G::legal(s);
                if ((r.spec_kinds & 2) && gl < G::A && ((legal >> gl) & 1u)) {
                    const bool rev = (lf.flags & 2u) && !(lf.flags & 4u);
                    const int rank = rev ? __popc(legal >> (gl + 1)) : __popc(legal & ((1u << gl) - 1u));
                    if (rank < r.spec_width) spec_row<G>(d, g, gl, 1, s, gl);
                }
            }
            parked = true;
            break;
        }
        // terminal leaf: MCTS.py:159-164 right away (no expansion, the constant value is backed up); a leaf whose evaluation
        // is in the table: MCTS.py:147-164 right away with the table's probs and value (the bits K3 would deliver)
        if (!hit) {
#pragma unroll
            for (int a = 0; a < G::A; a++) probs[a] = 0.f;
        }
        u32 st_children = 0;
        finish<G>(d, g, nodes, aug, lf, probs, cvalue, gl, nalloc, root, rN, rQ, rlink, p_idx, p_N, p_Q, ctr, st_children);
        ctr++;
        cnt.playouts++;
        if (hit) cnt.hits++, cnt.children += st_children;
        else cnt.term++;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (r.search && ms + 1 >= r.S) {   // az_search: the game's S-th playout; its state goes back to memory and it rests
            if (gl == 0) {
                d.n_alloc[g] = nalloc;
                d.sim_ctr[g] = ctr;
                d.flip_pos[g] = fpos;
                d.move_sims[g] = ms + 1;
                d.leaf_flags[g] = 0u;
                atomicAdd(&r.search_cnt[1], 1u);
            }
            return;
        }
        if (++ms >= r.S) {   // the move is complete: state back to memory, ply end, and the new root's state in again
            ms = 0;
            if (gl == 0) {
                d.n_alloc[g] = nalloc;
                d.sim_ctr[g] = ctr;
                d.flip_pos[g] = fpos;
                d.move_sims[g] = 0;
                d.leaf_flags[g] = 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            roll_ply_end<G>(d, r, g, gl);
            if (!d.active[g]) return;   // the game is over and the slot was not refilled; everything is in memory
            rs = load_root_state<G>(d, g);
            root = d.root[g];
            nalloc = d.n_alloc[g];
            fpos = d.flip_pos[g];
            ctr = d.sim_ctr[g];
            rr = nodes[root];
            rN = rr.x, rlink = rr.w, rQ = as_f(rr.y);
        }
    }
    if (gl == 0) {
        d.n_alloc[g] = nalloc;
        d.sim_ctr[g] = ctr;
        d.flip_pos[g] = fpos;
        d.move_sims[g] = ms;
        if (!parked) d.leaf_flags[g] = 0u;   // no leaf for the network this round
    }
}

// mode bit0: backup of the evaluated playout, bit1: select of the next one.  Between two evaluator launches the
// engine issues ONE kernel that finishes playout i and walks playout i+1 (same group, same wave, so the walk sees the
// statistics the backup has just written), instead of a backup launch followed by a select launch.
// BS: workgroup size.  Under the CNN workload every CU is held by a K3 workgroup that owns its whole register file, so a
// tree-step workgroup waits for a CU to come free: fewer, larger workgroups (512 threads = 64 Connect4 games) need
// fewer CUs and start sooner.
// Two shapes of the same kernel (round 2, A/B runs on one box with tools/bench_ab.sh):
//   k_tree_step       512 threads, as many registers as it likes (176).  It cannot share a CU with a K3 workgroup (384 registers
//                     per lane), so its workgroups wait for CUs that K3 workgroups have left: 68 us per launch instead of 8.
//   k_tree_step_small 256 threads capped at 128 registers per lane (37 dwords of scratch).  It fits BESIDE a K3 workgroup and
//                     starts at once -- and the K3 workgroup that shares its CU's issue ports runs longer.
// 2 048 slots in 3 cohorts (the bench): the big shape wins, 897 against 866 games/s; the GPU is bound by K3 workgroup time and
// the tree step's wait sits in a cohort's chain only.  16 384 slots in 4 cohorts (cfg3): the tree steps are a quarter of the
// GPU's work, the CUs they hold alone are missed, and the small shape wins, 750 against 720 games/s.  launch_tree_step picks by
// the number of games per launch.
// GPW: games per wavefront (64 / LG fills every lane).  A wave runs the union of its games' paths -- the deepest walk, every
// branch any of them takes, as many in-place playouts as its busiest game -- so with fewer games per wave each game waits
// for fewer others; the tree step is far from filling the chip (2 048 games are 256 full waves), so the extra waves are free.
template <class G, int BS, int GPW = 64 / G::LG, bool SPEC = false>
__device__ __forceinline__ void tree_step_body(const Dev& d, int mode, u32* clear_count, const Roll& r) {
    constexpr int LG = G::LG, GPB = (BS / 64) * GPW;
    static_assert(GPW * LG <= 64, "lane groups of one wave");
    __shared__ u32 s_idx[GPB][G::MAXD];
    __shared__ u32 s_N[GPB][G::MAXD];
    __shared__ float s_Q[GPB][G::MAXD];
    // clear_count: a leaf counter that the evaluator has already consumed and that nothing in THIS launch counts
    // into (the select half of this launch counts into d.eval_count, which is a different counter: see the engine).
    if (clear_count && blockIdx.x == 0 && threadIdx.x == 0) *clear_count = 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= GPW * LG) return;
    const int grp = wave * GPW + lane / LG, gl = lane % LG;
    const int g = d.g0 + blockIdx.x * GPB + grp;
    if (g >= d.g0 + d.gn) return;
    const u64 t_start = r.on && r.budget ? __builtin_amdgcn_s_memrealtime() : 0ull;
    bool played = false;
    Cnt cnt;
    if (mode & 1) played = backup_body<G>(d, g, gl, s_idx[grp], s_N[grp], s_Q[grp], cnt);
    if (r.on) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (played) roll_playout_done<G>(d, r, g, gl);
        if (mode & 2) select_body_roll<G, SPEC>(d, r, g, gl, s_idx[grp], s_N[grp], s_Q[grp], cnt, t_start);
    } else if (mode & 2) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        select_body<G>(d, g, gl, s_idx[grp], s_N[grp], s_Q[grp], cnt);
    }
    flush_counts(d, cnt, gl);
}
template <class G, int BS, int GPW = 64 / G::LG>
__global__ __launch_bounds__(BS) void k_tree_step(Dev d, int mode, u32* clear_count, Roll r) {
    tree_step_body<G, BS, GPW>(d, mode, clear_count, r);
}
// the rolling step of Connect4 games with speculative rows (Speculation, above)
template <int BS, int GPW>
__global__ __launch_bounds__(BS) void k_tree_step_spec(Dev d, int mode, u32* clear_count, Roll r) {
    tree_step_body<C4, BS, GPW, true>(d, mode, clear_count, r);
}
template <class G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_tree_step_small(Dev d, int mode, u32* clear_count,
                                                                                                    Roll r) {
    tree_step_body<G, 256>(d, mode, clear_count, r);
}

// ------------------------------------------------------------------------------------------------------------
// Pure Monte-Carlo tree search: the reference's MCTS class with utils.policy_value_fn (MCTS.py:57-63,90-128,
// utils.py:26-52) -- UCT selection, value = one rollout to the end of the game, no network.  It is the opponent of
// the Elo arena (Pipeline.py:125-135, player.MCTSPlayer).  Everything here is Python-float arithmetic in the
// reference, i.e. FLOAT64: the node record keeps Q as a double in words (y, z) (priors are not used by UCT):
//   x = n_visits   (y, z) = Q (float64 bits, meaningless while x == 0)   w = link (as above)
// tab64 = {c_init + ln((1+Np+c_base)/c_base), ln(Np)} built on the host with glibc; sqrt and divide are IEEE on device.
// ------------------------------------------------------------------------------------------------------------
__device__ inline double node_q64(const uint4& r) {
    return r.x ? __hiloint2double((int)r.z, (int)r.y) : 0.0;  // TreeNode.Q starts at 0 (MCTS.py:15)
}

template <int STEP>
__device__ inline void argmax_step_f64(double& score, int& idx) {
    const u32 lo = dpp_xchg<STEP>((u32)__double2loint(score)), hi = dpp_xchg<STEP>((u32)__double2hiint(score));
    const double os = __hiloint2double((int)hi, (int)lo);
    const int oi = (int)dpp_xchg<STEP>((u32)idx);
    if (os > score || (os == score && oi < idx)) {
        score = os;
        idx = oi;
    }
}
template <int LG>
__device__ inline int group_argmax_first_f64(double score, int gl) {
    static_assert(LG == 8 || LG == 16, "lane groups of 8 or 16");
    int idx = gl;
    argmax_step_f64<0>(score, idx);
    argmax_step_f64<1>(score, idx);
    argmax_step_f64<2>(score, idx);
    if constexpr (LG == 16) argmax_step_f64<3>(score, idx);
    return idx;
}

__device__ inline int nth_set_bit(u32 m, int n) {
    for (int k = 0; k < n; k++) m &= m - 1u;
    return __ffs((int)m) - 1;
}

// utils.evaluate_rollout (utils.py:26-40) on a copy of the leaf position: play valid moves to the end, return the
// outcome seen by the player to move at the leaf.  Move choice (utils.py:49-52 draws np.random.rand per valid move and
// takes the arg-max = a uniformly random valid move):
//   AZ_EVAL_ROLLOUT       counter-based device RNG, uniform over the valid moves (statistical parity);
//   AZ_EVAL_ROLLOUT_HASH  idx = (mix64(fold(words, seed) + 'ROLL') >> 33) % n_valid (the fixture harness's rule: bit parity).
template <class G>
__device__ inline int rollout(const Dev& d, int g, typename G::State r, u64 ctr, u32& plies) {
    const int player = G::turn(r);
    int w = 0;   // the leaf handed in is not terminal; from there on only the last move can make a line
    for (int it = 0; it < 1000; it++) {
        if (w != 0 || G::full(r)) break;
        const u32 legal = G::legal(r);
        const int nv = __popc(legal);
        int idx;
        if (d.evaluator == AZ_EVAL_ROLLOUT_HASH) {
            u64 h = d.eval_seed;
#pragma unroll
            for (int i = 0; i < G::W; i++) h = mix64(h ^ r.w[i]);
            idx = (int)((mix64(h + 0x524F4C4Cull) >> 33) % (u64)nv);
        } else {
            idx = (int)(rng_uniform(d.rng_seed, d.game_id[g], ctr, 0x4000u + (u32)it) * (float)nv);
            idx = idx < nv ? idx : nv - 1;
        }
        const int hint = G::move_hint(r);
        G::step(r, nth_set_bit(legal, idx));
        w = G::winner_moved(r, hint);
        plies++;
    }
    return w == 0 ? 0 : (w == player ? 1 : -1);
}

template <class G, int GPW>
__global__ __launch_bounds__(256) void k_search_pure(Dev d, int n_sims) {
    constexpr int LG = G::LG, GPB = 4 * GPW;
    static_assert(GPW * LG <= 64, "lane groups of one wave");
    __shared__ u32 s_idx[GPB][G::MAXD];
    __shared__ u32 s_N[GPB][G::MAXD];
    __shared__ double s_Q[GPB][G::MAXD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= GPW * LG) return;
    const int grp = wave * GPW + lane / LG, gl = lane % LG;
    const int g = blockIdx.x * GPB + grp;
    if (g >= d.G) return;
    if (!d.active[g]) return;
    uint4* nodes = d.nodes + (i64)g * d.cap;
    const typename G::State rs = load_root_state<G>(d, g);
    const u32 root = d.root[g];
    u32 nalloc = d.n_alloc[g];
    uint4 rr = nodes[root];
    u32 rN = rr.x, rlink = rr.w;
    double rQ = node_q64(rr);
    u32 fpos = d.flip_pos[g];
    u64 ctr = d.sim_ctr[g];
    u32 st_levels = 0, st_evals = 0, st_term = 0, st_children = 0, st_rplies = 0;
    for (int sim = 0; sim < n_sims; sim++) {
        typename G::State s = rs;  // env.copy()
        // ---- select_leaf_node(env, UCT=True): MCTS.py:108-113 with TreeNode.UCT (MCTS.py:57-63)
        u32 node = root, Np = rN, link = rlink;
        int depth = 0, hint = 0;
        if (gl == 0) {
            s_idx[grp][0] = root;
            s_N[grp][0] = rN;
            s_Q[grp][0] = rQ;
        }
        while (link_nchild(link) > 0 && depth < G::MAXD - 1) {
            const u32 k = link_nchild(link), first = link_first(link);
            double2 tb;
            if (Np < (u32)d.tab_n) tb = d.tab64[Np];
            else {
                atomicOr(d.err, 2u);
                tb = make_double2(d.c_init + log((double)(1u + Np + (u32)d.c_base) / (double)d.c_base), log((double)Np));
            }
            uint4 rec = make_uint4(0u, 0u, 0u, 0u);
            double score = -INFINITY;
            if ((u32)gl < k) {
                rec = nodes[first + gl];
                if (rec.x == 0u) score = INFINITY;  // u = inf for an unvisited child; -Q + inf = inf
                else {
                    double t = tb.y / (double)rec.x;    // math.log(parent.n_visits) / n_visits
                    double u = tb.x * sqrt(t);
                    score = -node_q64(rec) + u;
                }
            }
            const int best = group_argmax_first_f64<LG>(score, gl);
            const u32 cN = __shfl(rec.x, best, LG), cy = __shfl(rec.y, best, LG), cz = __shfl(rec.z, best, LG);
            const u32 cl = __shfl(rec.w, best, LG);
            node = first + (u32)best;
            depth++;
            if (gl == 0) {
                s_idx[grp][depth] = node;
                s_N[grp][depth] = cN;
                s_Q[grp][depth] = cN ? __hiloint2double((int)cz, (int)cy) : 0.0;
            }
            hint = G::move_hint(s);
            G::step(s, (int)link_action(cl));
            Np = cN;
            link = cl;
        }
        Leaf lf;
        lf.hint = hint;
        lf.node = node;
        lf.link = link;
        lf.depth = depth;
        lf.flags = 0;
        lf.tvalue = 0.f;
        typename G::State aug;
        leaf_prepare<G>(d, g, s, aug, lf, fpos, ctr);  // MCTS.py:117: the flip draw happens on every playout
        const bool term = lf.flags & 1u, flipped = lf.flags & 2u, sym = lf.flags & 4u;
        // ---- policy_value_fn(env_aug): uniform priors (unused by UCT) and the rollout value (utils.py:43-46).
        //      On a finished game the rollout returns at once with the outcome = lf.tvalue.
        double v = (double)lf.tvalue;
        if (!term) {
            v = (double)rollout<G>(d, g, aug, ctr, st_rplies);
            st_evals++;
            // ---- node.expand(action_probs) (MCTS.py:119-122, 71-77): children in the evaluated env's ascending
            //      valid-move order, actions mapped back through flip_action when the playout was mirrored
            const u32 legal = G::legal(aug);
            const int nv = __popc(legal);
            const u32 first = nalloc;
            if ((i64)first + nv <= d.cap && first + (u32)nv <= LINK_FIRST_MASK) {
                if (gl < nv) {
                    int a = nth_set_bit(legal, gl);
                    if (flipped) a = G::flip_action(sym, a);
                    nodes[first + gl] = make_uint4(0u, 0u, 0u, (u32)a << 28);
                }
                const u32 nl = (lf.link & 0xF0000000u) | first | ((u32)nv << 24);
                if (gl == 0) reinterpret_cast<u32*>(&nodes[lf.node])[3] = nl;
                if (lf.node == root) rlink = nl;
                nalloc = first + (u32)nv;
                st_children += (u32)nv;
            } else {
                atomicOr(d.err, 1u);
            }
        } else {
            st_term++;
        }
        // ---- node.update(leaf_value) (MCTS.py:82-87) in float64
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int dd = gl; dd <= depth; dd += LG) {
            const u32 idx = s_idx[grp][dd], N1 = s_N[grp][dd] + 1u;
            const double Q = s_Q[grp][dd];
            const double val = ((depth - dd) & 1) ? -v : v;
            double dq = val - Q;
            dq = dq / (double)N1;
            const double Q1 = Q + dq;
            u32* w = reinterpret_cast<u32*>(&nodes[idx]);
            w[0] = N1;
            w[1] = (u32)__double2loint(Q1);
            w[2] = (u32)__double2hiint(Q1);
        }
        {
            const double val = (depth & 1) ? -v : v;
            const u32 N1 = rN + 1u;
            double dq = val - rQ;
            dq = dq / (double)N1;
            rQ = rQ + dq;
            rN = N1;
        }
        st_levels += (u32)depth;
        ctr++;
    }
    if (gl == 0) {
        d.n_alloc[g] = nalloc;
        d.flip_pos[g] = fpos;
        d.sim_ctr[g] = ctr;
        atomicAdd(&d.stats[ST_PLAYOUTS], (unsigned long long)n_sims);
        atomicAdd(&d.stats[ST_LEVELS], (unsigned long long)st_levels);
        atomicAdd(&d.stats[ST_EVALS], (unsigned long long)st_evals);
        atomicAdd(&d.stats[ST_TERMINAL], (unsigned long long)st_term);
        atomicAdd(&d.stats[ST_CHILDREN], (unsigned long long)st_children);
        atomicAdd(&d.stats[ST_ROLLOUT_PLIES], (unsigned long long)st_rplies);
    }
}

// (action, visits, Q, prior) of the root children in insertion order: MCTS.py:170-171.
template <class G>
__global__ __launch_bounds__(256) void k_root_children(Dev d, i32* actions, i32* visits, float* q, float* prior,
                                                      i32* nch, i32* rootN, float* rootQ) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    const uint4* nodes = d.nodes + (i64)g * d.cap;
    uint4 rr = nodes[d.root[g]];
    u32 k = link_nchild(rr.w), first = link_first(rr.w);
    if (nch) nch[g] = (i32)k;
    if (rootN) rootN[g] = (i32)rr.x;
    if (rootQ) rootQ[g] = d.tab64 ? (float)node_q64(rr) : as_f(rr.y);
    for (int i = 0; i < G::A; i++) {
        uint4 c = (u32)i < k ? nodes[first + i] : make_uint4(0u, 0u, 0u, 0xF0000000u);
        if (actions) actions[(i64)g * G::A + i] = (u32)i < k ? (i32)link_action(c.w) : -1;
        if (visits) visits[(i64)g * G::A + i] = (i32)c.x;
        if (q) q[(i64)g * G::A + i] = d.tab64 ? (float)node_q64(c) : as_f(c.y);
        if (prior) prior[(i64)g * G::A + i] = as_f(c.z);
    }
}

// MCTS.py:92,135: TreeNode(None, 1) -- fresh single-node tree for every (selected) slot.
__global__ __launch_bounds__(256) void k_reset_trees(Dev d, const u8* only) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    if (only && !only[g]) return;
    d.nodes[(i64)g * d.cap] = make_uint4(0u, 0u, as_u(1.0f), 0u);
    d.root[g] = 0u;
    d.n_alloc[g] = 1u;
    d.leaf_flags[g] = 0u;
}

__global__ __launch_bounds__(256) void k_puct(i64 n, const i32* Np, const i32* nv, const float* Q, const float* P,
                                              const float2* tab, i32 tab_n, double c_init, i32 c_base, float* score) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 np = (u32)Np[i];
    float2 tb;
    if (np < (u32)tab_n) tb = tab[np];
    else {
        double A = c_init + log((double)(1u + np + (u32)c_base) / (double)c_base);
        tb = make_float2((float)A, (float)sqrt((double)np));
    }
    score[i] = puct_score((u32)nv[i], Q[i], P[i], tb);
}

#define AZ_DISPATCH_GAME(game, ...)                                 \
    switch (game) {                                                 \
        case AZ_GAME_C4: { using GT = C4; __VA_ARGS__; } break;     \
        case AZ_GAME_TTT: { using GT = TTT; __VA_ARGS__; } break;   \
        case AZ_GAME_NBTTT: { using GT = NBTTT; __VA_ARGS__; } break; \
        default: set_error("unknown game"); return AZ_ERR_ARG;      \
    }

static inline int game_of(const Dev& d) { return d.game; }

// games per wave of the fused searches (a game is touched only by its own lane group, so results do not depend on it).
// Always a power of two that one of the instantiated kernels has (8 / 4 / 2 / 1), so the grid computed from it matches the
// kernel that is launched; AZ_TREE_GPW overrides the default (measurement hook) and is rounded DOWN to such a value.
// Default by regime (round 3, tools/gpw_sweep.sh, profiles/r03_tree_env_pmc.txt; launch time in ms, full / half / quarter waves):
//   TicTacToe x 100      4 096 games 0.366 / 0.380 / 0.492    16 384: 0.578 / 0.921 / 1.448    65 536: 1.877 / 3.044 / 4.996
//   NBTTT x 800          4 096 games 5.25 / 4.33 / 5.46        8 192: 6.62 / 5.96 / 9.63        16 384: 10.54 / 10.74 / 17.5
//   pure MCTS C4 x 1 000 4 096 games 26.1 / 22.9 / 22.7                                         16 384: 31.3 / 40.5 / 58.6
// Below a few thousand games a launch lasts as long as one game's dependent chain and a wave as long as its slowest game:
// half-filled waves idle less behind the deepest walk / longest rollout.  Once the chip's issue slots are full (the launch
// time grows with the number of games) every extra wave costs its share of them and full waves win -- by 1.6 x where all
// games look alike (TicTacToe with uniform priors), by 7 % where they diverge (NBTTT).  The switch-over sits where the
// measured curves cross.
template <class GT>
static int gpw_full_from() {
    return GT::GAME == AZ_GAME_TTT ? 4096 : (GT::GAME == AZ_GAME_NBTTT ? 12288 : 8192);
}
template <class GT>
static int pick_gpw(const Dev& d) {
    constexpr int full = 64 / GT::LG;
    static const char* env = getenv("AZ_TREE_GPW");
    int gpw = env ? atoi(env) : (d.G >= gpw_full_from<GT>() ? full : full / 2);
    if (gpw < 1) gpw = 1;
    if (gpw > full) gpw = full;
    while (gpw & (gpw - 1)) gpw &= gpw - 1;   // round down to a power of two
    return gpw;
}

int launch_search_fused(const Dev& d, int n_sims, hipStream_t st) {
    AZ_DISPATCH_GAME(game_of(d), {
        const int gpw = pick_gpw<GT>(d), gpb = 4 * gpw;
        const dim3 grid((d.G + gpb - 1) / gpb), blk(256);
        if (gpw >= 8) {
            if constexpr (GT::LG <= 8) k_search_fused<GT, 8><<<grid, blk, 0, st>>>(d, n_sims);
        } else if (gpw >= 4) k_search_fused<GT, 4><<<grid, blk, 0, st>>>(d, n_sims);
        else if (gpw >= 2) k_search_fused<GT, 2><<<grid, blk, 0, st>>>(d, n_sims);
        else k_search_fused<GT, 1><<<grid, blk, 0, st>>>(d, n_sims);
    });
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int launch_search_pure(const Dev& d, int n_sims, hipStream_t st) {
    AZ_DISPATCH_GAME(game_of(d), {
        const int gpw = pick_gpw<GT>(d), gpb = 4 * gpw;
        const dim3 grid((d.G + gpb - 1) / gpb), blk(256);
        if (gpw >= 8) {
            if constexpr (GT::LG <= 8) k_search_pure<GT, 8><<<grid, blk, 0, st>>>(d, n_sims);
        } else if (gpw >= 4) k_search_pure<GT, 4><<<grid, blk, 0, st>>>(d, n_sims);
        else if (gpw >= 2) k_search_pure<GT, 2><<<grid, blk, 0, st>>>(d, n_sims);
        else k_search_pure<GT, 1><<<grid, blk, 0, st>>>(d, n_sims);
    });
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

#ifndef AZ_TREE_STEP_BS
#define AZ_TREE_STEP_BS 512
#endif
#ifndef AZ_TREE_STEP_SMALL_FROM
#define AZ_TREE_STEP_SMALL_FROM 1000   // games per launch from which the small shape is used (683 per launch: big wins 895 : 870; 1 024: small wins 905 : 850)
#endif
// mode 2 = select, 1 = backup, 3 = backup of playout i then select of playout i+1
static int launch_tree_step(const Dev& d, int mode, u32* clear_count, hipStream_t st, const Roll* roll = nullptr) {
    Roll r;
    memset(&r, 0, sizeof(r));
    if (roll) r = *roll;
    if (r.on && d.spec_rows && d.cache && game_of(d) == AZ_GAME_C4) {
        // up to 256 games per launch: one game per wave (nobody's walk waits for a neighbour's probes; 512 slots in two cohorts: 1 191
        // games/s against 947 with two games per wave); else the shapes of the plain step
        if (d.gn <= 256) k_tree_step_spec<256, 1><<<dim3((d.gn + 3) / 4), dim3(256), 0, st>>>(d, mode, clear_count, r);
        else if (d.gn <= 768) k_tree_step_spec<512, 2><<<dim3((d.gn + 15) / 16), dim3(512), 0, st>>>(d, mode, clear_count, r);
        else if (d.gn <= 1536) k_tree_step_spec<512, 4><<<dim3((d.gn + 31) / 32), dim3(512), 0, st>>>(d, mode, clear_count, r);
        else k_tree_step_spec<512, 8><<<dim3((d.gn + 63) / 64), dim3(512), 0, st>>>(d, mode, clear_count, r);
        AZ_HIP(hipGetLastError());
        return AZ_OK;
    }
    AZ_DISPATCH_GAME(game_of(d), {
        static const char* shape = getenv("AZ_TREE_STEP");   // "big" / "small": measurement hook
        // with the evaluation cache a launch runs several playouts per game and K3 launches no longer fill the chip: the big
        // shape wins at every size (2 048 slots 2 093 : 1 927 games/s, 16 384 slots 4 972 : 4 869)
        const bool small = shape ? shape[0] == 's' : (!d.cache && d.gn >= AZ_TREE_STEP_SMALL_FROM);
        if (small) {
            constexpr int gpb = 256 / GT::LG;
            k_tree_step_small<GT><<<dim3((d.gn + gpb - 1) / gpb), dim3(256), 0, st>>>(d, mode, clear_count, r);
        } else {
            constexpr int BS = AZ_TREE_STEP_BS, full = 64 / GT::LG;
            static const char* genv = getenv("AZ_TREE_STEP_GPW");   // measurement hook: games per wave of the big shape
            // default by the size of the launch (Connect4, 4 cohorts, cap 6, games/s with 8 / 4 / 2 / 1 games per wave): 512 games
            // per launch 2 168 / 2 347 / 2 487 / 2 409; 4 096 games per launch 4 972 / - / 4 037 / -.  A wave runs the union of
            // its games' paths, so few games per wave finish sooner while the chip has idle SIMDs; a big launch needs them
            int gpw = genv ? atoi(genv) : (d.gn <= 768 ? full / 4 : (d.gn <= 1536 ? full / 2 : full));
            if (gpw < 1) gpw = 1;
            gpw = gpw >= full ? full : (gpw >= full / 2 ? full / 2 : (gpw >= full / 4 && full / 4 >= 1 ? full / 4 : 1));
            const int gpb = (BS / 64) * gpw;
            const dim3 grid((d.gn + gpb - 1) / gpb), blk(BS);
            if (gpw == full) k_tree_step<GT, BS, full><<<grid, blk, 0, st>>>(d, mode, clear_count, r);
            else if (gpw == full / 2) k_tree_step<GT, BS, full / 2><<<grid, blk, 0, st>>>(d, mode, clear_count, r);
            else if (full / 4 >= 1 && gpw == full / 4) k_tree_step<GT, BS, (full / 4 >= 1 ? full / 4 : 1)><<<grid, blk, 0, st>>>(d, mode, clear_count, r);
            else k_tree_step<GT, BS, 1><<<grid, blk, 0, st>>>(d, mode, clear_count, r);
        }
    });
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}
int launch_select(const Dev& d, hipStream_t st, bool clear_count) {
    if (clear_count) AZ_HIP(hipMemsetAsync(d.eval_count, 0, sizeof(u32), st));
    return launch_tree_step(d, 2, nullptr, st);
}
// the evaluator has consumed d.eval_count: the backup clears it for the next selection
int launch_backup(const Dev& d, hipStream_t st) { return launch_tree_step(d, 1, d.eval_count, st); }
// backup + select in one launch.  The counter the new selection counts into (d.eval_count) cannot be cleared inside the
// launch (workgroups that already select count while others still back up), so evaluator rounds alternate between two
// counters: `stale_count` is the one the evaluator has just consumed; it is cleared here for the round after next.
int launch_backup_select(const Dev& d, u32* stale_count, hipStream_t st) {
    return launch_tree_step(d, 3, stale_count, st);
}
// rolling self-play: mode 2 = the very first selection, 3 = finish round i (+ ply ends) and select round i+1
int launch_roll_step(const Dev& d, int mode, u32* stale_count, const Play& p, int S, double temp, int first_n, int move_mode,
                     int reuse, unsigned long long target, u64 gid_base, hipStream_t st, u32* search_cnt) {
    Roll r;
    memset(&r, 0, sizeof(r));
    r.on = 1;
    r.search = search_cnt ? (mode == 2 ? 2 : 1) : 0;
    r.search_cnt = search_cnt;
    static const char* cap_env = getenv("AZ_ROLL_CAP");         // tuning hooks
    static const char* bud_env = getenv("AZ_ROLL_BUDGET_US");
    // In-place playouts per game and round.  Without the evaluation cache only terminal playouts are finished in place: at most
    // 3 (round 2's measurement).  With it most playouts are, a launch lasts as long as its slowest game, and a fixed count makes
    // every launch as long as the deepest game's chain of that many playouts (cap 4 / 6 / 8 -> 2 411 / 2 494 / 2 447 games/s at
    // 2 048 slots).  So the limit is the launch's AGE instead: no game starts another in-place playout once the launch is
    // `budget` old (s_memrealtime, 10 ns ticks); games with shallow trees get through more playouts in that time, none holds
    // the launch up.  One box, cap 6 without budget against cap 16 with: 512 games per launch 2 989 -> 3 216 / 3 524 / 3 450 /
    // 3 419 games/s for 50 / 70 / 90 / 110 us; 1 024 per launch 4 828 -> 4 828 / 5 057 / 5 166 for 80 / 110 / 150 us; 4 096 per launch
    // 7 158 -> 6 752 / 7 377 / 7 685 / 8 090 for 100 / 150 / 200 / 300 us (another box: 7 665 / 7 884 / 7 426 for 300 / 450 / 600 us).
    // ... and 2 048 per launch 5 232 -> 5 770 / 5 428 for 200 / 350 us, 1 024 per launch 5 270 / 5 127 / 4 894 for 120 / 160 / 200 us:
    // the best budget grows with the launch (more games per wave, more waves per SIMD): 40 us + 1 us per 16 games.
    const double bud_us = bud_env ? atof(bud_env) : (d.cache ? std::min(40.0 + d.gn / 16.0, 400.0) : 0.0);
    r.budget = bud_us > 0 ? (u32)(bud_us * 100.0) : 0u;
    r.cap = cap_env ? atoi(cap_env) : (d.cache ? (r.budget ? 16 : 6) : 3);
    // az_search over a handful of games (the drop-in player's single game): nobody waits for anybody, a game walks until it finds a
    // leaf for the network
    if (search_cnt && d.gn <= 8 && !bud_env && !cap_env) r.budget = 0u, r.cap = 64;
    // every speculative row is a row of K3 work: all of them while the launch stays within one workgroup per CU, fewer above
    static const char* sw_env = getenv("AZ_SPEC_WIDTH");   // measurement hook
    // (rolling self-play, games/s without / with: 32 slots 44 / 67 at width 7; 128: 168 / 272 at 7; 256: 346 / 382, 487, 518, 505 at
    // 7, 3, 2, 1; 512 in two cohorts: 768 / 847, 937, 1 014, 1 191 at 7, 3, 2, 1; 1 024 in three: 1 723 / 1 748 at 1, 1 972 with the next
    // sibling only; 2 048 in four: 3 433 / 2 135 at 1, 3 068 with the sibling only -- there K3 has no CU time to spare)
    r.spec_width = sw_env ? atoi(sw_env) : (d.gn <= 64 ? 7 : (d.gn <= 128 ? 3 : (d.gn <= 192 ? 2 : 1)));
    static const char* sk_env = getenv("AZ_SPEC_KINDS");   // measurement hook
    r.spec_kinds = sk_env ? atoi(sk_env) : (d.gn <= 256 ? 3 : 1);
    r.S = S;
    r.first_n = first_n;
    r.move_mode = move_mode;
    r.reuse = reuse;
    r.temp = temp;
    r.target = target;
    r.gid_base = gid_base;
    r.p = p;
    return launch_tree_step(d, mode, stale_count, st, &r);
}

int launch_root_children(const Dev& d, i32* actions, i32* visits, float* q, float* prior, i32* nch, i32* rootN,
                         float* rootQ, hipStream_t st) {
    AZ_DISPATCH_GAME(game_of(d), k_root_children<GT><<<dim3((d.G + 255) / 256), dim3(256), 0, st>>>(d, actions, visits, q, prior, nch, rootN, rootQ));
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int launch_reset_trees(const Dev& d, const u8* only, hipStream_t st) {
    k_reset_trees<<<dim3((d.G + 255) / 256), dim3(256), 0, st>>>(d, only);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

int launch_puct(double c_init, i32 c_base, i64 n, const i32* Np, const i32* nv, const float* Q, const float* P,
                const float2* tab, i32 tab_n, float* score, hipStream_t st) {
    k_puct<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, Np, nv, Q, P, tab, tab_n, c_init, c_base, score);
    AZ_HIP(hipGetLastError());
    return AZ_OK;
}

}  // namespace az
