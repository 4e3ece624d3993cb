// az_engine.hip -- host side of libazamd.so: the C ABI of include/azamd.h, engine handles, device arenas.
//
// Reference interfaces replaced (see include/azamd.h for the per-function citations):
//   MCTS / MCTS_AZ object life cycle and get_action_visits   MCTS.py:90-171
//   AlphaZeroPlayer.get_action                                player.py:101-116
//   Game.start_self_play / Pipeline.data_collector            game.py:35-58, Pipeline.py:53-63
// There is no CPU path in this file: every compute entry point launches gfx950 kernels or fails.
#include <math.h>
#include <string.h>

#include <mutex>
#include <algorithm>
#include <vector>

#include "az_cnn.hpp"
#include "az_internal.hpp"
#include "az_play.hpp"

namespace az {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
static int bad_arg(const char* fn) {
    set_error(std::string(fn) + ": bad argument (null pointer or count out of range)");
    return AZ_ERR_ARG;
}

struct GameInfo {
    int A, W, LG, MAXD, MAX_PLIES;
};
static bool game_info(int game, GameInfo& gi) {
    switch (game) {
        case AZ_GAME_C4: gi = {C4::A, C4::W, C4::LG, C4::MAXD, C4::MAX_PLIES}; return true;
        case AZ_GAME_TTT: gi = {TTT::A, TTT::W, TTT::LG, TTT::MAXD, TTT::MAX_PLIES}; return true;
        case AZ_GAME_NBTTT: gi = {NBTTT::A, NBTTT::W, NBTTT::LG, NBTTT::MAXD, NBTTT::MAX_PLIES}; return true;
    }
    return false;
}

static int use_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no usable HIP device: libazamd has no CPU path");
        return AZ_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device ordinal out of range");
        return AZ_ERR_ARG;
    }
    AZ_HIP(hipSetDevice(device));
    return AZ_OK;
}

// {f32(c_init + ln((1+Np+c_base)/c_base)), f32(sqrt(Np))}: float64 on the host (glibc), rounded once (MCTS.py:53-54).
static void build_puct_table(double c_init, int c_base, int n, std::vector<float2>& out) {
    out.resize(n);
    for (int np = 0; np < n; np++) {
        double A = c_init + log((double)(1 + np + c_base) / (double)c_base);
        out[np] = make_float2((float)A, (float)sqrt((double)np));
    }
}


// Device staging of the host-buffer entry points (the K1 batch calls, az_puct_batch, az_net_forward): the drop-in's Env / CNN
// methods come through here once per reference call (env.valid_move(), net.predict(): env_cython.pyx / Network.py:78-85), and five
// hipMalloc + hipFree per call were most of their ~150 us.  Small requests are carved out of a grow-only block per host thread
// and device, kept until the process ends (every such entry point synchronises before it returns, so the next call may reuse
// the block); requests beyond POOL_MAX bytes per call are allocated and freed as before.
struct StagePool {
    static constexpr size_t POOL_MAX = 8u << 20;
    void* base[64] = {};
    size_t cap[64] = {};
    // no destructor: a thread-local's destructor may run after the HIP runtime has shut down; 8 MB per thread and device stay
    // allocated until the process ends
};
static thread_local StagePool g_pool;

struct TmpBufs {
    std::vector<void*> v;   // allocations of this call outside the pool
    size_t used = 0;
    int dev = -1;
    ~TmpBufs() {
        for (void* p : v) (void)hipFree(p);
    }
    template <class T>
    int get(T** p, size_t n) {
        const size_t bytes = (n * sizeof(T) + 64 + 255) & ~(size_t)255;
        if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = -1;
        if (dev >= 0 && dev < 64 && used + bytes <= StagePool::POOL_MAX) {
            if (g_pool.cap[dev] == 0) {   // the whole block at once: later requests of this call must not move earlier ones
                void* q = nullptr;
                if (hipMalloc(&q, StagePool::POOL_MAX) == hipSuccess) {
                    g_pool.base[dev] = q;
                    g_pool.cap[dev] = StagePool::POOL_MAX;
                } else {
                    (void)hipGetLastError();
                }
            }
            if (used + bytes <= g_pool.cap[dev]) {
                *p = reinterpret_cast<T*>(static_cast<char*>(g_pool.base[dev]) + used);
                used += bytes;
                return AZ_OK;
            }
        }
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, bytes);
        if (e != hipSuccess) {
            set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
            (void)hipGetLastError();
            return AZ_ERR_HIP;
        }
        v.push_back(q);
        *p = (T*)q;
        return AZ_OK;
    }
};

// One contiguous device block and its host mirror: the inputs of a host-buffer call go down in ONE hipMemcpy and all its
// outputs come back in ONE (a drop-in Env method is a single position: five separate 1-to-16-byte copies were ~40 us each).
struct Block {
    char* dev = nullptr;
    std::vector<char> host;
    size_t size = 0;
    size_t reserve(size_t bytes) {
        const size_t o = size;
        size = (size + bytes + 255) & ~(size_t)255;
        return o;
    }
    int alloc(TmpBufs& t) {
        host.resize(size);
        return t.get(&dev, size);
    }
    template <class T>
    T* d(size_t off) { return reinterpret_cast<T*>(dev + off); }
    void put(size_t off, const void* src, size_t bytes) { memcpy(host.data() + off, src, bytes); }
    void take(size_t off, void* dst, size_t bytes) {
        if (dst) memcpy(dst, host.data() + off, bytes);
    }
};

// ---- az_net_forward_static: the staging memory of the allocation-free forward.  One file-scope device array; every
// GPU the library runs on holds its own copy at its own address until the process ends, so the address is looked up once
// per device (hipGetSymbolAddress with that device current) and only that pointer is ever used from host code.
struct StaticStage {
    u64 states[AZ_NET_FORWARD_STATIC_MAX * 2];
    float logp[AZ_NET_FORWARD_STATIC_MAX * 7];
    float logv[AZ_NET_FORWARD_STATIC_MAX * 3];
    float probs[AZ_NET_FORWARD_STATIC_MAX * 7];
    float value[AZ_NET_FORWARD_STATIC_MAX];
};
static __device__ __align__(256) unsigned char scratch[sizeof(StaticStage)];   // 352 KiB
constexpr int MAX_DEVICES = 64;
static StaticStage* g_stage[MAX_DEVICES] = {nullptr};
static std::mutex g_stage_mu;

// device `dev` is current.  Called from az_engine_create, i.e. never from inside the allocation-free entry point's hot path.
static int static_stage_init(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) {
        set_error("device ordinal beyond the static staging table");
        return AZ_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_stage_mu);
    if (g_stage[dev]) return AZ_OK;
    void* p = nullptr;
    AZ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) {
        set_error("hipGetSymbolAddress(scratch) returned a null address");
        return AZ_ERR_HIP;
    }
    g_stage[dev] = reinterpret_cast<StaticStage*>(p);
    return AZ_OK;
}

}  // namespace az

using namespace az;

enum { PROF_CNN = 0, PROF_SELECT = 1, PROF_BACKUP = 2, PROF_FUSED = 3 };

struct az_engine {
    az_config cfg;
    GameInfo gi;
    Dev d;
    Play p;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    u8* flip_bits = nullptr;
    i32* d_actions = nullptr;
    float* d_pi = nullptr;
    double* d_temps = nullptr;
    i32 *rc_actions = nullptr, *rc_visits = nullptr, *rc_n = nullptr, *rc_rootN = nullptr;
    float *rc_q = nullptr, *rc_p = nullptr, *rc_rootQ = nullptr;
    u8* d_flags8 = nullptr;
    Cnn* net = nullptr;
    int k3_kind = AZ_K3_DEFAULT;   // az_net_set_kernel
    u64 cache_loads = 0;           // az_net_load calls so far: part of the evaluation cache's epoch
    unsigned long long target = 0;
    int recording = 0;
    // rolling self-play (az_selfplay_rounds): leaves are parked between calls; parity of the leaf counter in use
    bool roll_started = false;
    u32* search_cnt = nullptr;     // [2] az_search on the rolling step: games taking part, games complete
    int roll_par = 0;
    // cohorts: the slots are cut into C ranges whose select -> CNN -> backup chains run on their own streams, so one
    // cohort's tree kernels and the ragged last round of its CNN launch overlap with the other cohort's CNN launch.
    int C = 1;
    hipStream_t cstream[AZ_MAX_COHORTS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[AZ_MAX_COHORTS] = {};
    // live kernel timing with HIP events on the engine stream (az_set_profiling / az_get_profile)
    int prof_on = 0;      // mask of the categories being timed
    bool prof_sample0 = false;   // categories other than the CNN: only every fourth launch on the engine stream (cohort 0)
    unsigned prof_tick = 0;
    bool prof_blocks = false;    // az_set_profiling(e, 4): only blocks of 64 consecutive rounds out of every 512
    unsigned prof_round = 0;     // rounds seen (CNN launches on the engine stream)
    bool prof_skip = true;
    std::vector<hipEvent_t> prof_ev;   // pairs: [2i] start, [2i+1] stop
    std::vector<int> prof_cat;
    size_t prof_used = 0;
    double prof_ms[4] = {0, 0, 0, 0};
    long long prof_n[4] = {0, 0, 0, 0};
    hipEvent_t prof_base = nullptr;          // time origin of the busy-interval union
    double prof_busy[4] = {0, 0, 0, 0};      // per category: time during which at least one launch was executing
    double prof_last_end[4] = {0, 0, 0, 0};

    template <class T>
    int dalloc(T** p, size_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, n * sizeof(T) + 64);
        if (e != hipSuccess) {
            set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
            (void)hipGetLastError();   // the runtime keeps the failure as its "last error": the next launch check would report it
            return AZ_ERR_HIP;
        }
        allocs.push_back(q);
        *p = (T*)q;
        return AZ_OK;
    }
};


static int prof_harvest(az_engine* e) {
    if (e->prof_used == 0) return AZ_OK;
    for (size_t i = 0; i < e->prof_used; i++) AZ_HIP(hipEventSynchronize(e->prof_ev[2 * i + 1]));
    std::vector<std::pair<double, double>> iv[4];
    for (size_t i = 0; i < e->prof_used; i++) {
        float ms = 0.f, t0 = 0.f;
        AZ_HIP(hipEventElapsedTime(&ms, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]));
        AZ_HIP(hipEventElapsedTime(&t0, e->prof_base, e->prof_ev[2 * i]));
        const int c = e->prof_cat[i];
        e->prof_ms[c] += ms;
        e->prof_n[c] += 1;
        iv[c].push_back({(double)t0, (double)t0 + (double)ms});
    }
    for (int c = 0; c < 4; c++) {  // union of the execution intervals (launches of different cohorts overlap)
        std::sort(iv[c].begin(), iv[c].end());
        for (auto& x : iv[c]) {
            double lo = std::max(x.first, e->prof_last_end[c]);
            if (x.second > lo) {
                e->prof_busy[c] += x.second - lo;
                e->prof_last_end[c] = x.second;
            }
        }
    }
    e->prof_used = 0;
    return AZ_OK;
}
static int prof_begin(az_engine* e, int cat, hipStream_t st) {
    e->prof_skip = !(e->prof_on & (1 << cat));
    // az_set_profiling(e, 3): of the tree-step launches only every fourth one of ONE cohort (the one on the engine stream).  An
    // event record is a packet of its own in the cohort's queue between two dependent kernels, i.e. a few microseconds on
    // every round's chain: timing every tree step as well cost the bench 3 %; a sample of 1 in 4 n_cohorts does not show.
    if (e->prof_sample0 && cat != PROF_CNN && (st != e->stream || (e->prof_tick++ & 3u))) e->prof_skip = true;
    // az_set_profiling(e, 4): the event packets sit between dependent kernels of a cohort's chain and cost a bench round 6 us per
    // gap (profiles/r03_bench_count6_stats.txt: 213 us per round with events around every K3 launch, 199 us with none).  Mode 4
    // times blocks of 64 consecutive rounds out of every 512: inside a block every cohort's K3 launch (so that the union of
    // their intervals means something) and cohort 0's tree step; outside it nothing.  The cohorts' streams drift apart by a
    // round or two between two synchronisations, so short blocks of different cohorts do not coincide in time and the union
    // comes out too long: blocks of 16 / 64 / 128 / 256 rounds read K3's rate 9 / 1 / 1 / 1 % below what timing every launch reads.
    if (e->prof_blocks) {
        if (cat == PROF_CNN && st == e->stream) e->prof_round++;
        static const char* benv = getenv("AZ_PROF_BLOCK");   // measurement hook: log2 of the block length (period = 8 blocks)
        const unsigned lb = benv ? (unsigned)atoi(benv) : 6u;
        const bool in_block = ((e->prof_round - 1u) & ((8u << lb) - 1u)) < (1u << lb);
        if (!in_block || (cat != PROF_CNN && st != e->stream)) e->prof_skip = true;
    }
    if (e->prof_skip) return AZ_OK;
    if (e->prof_used * 2 >= e->prof_ev.size()) {
        int rc = prof_harvest(e);
        if (rc != AZ_OK) return rc;
    }
    e->prof_cat[e->prof_used] = cat;
    AZ_HIP(hipEventRecord(e->prof_ev[2 * e->prof_used], st));
    return AZ_OK;
}
static int prof_end(az_engine* e, hipStream_t st) {
    if (e->prof_skip) return AZ_OK;
    AZ_HIP(hipEventRecord(e->prof_ev[2 * e->prof_used + 1], st));
    e->prof_used++;
    return AZ_OK;
}

#define AZ_TRY(x)                 \
    do {                          \
        int _r = (x);             \
        if (_r != AZ_OK) return _r; \
    } while (0)
#define AZ_CHECK_E(e)                    \
    if (!(e)) {                          \
        set_error("null engine handle"); \
        return AZ_ERR_ARG;               \
    }                                    \
    AZ_HIP(hipSetDevice((e)->cfg.device))

extern "C" {

const char* az_last_error(void) { return g_err.c_str(); }
int az_version(void) { return 100; }
int az_state_words(int game) {
    GameInfo gi;
    return game_info(game, gi) ? gi.W : -1;
}
int az_n_actions(int game) {
    GameInfo gi;
    return game_info(game, gi) ? gi.A : -1;
}
int az_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ------------------------------------------------------------------------------------------------ K1 host wrappers
int az_env_step_batch(int game, int device, int64_t n, const uint64_t* states_in, const int32_t* actions,
                      uint64_t* states_out, uint8_t* placed, int8_t* winner, uint8_t* done, uint16_t* legal) {
    GameInfo gi;
    if (!game_info(game, gi) || n < 0 || !states_in) {
        set_error("az_env_step_batch: bad argument");
        return AZ_ERR_ARG;
    }
    AZ_TRY(use_device(device));
    if (n == 0) return AZ_OK;
    TmpBufs t;
    Block in, out;
    const size_t sb = (size_t)n * gi.W * sizeof(u64);
    const size_t i_st = in.reserve(sb), i_act = actions ? in.reserve((size_t)n * 4) : 0;
    const size_t o_st = out.reserve(sb), o_pl = out.reserve(n), o_wn = out.reserve(n), o_dn = out.reserve(n),
                 o_lg = out.reserve((size_t)n * 2);
    AZ_TRY(in.alloc(t));
    AZ_TRY(out.alloc(t));
    in.put(i_st, states_in, sb);
    if (actions) in.put(i_act, actions, (size_t)n * 4);
    AZ_HIP(hipMemcpy(in.dev, in.host.data(), in.size, hipMemcpyHostToDevice));
    AZ_TRY(launch_env_step(game, n, in.d<u64>(i_st), actions ? in.d<i32>(i_act) : nullptr, out.d<u64>(o_st), out.d<u8>(o_pl),
                           out.d<i8>(o_wn), out.d<u8>(o_dn), out.d<u16>(o_lg), 0));
    AZ_HIP(hipMemcpy(out.host.data(), out.dev, out.size, hipMemcpyDeviceToHost));   // on the null stream: after the kernel
    out.take(o_st, states_out, sb);
    out.take(o_pl, placed, n);
    out.take(o_wn, winner, n);
    out.take(o_dn, done, n);
    out.take(o_lg, legal, (size_t)n * 2);
    return AZ_OK;
}

int az_env_status_batch(int game, int device, int64_t n, const uint64_t* states, int8_t* winner, uint8_t* done,
                        uint16_t* legal, uint8_t* full) {
    GameInfo gi;
    if (!game_info(game, gi) || n < 0 || !states) {
        set_error("az_env_status_batch: bad argument");
        return AZ_ERR_ARG;
    }
    AZ_TRY(use_device(device));
    if (n == 0) return AZ_OK;
    TmpBufs t;
    Block in, out;
    const size_t sb = (size_t)n * gi.W * sizeof(u64);
    const size_t i_st = in.reserve(sb);
    const size_t o_wn = out.reserve(n), o_dn = out.reserve(n), o_lg = out.reserve((size_t)n * 2), o_fl = out.reserve(n);
    AZ_TRY(in.alloc(t));
    AZ_TRY(out.alloc(t));
    in.put(i_st, states, sb);
    AZ_HIP(hipMemcpy(in.dev, in.host.data(), in.size, hipMemcpyHostToDevice));
    AZ_TRY(launch_env_status(game, n, in.d<u64>(i_st), out.d<i8>(o_wn), out.d<u8>(o_dn), out.d<u16>(o_lg), out.d<u8>(o_fl), 0));
    AZ_HIP(hipMemcpy(out.host.data(), out.dev, out.size, hipMemcpyDeviceToHost));
    out.take(o_wn, winner, n);
    out.take(o_dn, done, n);
    out.take(o_lg, legal, (size_t)n * 2);
    out.take(o_fl, full, n);
    return AZ_OK;
}

int az_env_planes_batch(int game, int device, int64_t n, const uint64_t* states, float* planes) {
    GameInfo gi;
    if (!game_info(game, gi) || n < 0 || !states || !planes) {
        set_error("az_env_planes_batch: bad argument");
        return AZ_ERR_ARG;
    }
    AZ_TRY(use_device(device));
    if (n == 0) return AZ_OK;
    size_t per = game == AZ_GAME_C4 ? 126 : (game == AZ_GAME_TTT ? 27 : 252);
    TmpBufs t;
    u64* din;
    float* dpl;
    AZ_TRY(t.get(&din, n * gi.W));
    AZ_TRY(t.get(&dpl, n * per));
    AZ_HIP(hipMemcpy(din, states, n * gi.W * sizeof(u64), hipMemcpyHostToDevice));
    AZ_TRY(launch_env_planes(game, n, din, dpl, 0));
    AZ_HIP(hipDeviceSynchronize());
    AZ_HIP(hipMemcpy(planes, dpl, n * per * sizeof(float), hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_env_flip_batch(int game, int device, int64_t n, const uint64_t* states, uint64_t* flipped, int8_t* flip_action) {
    GameInfo gi;
    if (!game_info(game, gi) || n < 0 || !states) {
        set_error("az_env_flip_batch: bad argument");
        return AZ_ERR_ARG;
    }
    AZ_TRY(use_device(device));
    if (n == 0) return AZ_OK;
    TmpBufs t;
    u64 *din, *dout;
    i8* dfa;
    AZ_TRY(t.get(&din, n * gi.W));
    AZ_TRY(t.get(&dout, n * gi.W));
    AZ_TRY(t.get(&dfa, n * gi.A));
    AZ_HIP(hipMemcpy(din, states, n * gi.W * sizeof(u64), hipMemcpyHostToDevice));
    AZ_TRY(launch_env_flip(game, n, din, dout, dfa, 0));
    AZ_HIP(hipDeviceSynchronize());
    if (flipped) AZ_HIP(hipMemcpy(flipped, dout, n * gi.W * sizeof(u64), hipMemcpyDeviceToHost));
    if (flip_action) AZ_HIP(hipMemcpy(flip_action, dfa, n * gi.A, hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_env_step_batch_dev(int game, int64_t n, const uint64_t* states_in_dev, const int32_t* actions_dev,
                          uint64_t* states_out_dev, uint8_t* placed_dev, int8_t* winner_dev, uint8_t* done_dev,
                          uint16_t* legal_dev) {
    return launch_env_step(game, n, states_in_dev, actions_dev, states_out_dev, placed_dev, winner_dev, done_dev,
                           legal_dev, 0);
}

int az_puct_batch(int device, double c_init, int32_t c_base, int64_t n, const int32_t* Np, const int32_t* nv,
                  const float* Q, const float* P, float* score) {
    if (n < 0 || !Np || !nv || !Q || !P || !score) {
        set_error("az_puct_batch: bad argument");
        return AZ_ERR_ARG;
    }
    AZ_TRY(use_device(device));
    if (n == 0) return AZ_OK;
    int maxnp = 0;
    for (int64_t i = 0; i < n; i++)
        if (Np[i] > maxnp) maxnp = Np[i];
    std::vector<float2> tab;
    build_puct_table(c_init, c_base, maxnp + 1, tab);
    TmpBufs t;
    i32 *dNp, *dnv;
    float *dQ, *dP, *ds;
    float2* dtab;
    AZ_TRY(t.get(&dNp, n));
    AZ_TRY(t.get(&dnv, n));
    AZ_TRY(t.get(&dQ, n));
    AZ_TRY(t.get(&dP, n));
    AZ_TRY(t.get(&ds, n));
    AZ_TRY(t.get(&dtab, tab.size()));
    AZ_HIP(hipMemcpy(dNp, Np, n * 4, hipMemcpyHostToDevice));
    AZ_HIP(hipMemcpy(dnv, nv, n * 4, hipMemcpyHostToDevice));
    AZ_HIP(hipMemcpy(dQ, Q, n * 4, hipMemcpyHostToDevice));
    AZ_HIP(hipMemcpy(dP, P, n * 4, hipMemcpyHostToDevice));
    AZ_HIP(hipMemcpy(dtab, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
    AZ_TRY(launch_puct(c_init, c_base, n, dNp, dnv, dQ, dP, dtab, (i32)tab.size(), ds, 0));
    AZ_HIP(hipDeviceSynchronize());
    AZ_HIP(hipMemcpy(score, ds, n * 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

// ------------------------------------------------------------------------------------------------ engine
int az_engine_create(const az_config* cfg, az_engine** out) {
    if (!cfg || !out) {
        set_error("az_engine_create: null argument");
        return AZ_ERR_ARG;
    }
    GameInfo gi;
    if (!game_info(cfg->game, gi) || cfg->n_games <= 0 || cfg->n_playout <= 0) {
        set_error("az_engine_create: bad game / n_games / n_playout");
        return AZ_ERR_ARG;
    }
    if (cfg->evaluator == AZ_EVAL_CNN && cfg->game != AZ_GAME_C4) {
        set_error("AZ_EVAL_CNN exists for Connect4 only (the reference has no usable TTT/NBTTT checkpoint)");
        return AZ_ERR_ARG;
    }
    AZ_TRY(use_device(cfg->device));
    // the staging memory of az_net_forward_static is looked up here, outside that entry point's hot path; an engine that
    // never calls it is not refused over it (a device ordinal beyond the table: az_net_forward_static returns AZ_ERR_STATE)
    (void)static_stage_init(cfg->device);
    az_engine* e = new az_engine();
    e->cfg = *cfg;
    e->gi = gi;
    if (e->cfg.c_base <= 0) e->cfg.c_base = 500;
    int max_plies = cfg->max_plies > 0 ? cfg->max_plies : gi.MAX_PLIES;
    if (max_plies > gi.MAX_PLIES) max_plies = gi.MAX_PLIES;
    i64 cap = cfg->nodes_per_game;
    if (cap <= 0) cap = 1 + (i64)gi.A * cfg->n_playout * (cfg->reuse_tree ? max_plies : 1);
    if (cap > (i64)LINK_FIRST_MASK) cap = LINK_FIRST_MASK;
    const i64 G = cfg->n_games;
    Dev& d = e->d;
    memset(&d, 0, sizeof(d));
    memset(&e->p, 0, sizeof(e->p));
    d.game = cfg->game;
    d.G = (i32)G;
    d.g0 = 0;
    d.gn = (i32)G;
    d.S = cfg->n_playout;
    d.A = gi.A;
    d.W = gi.W;
    d.MAXD = gi.MAXD;
    d.cap = cap;
    d.c_init = cfg->c_init;
    d.c_base = e->cfg.c_base;
    d.evaluator = cfg->evaluator;
    d.eval_seed = cfg->eval_seed;
    d.flip_mode = gi.LG && cfg->game != AZ_GAME_NBTTT ? cfg->flip_mode : AZ_FLIP_NEVER;
    d.rng_seed = cfg->rng_seed;
    d.gid_stride = cfg->game_id_stride ? cfg->game_id_stride : 1ull;
    e->cfg.game_id_stride = d.gid_stride;
    d.noise_alpha = cfg->dirichlet_alpha > 0 ? (float)cfg->dirichlet_alpha : 0.f;
    int rc = AZ_OK;
#define A_(ptr, n) if (rc == AZ_OK) rc = e->dalloc(&(ptr), (size_t)(n))
    A_(d.nodes, G * cap);
    A_(d.root_state, G * gi.W);
    A_(d.root, G);
    A_(d.n_alloc, G);
    A_(d.active, G);
    A_(d.flip_pos, G);
    A_(d.sim_ctr, G);
    A_(d.move_sims, G);
    A_(d.game_id, G);
    A_(d.path_idx, G * gi.MAXD);
    A_(d.path_N, G * gi.MAXD);
    A_(d.path_Q, G * gi.MAXD);
    A_(d.leaf_depth, G);
    A_(d.leaf_flags, G);
    A_(d.leaf_tvalue, G);
    // speculative rows (az_tree.hip "Speculation"): engines of up to a few hundred games whose leaves go through K3 and its
    // evaluation cache -- first of all the drop-in player's single game.  2 kinds x LG lanes x 2 orientations per slot.
    {
        static const char* spec_env = getenv("AZ_SPEC");   // "0": off (A/B)
        const bool cache_wanted = cfg->eval_cache_entries >= 0 && !(getenv("AZ_EVAL_CACHE") && atoi(getenv("AZ_EVAL_CACHE")) == 0);
        // ms per move of az_search x 500 playouts, lockstep chain against rolling form with speculation (tools/single_game_rate.py): 1
        // game 16.7 -> 6.9, 8 games 25.1 -> 13.8, 16: 26.1 -> 15.1, 32: 29.2 -> 16.5, 64: 30.9 -> 17.7, 128: 36.8 -> 20.9, 256: 37.8 -> 25.2,
        // 512: 38.0 -> 31.5, 1 024: 40.4 -> 46.0 (a K3 launch carries up to 33 rows per game: from a thousand games on the extra rows
        // cost more CU time than the rounds they save), 2 048: 84.7 -> 83.5
        static const char* specg_env = getenv("AZ_SPEC_MAX_G");   // measurement hook
        const int spec_max_g = specg_env ? atoi(specg_env) : 1024;
        d.spec_rows = (cfg->evaluator == AZ_EVAL_CNN && cfg->game == AZ_GAME_C4 && G <= spec_max_g && cache_wanted &&
                       !(spec_env && atoi(spec_env) == 0)) ? 32 : 0;
        d.spec_mark = nullptr;
        d.spec_tag = 1;
    }
    const i64 rows = (i64)G * (1 + d.spec_rows);
    A_(d.leaf_state, rows * gi.W);
    A_(d.eval_probs, rows * gi.A);
    A_(d.eval_value, rows);
    A_(d.eval_list, rows);
    if (d.spec_rows) {
        A_(d.spec_mark, (i64)G * d.spec_rows);
        if (rc == AZ_OK) (void)hipMemset(d.spec_mark, 0, (size_t)G * d.spec_rows);
    }
    A_(d.eval_count, 4 * AZ_MAX_COHORTS);
    A_(d.winner, G);
    A_(d.done, G);
    A_(d.ply, G);
    A_(d.stats, ST_N);
    A_(d.err, 4);
    A_(e->d_actions, G);
    A_(e->d_pi, G * gi.A);
    A_(e->d_temps, G);
    A_(e->rc_actions, G * gi.A);
    A_(e->rc_visits, G * gi.A);
    A_(e->rc_q, G * gi.A);
    A_(e->rc_p, G * gi.A);
    A_(e->rc_n, G);
    A_(e->rc_rootN, G);
    A_(e->rc_rootQ, G);
    A_(e->d_flags8, G);
    Play& p = e->p;
    A_(p.tj_state, G * gi.MAX_PLIES * gi.W);
    A_(p.tj_pi, G * gi.MAX_PLIES * gi.A);
    A_(p.tj_action, G * gi.MAX_PLIES);
    A_(p.tj_player, G * gi.MAX_PLIES);
    p.out_cap = G * gi.MAX_PLIES * 2;
    A_(p.out_state, p.out_cap * gi.W);
    A_(p.out_pi, p.out_cap * gi.A);
    A_(p.out_z, p.out_cap);
    A_(p.out_player, p.out_cap);
    A_(p.out_action, p.out_cap);
    A_(p.out_done, p.out_cap);
    A_(p.out_gid, p.out_cap);
    A_(p.out_count, 1);
    A_(p.out_games, 1);
    A_(p.games_started, 1);
    // PUCT table: root visit counts can reach S*(plies+1) with tree reuse.
    int tab_n = cfg->n_playout * (max_plies + 2) + 64;
    if (tab_n < 65536) tab_n = 65536;
    std::vector<float2> tab;
    build_puct_table(cfg->c_init, e->cfg.c_base, tab_n, tab);
    float2* dtab = nullptr;
    A_(dtab, tab_n);
#undef A_
    if (rc != AZ_OK) {
        az_engine_destroy(e);
        return rc;
    }
    d.tab = dtab;
    d.tab_n = tab_n;
    if (cfg->evaluator == AZ_EVAL_ROLLOUT || cfg->evaluator == AZ_EVAL_ROLLOUT_HASH) {
        // TreeNode.UCT (MCTS.py:57-63) is float64 in the reference; both logarithms come from glibc on the host
        std::vector<double2> t64((size_t)tab_n);
        for (int n = 0; n < tab_n; n++)
            t64[n] = make_double2(cfg->c_init + log((double)(1 + n + e->cfg.c_base) / (double)e->cfg.c_base),
                                  n > 0 ? log((double)n) : 0.0);
        double2* dt64 = nullptr;
        int rc64 = e->dalloc(&dt64, (size_t)tab_n);
        if (rc64 != AZ_OK) {
            az_engine_destroy(e);
            return rc64;
        }
        (void)hipMemcpy(dt64, t64.data(), t64.size() * sizeof(double2), hipMemcpyHostToDevice);
        d.tab64 = dt64;
    }
    // evaluation cache (az_tree.hip "Evaluation cache"): AZ_EVAL_CNN engines, unless switched off.  Auto size: 128 Ki entries
    // (8 MB) per slot, 2^16 .. 2^30 entries of 64 B -- 17 GB for 2 048 slots, 68 GB for 16 384: a game evaluates ~8 k distinct
    // positions and what pays is the history of MANY generations of games (measured, games/s at 2^26 / 2^28 / 2^30 entries:
    // 2 048 slots 2 744 / 2 867 / 2 841, 16 384 slots 4 917 / 6 502 / 7 414; 2^31: 7 489); the board has 288 GB
    e->cfg.eval_cache_entries = -1;
    {
        static const char* cenv = getenv("AZ_EVAL_CACHE");   // "0": off whatever the configuration says (A/B runs); "n": 2^n entries
        i64 want = cfg->eval_cache_entries;
        if (cenv && atoi(cenv) == 0) want = -1;
        else if (cenv && atoi(cenv) > 0 && want == 0) want = 1ll << std::min(atoi(cenv), 31);
        if (cfg->evaluator == AZ_EVAL_CNN && want >= 0) {
            const bool auto_size = want == 0;
            // (with speculative rows thirteen positions enter the table for every one a game asks for: 2^22 entries at least)
            if (auto_size) want = std::min<i64>(std::max<i64>(G * 131072, d.spec_rows ? 1ll << 22 : 1ll << 16), 1ll << 30);
            i64 n = 1;
            while (n < want && n < (1ll << 31)) n <<= 1;
            uint4* tab = nullptr;
            int rcc = AZ_OK;
            if (auto_size) {
                // the table is an optimisation, never a reason to refuse an engine: an auto-sized one takes at most half of the
                // memory that is free right now (other engines, the caller's own tensors) and halves again if hipMalloc says no;
                // a size the caller asked for is allocated as asked or fails loudly
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess)
                    while (n > (1ll << 16) && (size_t)n * 64 > fr / 2) n >>= 1;
                while ((rcc = e->dalloc(&tab, (size_t)n * 4)) != AZ_OK && n > (1ll << 16)) {
                    (void)hipGetLastError();
                    n >>= 1;
                }
            } else {
                rcc = e->dalloc(&tab, (size_t)n * 4);
            }
            // zeroed: hipMalloc may hand back a table of an engine that was destroyed earlier in this process
            if (rcc == AZ_OK && hipMemset(tab, 0, (size_t)n * 64) != hipSuccess) {
                set_error("hipMemset of the evaluation cache failed");
                rcc = AZ_ERR_HIP;
            }
            if (rcc != AZ_OK) {
                az_engine_destroy(e);
                return rcc;
            }
            d.cache = tab;
            d.cache_mask = (u64)n - 1;
            e->cfg.eval_cache_entries = n;
        }
    }
    hipError_t he = hipStreamCreate(&e->stream);
    // cohorts only pay off where an evaluator launch sits between select and backup (AZ_EVAL_CNN)
    int C = cfg->n_cohorts;
    // measured: without the evaluation cache 3 cohorts at 2 048 slots (K3 launches fill the chip), with it 4
    // (K3 launches are a third of the slots: 1 / 2 / 3 / 4 cohorts -> 1 518 / 1 870 / 2 020 / 2 098 games/s; 6 and 8: 1 927, 1 904)
    if (C <= 0) C = cfg->evaluator != AZ_EVAL_CNN ? 1 : (G >= (d.cache ? 2048 : 4096) ? 4 : G >= 768 ? 3 : (G >= 512 ? 2 : 1));
    if (C > AZ_MAX_COHORTS) C = AZ_MAX_COHORTS;
    if (cfg->evaluator != AZ_EVAL_CNN || C > G) C = 1;
    e->C = C;
    e->cfg.n_cohorts = C;
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
    for (int c = 1; c < C && he == hipSuccess; c++) {
        he = hipStreamCreate(&e->cstream[c]);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_join[c], hipEventDisableTiming);
    }
    if (he != hipSuccess) {
        set_error("hipStreamCreate / hipEventCreate failed");
        az_engine_destroy(e);
        return AZ_ERR_HIP;
    }
    (void)hipMemcpy(dtab, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice);
    (void)hipMemset(d.stats, 0, ST_N * sizeof(unsigned long long));
    (void)hipMemset(d.err, 0, 16);
    (void)hipMemset(d.eval_count, 0, 16 * AZ_MAX_COHORTS);
    (void)hipMemset(p.out_count, 0, 8);
    (void)hipMemset(p.out_games, 0, 8);
    (void)hipMemset(p.games_started, 0, 8);
    (void)hipMemset(d.root_state, 0, G * gi.W * sizeof(u64));
    *out = e;
    rc = az_reset_games(e);
    if (rc != AZ_OK) {
        az_engine_destroy(e);
        *out = nullptr;
        return rc;
    }
    return AZ_OK;
}

void az_engine_destroy(az_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    for (int c = 0; c < AZ_MAX_COHORTS; c++) {
        if (e->cstream[c]) {
            (void)hipStreamSynchronize(e->cstream[c]);
            (void)hipStreamDestroy(e->cstream[c]);
        }
        if (e->ev_join[c]) (void)hipEventDestroy(e->ev_join[c]);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->prof_base) (void)hipEventDestroy(e->prof_base);
    if (e->stream) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamDestroy(e->stream);
    }
    if (e->net) cnn_destroy(e->net);
    for (auto& ev : e->prof_ev) (void)hipEventDestroy(ev);
    for (void* q : e->allocs) (void)hipFree(q);
    if (e->flip_bits) (void)hipFree(e->flip_bits);
    delete e;
}

int az_engine_sync(az_engine* e) {
    AZ_CHECK_E(e);
    AZ_HIP(hipStreamSynchronize(e->stream));
    u32 err = 0;
    AZ_HIP(hipMemcpy(&err, e->d.err, sizeof(u32), hipMemcpyDeviceToHost));
    if (err & 5u) AZ_HIP(hipMemset(e->d.err, 0, sizeof(u32)));  // reported once: the trees that fit stay readable
    if (err & 1u) {
        set_error("node arena overflow: raise az_config.nodes_per_game / max_plies");
        return AZ_ERR_ARENA;
    }
    if (err & 4u) {
        set_error("trajectory buffer overflow: fetch trajectories more often");
        return AZ_ERR_ARENA;
    }
    return AZ_OK;
}

int az_get_config(az_engine* e, az_config* out) {
    AZ_CHECK_E(e);
    if (!out) return bad_arg("az_get_config");
    *out = e->cfg;
    out->nodes_per_game = e->d.cap;
    return AZ_OK;
}

int az_get_stats(az_engine* e, az_stats* out) {
    AZ_CHECK_E(e);
    if (!out) return bad_arg("az_get_stats");
    AZ_HIP(hipStreamSynchronize(e->stream));
    unsigned long long st[ST_N];
    AZ_HIP(hipMemcpy(st, e->d.stats, sizeof(st), hipMemcpyDeviceToHost));
    std::vector<u32> na(e->d.G);
    AZ_HIP(hipMemcpy(na.data(), e->d.n_alloc, e->d.G * sizeof(u32), hipMemcpyDeviceToHost));
    i64 used = 0;
    for (u32 v : na) used += v;
    out->playouts = (i64)st[ST_PLAYOUTS];
    out->levels = (i64)st[ST_LEVELS];
    out->leaf_evals = (i64)st[ST_EVALS];
    out->terminal_hits = (i64)st[ST_TERMINAL];
    out->children = (i64)st[ST_CHILDREN];
    out->nodes_used = used;
    out->games_finished = (i64)st[ST_GAMES];
    out->plies = (i64)st[ST_PLIES];
    out->rollout_plies = (i64)st[ST_ROLLOUT_PLIES];
    out->cache_hits = (i64)st[ST_CACHE_HITS];
    return AZ_OK;
}

int az_set_profiling(az_engine* e, int on) {
    AZ_CHECK_E(e);
    AZ_HIP(hipStreamSynchronize(e->stream));
    AZ_TRY(prof_harvest(e));
    if (on && e->prof_ev.empty()) {
        const size_t pairs = 131072;   // a whole bench window of K3 launches: the read-back happens after the timed region
        e->prof_ev.resize(2 * pairs);
        e->prof_cat.resize(pairs);
        for (auto& ev : e->prof_ev) AZ_HIP(hipEventCreate(&ev));
    }
    e->prof_on = (on == 1 || on == 3 || on == 4) ? 0xF : (on == 2 ? (1 << PROF_CNN) : 0);   // category mask
    e->prof_sample0 = on == 3;
    e->prof_blocks = on == 4;
    e->prof_round = 0;
    for (int i = 0; i < 4; i++) {
        e->prof_ms[i] = 0;
        e->prof_n[i] = 0;
        e->prof_busy[i] = 0;
        e->prof_last_end[i] = 0;
    }
    if (on) {
        if (!e->prof_base) AZ_HIP(hipEventCreate(&e->prof_base));
        AZ_HIP(hipEventRecord(e->prof_base, e->stream));
        AZ_HIP(hipStreamSynchronize(e->stream));
    }
    return AZ_OK;
}

int az_get_profile(az_engine* e, double* ms /*[4]*/, int64_t* launches /*[4]*/) {
    AZ_CHECK_E(e);
    AZ_HIP(hipStreamSynchronize(e->stream));
    AZ_TRY(prof_harvest(e));
    for (int i = 0; i < 4; i++) {
        if (ms) ms[i] = e->prof_ms[i];
        if (launches) launches[i] = e->prof_n[i];
    }
    return AZ_OK;
}

int az_get_profile_busy(az_engine* e, double* busy_ms /*[4]*/) {
    AZ_CHECK_E(e);
    if (!busy_ms) return bad_arg("az_get_profile_busy");
    AZ_HIP(hipStreamSynchronize(e->stream));
    AZ_TRY(prof_harvest(e));
    for (int i = 0; i < 4; i++) busy_ms[i] = e->prof_busy[i];
    return AZ_OK;
}

int az_reset_stats(az_engine* e) {
    AZ_CHECK_E(e);
    AZ_HIP(hipMemsetAsync(e->d.stats, 0, ST_N * sizeof(unsigned long long), e->stream));
    return AZ_OK;
}

int az_set_noise(az_engine* e, double alpha) {
    AZ_CHECK_E(e);
    e->d.noise_alpha = alpha > 0 ? (float)alpha : 0.f;
    e->cfg.dirichlet_alpha = alpha;
    return AZ_OK;
}

int az_set_n_playout(az_engine* e, int32_t n) {
    AZ_CHECK_E(e);
    if (n <= 0) return bad_arg("az_set_n_playout");
    e->cfg.n_playout = n;
    e->d.S = n;
    return AZ_OK;
}

// Grow the committed-trajectory arrays to hold `need` plies (only while they are empty).
static int ensure_out_cap(az_engine* e, i64 need) {
    Play& p = e->p;
    if (need <= p.out_cap) return AZ_OK;
    AZ_HIP(hipStreamSynchronize(e->stream));
    unsigned long long cnt = 0;
    AZ_HIP(hipMemcpy(&cnt, p.out_count, 8, hipMemcpyDeviceToHost));
    if (cnt != 0) {
        set_error("trajectory buffer not empty: az_traj_fetch(clear) before asking for more games");
        return AZ_ERR_STATE;
    }
    const GameInfo& gi = e->gi;
    int rc = AZ_OK;
    rc = e->dalloc(&p.out_state, (size_t)need * gi.W);
    if (rc == AZ_OK) rc = e->dalloc(&p.out_pi, (size_t)need * gi.A);
    if (rc == AZ_OK) rc = e->dalloc(&p.out_z, (size_t)need);
    if (rc == AZ_OK) rc = e->dalloc(&p.out_player, (size_t)need);
    if (rc == AZ_OK) rc = e->dalloc(&p.out_action, (size_t)need);
    if (rc == AZ_OK) rc = e->dalloc(&p.out_done, (size_t)need);
    if (rc == AZ_OK) rc = e->dalloc(&p.out_gid, (size_t)need);
    if (rc == AZ_OK) p.out_cap = need;  // the smaller old arrays stay owned by the engine until it is destroyed
    return rc;
}

int az_reset_games(az_engine* e) {
    AZ_CHECK_E(e);
    const Dev& d = e->d;
    e->roll_started = false;
    e->roll_par = 0;
    AZ_HIP(hipMemsetAsync(d.eval_count, 0, 16 * AZ_MAX_COHORTS, e->stream));
    // leaves that were waiting for the network are dropped with their lists: so are the speculative rows that waited with them
    if (d.spec_rows) AZ_HIP(hipMemsetAsync(d.spec_mark, 0, (size_t)d.G * d.spec_rows, e->stream));
    AZ_HIP(hipMemsetAsync(d.err, 0, sizeof(u32), e->stream));
    AZ_HIP(hipMemsetAsync(d.root_state, 0, (size_t)d.G * d.W * sizeof(u64), e->stream));
    AZ_TRY(launch_init_slots(d, e->cfg.game == AZ_GAME_NBTTT ? 1 : 0, e->cfg.game_id_base, e->stream));
    unsigned long long g = (unsigned long long)d.G;
    AZ_HIP(hipMemcpyAsync(e->p.games_started, &g, 8, hipMemcpyHostToDevice, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

int az_set_roots(az_engine* e, const uint64_t* states) {
    AZ_CHECK_E(e);
    if (!states) return bad_arg("az_set_roots");
    const Dev& d = e->d;
    e->roll_started = false;
    e->roll_par = 0;
    AZ_HIP(hipMemsetAsync(d.eval_count, 0, 16 * AZ_MAX_COHORTS, e->stream));
    // leaves that were waiting for the network are dropped with their lists: so are the speculative rows that waited with them
    if (d.spec_rows) AZ_HIP(hipMemsetAsync(d.spec_mark, 0, (size_t)d.G * d.spec_rows, e->stream));
    AZ_HIP(hipMemcpyAsync(d.root_state, states, (size_t)d.G * d.W * sizeof(u64), hipMemcpyHostToDevice, e->stream));
    AZ_TRY(launch_init_slots(d, 0, e->cfg.game_id_base, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

int az_get_roots(az_engine* e, uint64_t* states, int8_t* winner, uint8_t* done) {
    AZ_CHECK_E(e);
    const Dev& d = e->d;
    AZ_HIP(hipStreamSynchronize(e->stream));
    if (states) AZ_HIP(hipMemcpy(states, d.root_state, (size_t)d.G * d.W * sizeof(u64), hipMemcpyDeviceToHost));
    if (winner) AZ_HIP(hipMemcpy(winner, d.winner, d.G, hipMemcpyDeviceToHost));
    if (done) AZ_HIP(hipMemcpy(done, d.done, d.G, hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_set_flip_stream(az_engine* e, const uint8_t* bits, int64_t n_per_game) {
    AZ_CHECK_E(e);
    if (!bits || n_per_game <= 0) return bad_arg("az_set_flip_stream");
    AZ_HIP(hipStreamSynchronize(e->stream));
    if (e->flip_bits) (void)hipFree(e->flip_bits);
    e->flip_bits = nullptr;
    size_t bytes = (size_t)e->d.G * n_per_game;
    AZ_HIP(hipMalloc((void**)&e->flip_bits, bytes));
    AZ_HIP(hipMemcpy(e->flip_bits, bits, bytes, hipMemcpyHostToDevice));
    AZ_HIP(hipMemset(e->d.flip_pos, 0, e->d.G * sizeof(u32)));
    e->d.flip_bits = e->flip_bits;
    e->d.flip_n = n_per_game;
    return AZ_OK;
}

// The engine's slots cut into its C cohorts: per cohort a view of the device state for each of the two leaf counters that
// successive rounds alternate between, and the stream its chain runs on (cohort 0 rides on the engine stream: a process has
// few hardware queues).
static void cohort_views(const az_engine* e, Dev (*dc)[2], hipStream_t* cs, bool speculating = false) {
    const Dev& d = e->d;
    const int C = e->C;
    for (int c = 0; c < C; c++) {
        const i64 lo = (i64)d.G * c / C, hi = (i64)d.G * (c + 1) / C;
        for (int par = 0; par < 2; par++) {
            dc[c][par] = d;
            dc[c][par].g0 = (i32)lo;
            dc[c][par].gn = (i32)(hi - lo);
            dc[c][par].eval_count = d.eval_count + 4 * c + par;
            dc[c][par].eval_list = d.eval_list + lo * (1 + d.spec_rows);   // a cohort's list holds its slots' rows and their speculative rows
            // only az_search's rolling form speculates: everywhere else a K3 launch is sized for the slots' own rows
            if (!speculating) dc[c][par].spec_rows = 0;
        }
        cs[c] = c == 0 ? e->stream : e->cstream[c];
    }
}

// az_search for AZ_EVAL_CNN engines on the rolling tree step (az_tree.hip, Roll::search): every game runs its n_sims playouts one
// after the other exactly as in the lockstep chain below, but a playout that ends on a terminal node, or on a position whose
// evaluation is in the cache, is finished in place and the game walks on -- a round of { K3 ; tree step } is spent only on
// leaves the network really has to see.  The lockstep chain spends one round on EVERY playout (500 rounds per move; late in a
// game 20-60 % of them carry no leaf at all).  The number of rounds a search needs is not known ahead, so the host reads the
// two counters every few rounds; the call returns when every game that takes part has had its n_sims playouts.  Used for
// engines of a handful of games (az_search); both forms grow the same trees bit for bit (tests/test_gpu_round3.py).
static int search_cnn_rolling(az_engine* e, int n_sims) {
    const Dev& d = e->d;
    const int C = e->C;
    Dev dc[AZ_MAX_COHORTS][2];
    hipStream_t cs[AZ_MAX_COHORTS];
    cohort_views(e, dc, cs, true);
    if (!e->search_cnt) AZ_TRY(e->dalloc(&e->search_cnt, 2));
    AZ_HIP(hipMemsetAsync(e->search_cnt, 0, 2 * sizeof(u32), e->stream));
    if (d.spec_rows) AZ_HIP(hipMemsetAsync(d.spec_mark, 0, (size_t)d.G * d.spec_rows, e->stream));
    AZ_HIP(hipMemsetAsync(d.move_sims, 0, (size_t)d.G * sizeof(i32), e->stream));
    auto fork = [&]() -> int {
        if (C > 1) {
            AZ_HIP(hipEventRecord(e->ev_fork, e->stream));
            for (int c = 1; c < C; c++) AZ_HIP(hipStreamWaitEvent(cs[c], e->ev_fork, 0));
        }
        return AZ_OK;
    };
    auto join = [&]() -> int {
        for (int c = 1; c < C; c++) {
            AZ_HIP(hipEventRecord(e->ev_join[c], cs[c]));
            AZ_HIP(hipStreamWaitEvent(e->stream, e->ev_join[c], 0));
        }
        return AZ_OK;
    };
    auto roll = [&](const Dev& dd, int mode, u32* stale, hipStream_t st) {
        return launch_roll_step(dd, mode, stale, e->p, n_sims, 1.0, -1, e->cfg.move_mode, e->cfg.reuse_tree, 0ull,
                                e->cfg.game_id_base, st, e->search_cnt);
    };
    AZ_TRY(fork());
    for (int c = 0; c < C; c++) {
        AZ_TRY(prof_begin(e, PROF_SELECT, cs[c]));
        AZ_TRY(roll(dc[c][0], 2, nullptr, cs[c]));
        AZ_TRY(prof_end(e, cs[c]));
    }
    int par = 0, rounds = 0;
    u32 cnt[2] = {1u, 0u};
    // rounds between two looks at the counters: a look drains the streams (a few tens of microseconds)
    int chunk = std::max(4, std::min(32, n_sims / 8));
    while (cnt[1] < cnt[0]) {
        if (rounds > n_sims + 64) {   // every playout a miss is n_sims rounds: more than that is a bug, not a long search
            set_error("az_search: the rolling search did not complete");
            return AZ_ERR_STATE;
        }
        for (int k = 0; k < chunk; k++, rounds++) {
            for (int c = 0; c < C; c++) {
                AZ_TRY(prof_begin(e, PROF_CNN, cs[c]));
                AZ_TRY(cnn_eval_leaves(e->net, dc[c][par], cs[c]));
                AZ_TRY(prof_end(e, cs[c]));
                AZ_TRY(prof_begin(e, PROF_BACKUP, cs[c]));
                AZ_TRY(roll(dc[c][par ^ 1], 3, dc[c][par].eval_count, cs[c]));
                AZ_TRY(prof_end(e, cs[c]));
            }
            par ^= 1;
        }
        AZ_TRY(join());
        AZ_HIP(hipMemcpyAsync(cnt, e->search_cnt, sizeof(cnt), hipMemcpyDeviceToHost, e->stream));
        AZ_HIP(hipStreamSynchronize(e->stream));
        if (cnt[1] < cnt[0]) AZ_TRY(fork());
    }
    // nothing is parked now (a complete game selects no further leaf) and both leaf counters are zero; the per-move counter goes
    // back to zero for the rolling self-play entry points
    AZ_HIP(hipMemsetAsync(d.move_sims, 0, (size_t)d.G * sizeof(i32), e->stream));
    return AZ_OK;
}

int az_search(az_engine* e, int32_t n_sims) {
    AZ_CHECK_E(e);
    if (n_sims <= 0) n_sims = e->cfg.n_playout;
    const Dev& d = e->d;
    if (d.flip_mode == AZ_FLIP_STREAM && !d.flip_bits) {
        set_error("AZ_FLIP_STREAM needs az_set_flip_stream first");
        return AZ_ERR_STATE;
    }
    if (d.evaluator == AZ_EVAL_UNIFORM || d.evaluator == AZ_EVAL_HASH) {
        AZ_TRY(prof_begin(e, PROF_FUSED, e->stream));
        AZ_TRY(launch_search_fused(d, n_sims, e->stream));
        return prof_end(e, e->stream);
    }
    if (d.evaluator == AZ_EVAL_ROLLOUT || d.evaluator == AZ_EVAL_ROLLOUT_HASH) {
        AZ_TRY(prof_begin(e, PROF_FUSED, e->stream));
        AZ_TRY(launch_search_pure(d, n_sims, e->stream));
        return prof_end(e, e->stream);
    }
    if (d.evaluator == AZ_EVAL_CNN) {
        if (!e->net) {
            set_error("AZ_EVAL_CNN: call az_net_load first");
            return AZ_ERR_NO_NET;
        }
        // A handful of games (the drop-in player's single game, BASELINE configs[0]), and every engine with speculative rows: the rolling form.  One game, 500 playouts per
        // move, ms per move: first game of an engine 20.4 -> 19.0, a game whose positions the cache already holds 7.95 -> 4.5.  Many
        // games: the lockstep chain -- a search then lasts as long as the game with the most misses whatever the form, and the
        // chain's rounds are shorter (ms per move, chain / rolling: 8 games 26.8 / 25.0, 16: 27.2 / 25.8, 32: 29.6 / 29.2, 64: 30.9 / 31.2,
        // 256: 38.0 / 40.6).  AZ_SEARCH_ROLL=0|1 forces one form (A/B, parity).
        if (e->roll_started) {   // az_selfplay_rounds keeps leaves parked and playout counters running between calls
            set_error("az_search: rolling self-play is in progress on this engine (az_reset_games or az_set_roots first)");
            return AZ_ERR_STATE;
        }
        static const char* sr_env = getenv("AZ_SEARCH_ROLL");
        if (sr_env ? atoi(sr_env) != 0 : (d.G <= 32 || d.spec_rows > 0)) return search_cnn_rolling(e, n_sims);
        // One chain select -> CNN -> backup per cohort and playout.  The chains of different cohorts are independent
        // (disjoint slots, own leaf list and counter) and are issued round-robin onto their own streams.
        const int C = e->C;
        Dev dc[AZ_MAX_COHORTS][2];   // [cohort][leaf-counter parity]
        hipStream_t cs[AZ_MAX_COHORTS];
        cohort_views(e, dc, cs);
        if (C > 1) {  // fork: the cohort streams start after everything queued on the engine stream
            AZ_HIP(hipEventRecord(e->ev_fork, e->stream));
            for (int c = 1; c < C; c++) AZ_HIP(hipStreamWaitEvent(cs[c], e->ev_fork, 0));
        }
        // Per cohort:  select | CNN, backup+select | CNN, backup+select | ... | CNN, backup.  Playout s counts its
        // leaves into counter (s & 1); the fused tree step clears the counter the evaluator has just consumed.
        for (int c = 0; c < C; c++) {
            AZ_TRY(prof_begin(e, PROF_SELECT, cs[c]));
            AZ_TRY(launch_select(dc[c][0], cs[c], false));
            AZ_TRY(prof_end(e, cs[c]));
        }
        for (int s = 0; s < n_sims; s++) {
            const int par = s & 1;
            for (int c = 0; c < C; c++) {
                AZ_TRY(prof_begin(e, PROF_CNN, cs[c]));
                AZ_TRY(cnn_eval_leaves(e->net, dc[c][par], cs[c]));
                AZ_TRY(prof_end(e, cs[c]));
                AZ_TRY(prof_begin(e, PROF_BACKUP, cs[c]));
                if (s + 1 < n_sims) AZ_TRY(launch_backup_select(dc[c][par ^ 1], dc[c][par].eval_count, cs[c]));
                else AZ_TRY(launch_backup(dc[c][par], cs[c]));
                AZ_TRY(prof_end(e, cs[c]));
            }
        }
        if (C > 1) {  // join: later work on the engine stream (move selection, advance) waits for every cohort
            for (int c = 1; c < C; c++) {
                AZ_HIP(hipEventRecord(e->ev_join[c], cs[c]));
                AZ_HIP(hipStreamWaitEvent(e->stream, e->ev_join[c], 0));
            }
        }
        return AZ_OK;
    }
    set_error("az_search: AZ_EVAL_HOST engines are driven with az_select / az_backup");
    return AZ_ERR_STATE;
}

int az_select(az_engine* e, uint64_t* leaf_states, uint8_t* flags) {
    AZ_CHECK_E(e);
    const Dev& d = e->d;
    if (d.flip_mode == AZ_FLIP_STREAM && !d.flip_bits) {
        set_error("AZ_FLIP_STREAM needs az_set_flip_stream first");
        return AZ_ERR_STATE;
    }
    if (d.tab64) {
        set_error("az_select: pure-MCTS engines (AZ_EVAL_ROLLOUT*) are driven with az_search");
        return AZ_ERR_STATE;
    }
    AZ_TRY(launch_select(d, e->stream, true));
    AZ_HIP(hipStreamSynchronize(e->stream));
    if (leaf_states)
        AZ_HIP(hipMemcpy(leaf_states, d.leaf_state, (size_t)d.G * d.W * sizeof(u64), hipMemcpyDeviceToHost));
    if (flags) {
        std::vector<u32> f(d.G);
        AZ_HIP(hipMemcpy(f.data(), d.leaf_flags, d.G * sizeof(u32), hipMemcpyDeviceToHost));
        for (int g = 0; g < d.G; g++)
            flags[g] = (f[g] & 8u) ? (uint8_t)((f[g] & 1u) | (f[g] & 2u)) : (uint8_t)4u;  // bit2: slot idle
    }
    return AZ_OK;
}

int az_backup(az_engine* e, const float* probs, const float* values) {
    AZ_CHECK_E(e);
    if (!probs || !values) return bad_arg("az_backup");
    const Dev& d = e->d;
    AZ_HIP(hipMemcpyAsync(d.eval_probs, probs, (size_t)d.G * d.A * sizeof(float), hipMemcpyHostToDevice, e->stream));
    AZ_HIP(hipMemcpyAsync(d.eval_value, values, (size_t)d.G * sizeof(float), hipMemcpyHostToDevice, e->stream));
    AZ_TRY(launch_backup(d, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

int az_get_root_children(az_engine* e, int32_t* actions, int32_t* visits, float* q, float* prior, int32_t* n_children,
                         int32_t* root_visits, float* root_q) {
    AZ_CHECK_E(e);
    const Dev& d = e->d;
    AZ_TRY(launch_root_children(d, e->rc_actions, e->rc_visits, e->rc_q, e->rc_p, e->rc_n, e->rc_rootN, e->rc_rootQ,
                                e->stream));
    AZ_TRY(az_engine_sync(e));
    size_t ga = (size_t)d.G * d.A;
    if (actions) AZ_HIP(hipMemcpy(actions, e->rc_actions, ga * 4, hipMemcpyDeviceToHost));
    if (visits) AZ_HIP(hipMemcpy(visits, e->rc_visits, ga * 4, hipMemcpyDeviceToHost));
    if (q) AZ_HIP(hipMemcpy(q, e->rc_q, ga * 4, hipMemcpyDeviceToHost));
    if (prior) AZ_HIP(hipMemcpy(prior, e->rc_p, ga * 4, hipMemcpyDeviceToHost));
    if (n_children) AZ_HIP(hipMemcpy(n_children, e->rc_n, d.G * 4, hipMemcpyDeviceToHost));
    if (root_visits) AZ_HIP(hipMemcpy(root_visits, e->rc_rootN, d.G * 4, hipMemcpyDeviceToHost));
    if (root_q) AZ_HIP(hipMemcpy(root_q, e->rc_rootQ, d.G * 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

// test hook: every slot's tree becomes a root with the given children (insertion order), so that move selection can be
// checked on recorded (actions, visits) pairs without a search that happens to produce them
int az_set_root_children(az_engine* e, const int32_t* actions, const int32_t* visits, const int32_t* n_children) {
    AZ_CHECK_E(e);
    if (!actions || !visits || !n_children) return bad_arg("az_set_root_children");
    const Dev& d = e->d;
    if (d.cap < 1 + d.A) {
        set_error("az_set_root_children: the node arena is smaller than a root and its children");
        return AZ_ERR_STATE;
    }
    AZ_HIP(hipStreamSynchronize(e->stream));
    std::vector<uint4> rec((size_t)(1 + d.A));
    std::vector<u32> one(d.G, 0u), na(d.G, (u32)(1 + d.A));
    for (int g = 0; g < d.G; g++) {
        const int k = n_children[g];
        if (k < 0 || k > d.A) return bad_arg("az_set_root_children");
        long long tot = 1;
        for (int i = 0; i < d.A; i++) {
            const int a = i < k ? actions[(size_t)g * d.A + i] : 0, v = i < k ? visits[(size_t)g * d.A + i] : 0;
            if (i < k && (a < 0 || a >= d.A || v < 0)) return bad_arg("az_set_root_children");
            rec[1 + i] = make_uint4((u32)v, 0u, 0u, (u32)a << 28);
            tot += v;
        }
        float pone = 1.0f;
        u32 pbits;
        memcpy(&pbits, &pone, 4);
        rec[0] = make_uint4((u32)tot, 0u, pbits, 1u | ((u32)k << 24));
        AZ_HIP(hipMemcpy(d.nodes + (i64)g * d.cap, rec.data(), rec.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    AZ_HIP(hipMemcpy(d.root, one.data(), d.G * sizeof(u32), hipMemcpyHostToDevice));
    AZ_HIP(hipMemcpy(d.n_alloc, na.data(), d.G * sizeof(u32), hipMemcpyHostToDevice));
    return AZ_OK;
}

int az_choose_moves(az_engine* e, const double* temps, double temp_all, int32_t* actions, float* pi) {
    AZ_CHECK_E(e);
    const Dev& d = e->d;
    if (temps) AZ_HIP(hipMemcpyAsync(e->d_temps, temps, d.G * sizeof(double), hipMemcpyHostToDevice, e->stream));
    AZ_TRY(launch_choose(d, e->p, temps ? e->d_temps : nullptr, temp_all, -1, e->cfg.move_mode, e->d_actions, e->d_pi,
                         e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    if (actions) AZ_HIP(hipMemcpy(actions, e->d_actions, d.G * 4, hipMemcpyDeviceToHost));
    if (pi) AZ_HIP(hipMemcpy(pi, e->d_pi, (size_t)d.G * d.A * 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_advance(az_engine* e, const int32_t* actions) {
    AZ_CHECK_E(e);
    if (!actions) return bad_arg("az_advance");
    const Dev& d = e->d;
    for (int g = 0; g < d.G; g++)  // < 0 = leave the game where it is; the kernels index boards with the rest
        if (actions[g] >= d.A) {
            set_error("az_advance: action " + std::to_string(actions[g]) + " of game " + std::to_string(g) +
                      " is outside [0, " + std::to_string(d.A) + ")");
            return AZ_ERR_ARG;
        }
    AZ_HIP(hipMemcpyAsync(e->d_actions, actions, d.G * 4, hipMemcpyHostToDevice, e->stream));
    AZ_TRY(launch_advance(d, e->p, e->d_actions, e->d_pi, e->cfg.reuse_tree, 0, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

int az_selfplay_step(az_engine* e, double temp, int32_t first_n_steps, int32_t refill) {
    AZ_CHECK_E(e);
    const Dev& d = e->d;
    AZ_TRY(az_search(e, e->cfg.n_playout));
    const unsigned long long target = refill ? (e->target ? e->target : ~0ull) : 0ull;
    return launch_ply_end(d, e->p, temp, first_n_steps, e->cfg.move_mode, e->cfg.reuse_tree, target, e->cfg.game_id_base,
                          e->stream);
}

// Rolling self-play: n_rounds rounds of { CNN on every cohort's parked leaves ; tree step = finish those playouts, end the
// plies that complete (move selection, env.step, prune_root, trajectory, refill) and walk every live game to its next
// NON-terminal leaf }.  See az_tree.hip (Roll).  Leaves stay parked between calls.
int az_selfplay_rounds(az_engine* e, int32_t n_rounds, double temp, int32_t first_n_steps, int32_t refill) {
    AZ_CHECK_E(e);
    const Dev& d = e->d;
    if (d.evaluator != AZ_EVAL_CNN) {
        set_error("az_selfplay_rounds: needs an AZ_EVAL_CNN engine (the synthetic evaluators run the fused search)");
        return AZ_ERR_STATE;
    }
    if (!e->net) {
        set_error("AZ_EVAL_CNN: call az_net_load first");
        return AZ_ERR_NO_NET;
    }
    if (d.flip_mode == AZ_FLIP_STREAM && !d.flip_bits) {
        set_error("AZ_FLIP_STREAM needs az_set_flip_stream first");
        return AZ_ERR_STATE;
    }
    if (n_rounds <= 0) n_rounds = e->cfg.n_playout;
    const unsigned long long target = refill ? (e->target ? e->target : ~0ull) : 0ull;
    const int C = e->C, S = e->cfg.n_playout;
    Dev dc[AZ_MAX_COHORTS][2];
    hipStream_t cs[AZ_MAX_COHORTS];
    // speculative rows (az_tree.hip "Speculation") also in rolling self-play while the chip has CUs to spare (launch_roll_step has
    // the numbers: +50 % and more up to 512 slots, +14 % at 1 024, a loss at 2 048)
    static const char* srg_env = getenv("AZ_SPEC_ROLL_MAX_G");   // measurement hook
    cohort_views(e, dc, cs, d.G <= (srg_env ? atoi(srg_env) : 1024));
    if (C > 1) {
        AZ_HIP(hipEventRecord(e->ev_fork, e->stream));
        for (int c = 1; c < C; c++) AZ_HIP(hipStreamWaitEvent(cs[c], e->ev_fork, 0));
    }
    auto roll = [&](const Dev& dd, int mode, u32* stale, hipStream_t st) {
        return launch_roll_step(dd, mode, stale, e->p, S, temp, first_n_steps, e->cfg.move_mode, e->cfg.reuse_tree, target,
                                e->cfg.game_id_base, st);
    };
    if (!e->roll_started) {
        for (int c = 0; c < C; c++) {
            AZ_TRY(prof_begin(e, PROF_SELECT, cs[c]));
            AZ_TRY(roll(dc[c][e->roll_par], 2, nullptr, cs[c]));
            AZ_TRY(prof_end(e, cs[c]));
        }
        e->roll_started = true;
    }
    for (int s = 0; s < n_rounds; s++) {
        const int par = e->roll_par;
        for (int c = 0; c < C; c++) {
            AZ_TRY(prof_begin(e, PROF_CNN, cs[c]));
            AZ_TRY(cnn_eval_leaves(e->net, dc[c][par], cs[c]));
            AZ_TRY(prof_end(e, cs[c]));
            AZ_TRY(prof_begin(e, PROF_BACKUP, cs[c]));
            AZ_TRY(roll(dc[c][par ^ 1], 3, dc[c][par].eval_count, cs[c]));
            AZ_TRY(prof_end(e, cs[c]));
        }
        e->roll_par ^= 1;
    }
    if (C > 1) {
        for (int c = 1; c < C; c++) {
            AZ_HIP(hipEventRecord(e->ev_join[c], cs[c]));
            AZ_HIP(hipStreamWaitEvent(e->stream, e->ev_join[c], 0));
        }
    }
    return AZ_OK;
}

int64_t az_selfplay_rolling(az_engine* e, int64_t n_games_target, double temp, int32_t first_n_steps, int64_t max_calls) {
    AZ_CHECK_E(e);
    if (n_games_target <= 0) return bad_arg("az_selfplay_rolling");
    AZ_TRY(az_reset_games(e));
    unsigned long long zero = 0;
    AZ_HIP(hipMemcpy(e->p.out_count, &zero, 8, hipMemcpyHostToDevice));
    AZ_HIP(hipMemcpy(e->p.out_games, &zero, 8, hipMemcpyHostToDevice));
    AZ_TRY(ensure_out_cap(e, (n_games_target + e->d.G) * (i64)e->gi.MAX_PLIES));
    e->target = (unsigned long long)n_games_target;
    const Dev& d = e->d;
    if (n_games_target < d.G) {  // fewer games than slots: park the surplus slots
        std::vector<u8> act(d.G, 0);
        for (i64 g = 0; g < n_games_target; g++) act[g] = 1;
        AZ_HIP(hipMemcpy(d.active, act.data(), d.G, hipMemcpyHostToDevice));
    }
    unsigned long long finished = 0;
    for (int64_t call = 0; max_calls <= 0 || call < max_calls; call++) {
        AZ_TRY(az_selfplay_rounds(e, e->cfg.n_playout, temp, first_n_steps, 1));
        int rc = az_engine_sync(e);
        if (rc != AZ_OK) return rc;
        AZ_HIP(hipMemcpy(&finished, e->p.out_games, 8, hipMemcpyDeviceToHost));
        if (finished >= (unsigned long long)n_games_target) break;
    }
    return (int64_t)finished;
}

int az_set_game_target(az_engine* e, int64_t n_games_target) {
    AZ_CHECK_E(e);
    e->target = n_games_target > 0 ? (unsigned long long)n_games_target : 0ull;
    return AZ_OK;
}

int az_set_active(az_engine* e, const uint8_t* mask) {
    AZ_CHECK_E(e);
    if (!mask) return bad_arg("az_set_active");
    const Dev& d = e->d;
    AZ_HIP(hipStreamSynchronize(e->stream));
    std::vector<u8> done(d.G), act(d.G);
    AZ_HIP(hipMemcpy(done.data(), d.done, d.G, hipMemcpyDeviceToHost));
    for (int g = 0; g < d.G; g++) act[g] = (mask[g] && !done[g]) ? 1 : 0;   // a finished game is never searched again
    AZ_HIP(hipMemcpy(d.active, act.data(), d.G, hipMemcpyHostToDevice));
    return AZ_OK;
}

int64_t az_selfplay(az_engine* e, int64_t n_games_target, double temp, int32_t first_n_steps, int64_t max_steps) {
    AZ_CHECK_E(e);
    if (n_games_target <= 0) return bad_arg("az_selfplay");
    AZ_TRY(az_reset_games(e));
    unsigned long long zero = 0;
    AZ_HIP(hipMemcpy(e->p.out_count, &zero, 8, hipMemcpyHostToDevice));
    AZ_HIP(hipMemcpy(e->p.out_games, &zero, 8, hipMemcpyHostToDevice));
    AZ_TRY(ensure_out_cap(e, (n_games_target + e->d.G) * (i64)e->gi.MAX_PLIES));
    e->target = (unsigned long long)n_games_target;
    const Dev& d = e->d;
    if (n_games_target < d.G) {  // fewer games than slots: park the surplus slots
        std::vector<u8> act(d.G, 0);
        for (i64 g = 0; g < n_games_target; g++) act[g] = 1;
        AZ_HIP(hipMemcpy(d.active, act.data(), d.G, hipMemcpyHostToDevice));
    }
    unsigned long long finished = 0;
    for (int64_t step = 0; max_steps <= 0 || step < max_steps; step++) {
        AZ_TRY(az_selfplay_step(e, temp, first_n_steps, 1));
        int rc = az_engine_sync(e);
        if (rc != AZ_OK) return rc;
        AZ_HIP(hipMemcpy(&finished, e->p.out_games, 8, hipMemcpyDeviceToHost));
        if (finished >= (unsigned long long)n_games_target) break;
    }
    return (int64_t)finished;
}

int az_traj_count(az_engine* e, int64_t* n_games, int64_t* n_plies) {
    AZ_CHECK_E(e);
    AZ_HIP(hipStreamSynchronize(e->stream));
    unsigned long long a = 0, b = 0;
    AZ_HIP(hipMemcpy(&a, e->p.out_games, 8, hipMemcpyDeviceToHost));
    AZ_HIP(hipMemcpy(&b, e->p.out_count, 8, hipMemcpyDeviceToHost));
    if ((i64)b > e->p.out_cap) b = (unsigned long long)e->p.out_cap;
    if (n_games) *n_games = (int64_t)a;
    if (n_plies) *n_plies = (int64_t)b;
    return AZ_OK;
}

int64_t az_traj_fetch(az_engine* e, int64_t max_plies, uint64_t* states, float* pi, float* z, int8_t* player,
                      int8_t* action, uint8_t* done, uint64_t* game_id, int clear) {
    AZ_CHECK_E(e);
    int64_t ng = 0, np = 0;
    AZ_TRY(az_traj_count(e, &ng, &np));
    int64_t n = np < max_plies ? np : max_plies;
    const Dev& d = e->d;
    const Play& p = e->p;
    if (n > 0) {
        if (states) AZ_HIP(hipMemcpy(states, p.out_state, (size_t)n * d.W * 8, hipMemcpyDeviceToHost));
        if (pi) AZ_HIP(hipMemcpy(pi, p.out_pi, (size_t)n * d.A * 4, hipMemcpyDeviceToHost));
        if (z) AZ_HIP(hipMemcpy(z, p.out_z, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (player) AZ_HIP(hipMemcpy(player, p.out_player, (size_t)n, hipMemcpyDeviceToHost));
        if (action) AZ_HIP(hipMemcpy(action, p.out_action, (size_t)n, hipMemcpyDeviceToHost));
        if (done) AZ_HIP(hipMemcpy(done, p.out_done, (size_t)n, hipMemcpyDeviceToHost));
        if (game_id) AZ_HIP(hipMemcpy(game_id, p.out_gid, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    if (clear) {
        AZ_HIP(hipMemset(p.out_count, 0, 8));
        AZ_HIP(hipMemset(p.out_games, 0, 8));
    }
    return n;
}

int az_traj_record_bytes(int game) {
    GameInfo gi;
    return game_info(game, gi) ? gi.W * 8 + gi.A * 4 + 4 : -1;
}

int64_t az_traj_pack_dev(az_engine* e, int64_t max_plies, void* out_dev, int clear) {
    AZ_CHECK_E(e);
    int64_t ng = 0, np = 0;
    AZ_TRY(az_traj_count(e, &ng, &np));
    int64_t n = np < max_plies ? np : max_plies;
    if (n > 0) {
        if (!out_dev) return bad_arg("az_traj_pack_dev");
        AZ_TRY(launch_traj_pack(e->cfg.game, e->p, n, (u8*)out_dev, e->stream));
        AZ_HIP(hipStreamSynchronize(e->stream));
    }
    if (clear) {
        AZ_HIP(hipMemset(e->p.out_count, 0, 8));
        AZ_HIP(hipMemset(e->p.out_games, 0, 8));
    }
    return n;
}

int az_traj_unpack_dev(int game, int64_t n, const void* packed_dev, float* state_dev, float* prob_dev,
                       float* value_dev, int32_t* winner_dev, float* next_state_dev, uint8_t* done_dev) {
    GameInfo gi;
    if (!game_info(game, gi) || n < 0 || (n > 0 && !packed_dev)) {
        set_error("az_traj_unpack_dev: bad argument");
        return AZ_ERR_ARG;
    }
    AZ_TRY(launch_traj_unpack(game, n, (const u8*)packed_dev, state_dev, prob_dev, value_dev, winner_dev,
                              next_state_dev, done_dev, 0));
    AZ_HIP(hipStreamSynchronize(0));
    return AZ_OK;
}

// ------------------------------------------------------------------------------------------------ network
int az_net_load(az_engine* e, const float* blob, size_t n_floats) {
    AZ_CHECK_E(e);
    if (e->cfg.game != AZ_GAME_C4) {
        set_error("az_net_load: the CNN is the Connect4 network");
        return AZ_ERR_ARG;
    }
    // the new network is built beside the live one and swapped in only when the blob was accepted: a refused blob (wrong
    // size, a NaN from a diverged training step) leaves the engine searching on the weights it had
    Cnn* fresh = nullptr;
    AZ_TRY(cnn_create(blob, n_floats, e->k3_kind, &fresh));
    hipError_t he = hipStreamSynchronize(e->stream);
    for (int c = 1; c < e->C && he == hipSuccess; c++) he = hipStreamSynchronize(e->cstream[c]);
    if (he != hipSuccess) {
        cnn_destroy(fresh);
        set_error(std::string("az_net_load: hipStreamSynchronize: ") + hipGetErrorString(he));
        return AZ_ERR_HIP;
    }
    if (e->net) cnn_destroy(e->net);
    e->net = fresh;
    // new weights (or another K3 kernel: its bits differ in the last place): every entry of the evaluation cache is stale.
    // The epoch is folded into the entries' tags, so none of them validates any more; nothing is cleared.
    e->cache_loads++;
    e->d.cache_epoch = mix64(0xA24BAED4963EE407ull + e->cache_loads * 0x9FB21C651E98DF25ull + (u64)(uintptr_t)e);
    // speculative rows marked under the previous weights and not yet stored (still on a list, or a slot parked by az_set_active) must
    // not enter the table under the new epoch: a row's mark is the tag of the weights it was marked under
    e->d.spec_tag = (u8)(1 + e->cache_loads % 250);
    return AZ_OK;
}

int az_net_set_kernel(az_engine* e, int kind) {
    AZ_CHECK_E(e);
    if (kind < AZ_K3_DEFAULT || kind > AZ_K3_DIRECT) return bad_arg("az_net_set_kernel");
    e->k3_kind = kind;
    return AZ_OK;
}

int az_net_get_kernel(az_engine* e) {
    AZ_CHECK_E(e);
    if (!e->net) {
        set_error("az_net_get_kernel: call az_net_load first");
        return AZ_ERR_NO_NET;
    }
    return cnn_kind(e->net);
}

int az_net_forward(az_engine* e, int64_t n, const uint64_t* states, float* logp, float* logv, float* probs,
                   float* value) {
    AZ_CHECK_E(e);
    if (!e->net) {
        set_error("az_net_forward: call az_net_load first");
        return AZ_ERR_NO_NET;
    }
    if (n < 0 || !states) return bad_arg("az_net_forward");
    if (n == 0) return AZ_OK;
    TmpBufs t;
    Block out;
    u64* ds;
    AZ_TRY(t.get(&ds, n * 2));
    const size_t o_lp = out.reserve((size_t)n * 28), o_lv = out.reserve((size_t)n * 12), o_p = out.reserve((size_t)n * 28),
                 o_v = out.reserve((size_t)n * 4);
    AZ_TRY(out.alloc(t));
    AZ_HIP(hipMemcpyAsync(ds, states, n * 16, hipMemcpyHostToDevice, e->stream));
    AZ_TRY(cnn_forward(e->net, n, ds, out.d<float>(o_lp), out.d<float>(o_lv), out.d<float>(o_p), out.d<float>(o_v), e->stream));
    AZ_HIP(hipMemcpyAsync(out.host.data(), out.dev, out.size, hipMemcpyDeviceToHost, e->stream));   // all four outputs at once
    AZ_HIP(hipStreamSynchronize(e->stream));
    out.take(o_lp, logp, (size_t)n * 28);
    out.take(o_lv, logv, (size_t)n * 12);
    out.take(o_p, probs, (size_t)n * 28);
    out.take(o_v, value, (size_t)n * 4);
    return AZ_OK;
}

int az_net_forward_static(az_engine* e, int64_t n, const uint64_t* states, float* logp, float* logv, float* probs,
                          float* value) {
    AZ_CHECK_E(e);
    if (!e->net) {
        set_error("az_net_forward_static: call az_net_load first");
        return AZ_ERR_NO_NET;
    }
    if (n < 0 || !states) return bad_arg("az_net_forward_static");
    if (n > AZ_NET_FORWARD_STATIC_MAX) {
        set_error("az_net_forward_static: " + std::to_string(n) + " positions exceed AZ_NET_FORWARD_STATIC_MAX (" +
                  std::to_string(AZ_NET_FORWARD_STATIC_MAX) + "); use az_net_forward or split the batch");
        return AZ_ERR_ARG;
    }
    if (n == 0) return AZ_OK;
    StaticStage* st = e->cfg.device < MAX_DEVICES ? g_stage[e->cfg.device] : nullptr;   // set by az_engine_create on this device
    if (!st) {
        set_error("az_net_forward_static: static staging memory was not initialised on this device");
        return AZ_ERR_STATE;
    }
    // every byte the kernels read (n positions) is written here first and every output row of the n positions is
    // written by the kernel, so nothing in the array has to start at zero; rows beyond n keep the last call's contents
    AZ_HIP(hipMemcpyAsync(st->states, states, (size_t)n * 16, hipMemcpyHostToDevice, e->stream));
    AZ_TRY(cnn_forward(e->net, n, st->states, st->logp, st->logv, st->probs, st->value, e->stream));
    if (logp) AZ_HIP(hipMemcpyAsync(logp, st->logp, (size_t)n * 28, hipMemcpyDeviceToHost, e->stream));
    if (logv) AZ_HIP(hipMemcpyAsync(logv, st->logv, (size_t)n * 12, hipMemcpyDeviceToHost, e->stream));
    if (probs) AZ_HIP(hipMemcpyAsync(probs, st->probs, (size_t)n * 28, hipMemcpyDeviceToHost, e->stream));
    if (value) AZ_HIP(hipMemcpyAsync(value, st->value, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream));
    AZ_HIP(hipStreamSynchronize(e->stream));
    return AZ_OK;
}

int az_net_forward_dev(az_engine* e, int64_t n, const uint64_t* states_dev, float* probs_dev, float* value_dev) {
    AZ_CHECK_E(e);
    if (!e->net) {
        set_error("az_net_forward_dev: call az_net_load first");
        return AZ_ERR_NO_NET;
    }
    return cnn_forward(e->net, n, states_dev, nullptr, nullptr, probs_dev, value_dev, e->stream);
}

}  // extern "C"
