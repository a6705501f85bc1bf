// HBM-resident episodic replay buffer with a gather + n-step-return relabel kernel.
// Replaces /root/reference/utils/replay_buffer.py:153-239 (ReplayBuffer._store_episode/_sample) and the
// DataLoader collate at :260-277: one launch emits a whole coalesced (s, a, R, D, s') minibatch.
//
// Layout: a row arena (SoA) — obs[rows][obs_bytes], action[rows][A], reward[rows], discount[rows],
// meta[rows][Mt] — episodes are len+1 consecutive rows (row 0 = dummy reset step, replay_buffer.py:13-15);
// an episode table (row0, len) in the reference's sorted `_episode_fns` order lives beside it.
// Kernel: one wave per sample; lanes stream the obs / next_obs rows as 16-byte (or 4-byte) words, lane 0
// runs the n-step recurrence with products and sums rounded separately (no FMA contraction) so values are
// bit-identical to NumPy's fp32 arithmetic (replay_buffer.py:229-234). HBM-bound, ~2*(2*obs_bytes+4A+8) B/sample.
//
// Index streams: EXORL_SAMPLER_MT19937 reproduces the reference's two MT19937 streams on the host
// (CPython random.choice -> _randbelow_with_getrandbits; NumPy legacy randint masked rejection) and ships the
// B index pairs with one async copy; EXORL_SAMPLER_PHILOX draws them in the kernel (Philox4x32-10).
#include <algorithm>
#include <vector>

#include "kernels.h"

namespace exorl {

// ---- MT19937 (host) ----------------------------------------------------------------------------
struct MT19937 {
    uint32_t mt[624];
    int pos = 624;
    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        pos = 624;
    }
    void init_by_array(const uint32_t* key, int klen) {
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = (624 > klen ? 624 : klen); k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            ++i; ++j;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= klen) j = 0;
        }
        for (int k = 623; k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            ++i;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
        pos = 624;
    }
    void twist() {
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7FFFFFFFu);
            uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908B0DFu;
            mt[k] = v;
        }
        pos = 0;
    }
    uint32_t next() {
        if (pos >= 624) twist();
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9D2C5680u;
        y ^= (y << 15) & 0xEFC60000u;
        y ^= y >> 18;
        return y;
    }
    // CPython random._randbelow_with_getrandbits(n), 0 < n < 2^32
    uint32_t py_randbelow(uint32_t n) {
        int k = 32 - __builtin_clz(n);
        uint32_t r = next() >> (32 - k);
        while (r >= n) r = next() >> (32 - k);
        return r;
    }
    // NumPy legacy RandomState.randint(0, hi), hi >= 1
    uint32_t np_randint0(uint32_t hi) {
        const uint32_t rng = hi - 1;
        if (rng == 0) return 0;
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v = next() & mask;
        while (v > rng) v = next() & mask;
        return v;
    }
};

struct Slot {
    int64_t row0;
    int32_t rows;
    bool live;
};

}  // namespace exorl

using namespace exorl;

struct exorl_replay {
    exorl_replay_cfg cfg;
    unsigned char* obs = nullptr;
    float *act = nullptr, *rew = nullptr, *disc = nullptr, *meta = nullptr;
    int64_t used_rows = 0, live_rows = 0;
    std::vector<Slot> slots;
    std::vector<int32_t> order;
    int64_t* d_row0 = nullptr;
    int32_t* d_len = nullptr;
    int32_t* d_pairs = nullptr;
    int pairs_cap = 0;
    bool table_dirty = true;
    int32_t min_len = 0;
    MT19937 py, np;
    bool mt_seeded = false;
    uint64_t philox_seed = 0, philox_counter = 0;
    std::vector<int32_t> h_pairs;
    std::vector<int64_t> h_row0;
    std::vector<int32_t> h_len;
};

namespace exorl {

struct ReplayView {
    const unsigned char* obs;
    const float *act, *rew, *disc, *meta;
    const int64_t* row0;
    const int32_t* len;
    int32_t obs_bytes, act_dim, meta_dim, n_episodes;
};

__device__ __forceinline__ void copy_row(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                         int bytes, int lane, bool vec16) {
    if (vec16) {
        const uint4* s = reinterpret_cast<const uint4*>(src);
        uint4* d = reinterpret_cast<uint4*>(dst);
        for (int i = lane; i < bytes / 16; i += 64) d[i] = s[i];
    } else {
        const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        for (int i = lane; i < bytes / 4; i += 64) d[i] = s[i];
    }
}

__global__ __launch_bounds__(256) void gather_nstep_kernel(ReplayView v, const int32_t* pairs_in,
                                                           int32_t* pairs_out, exorl_batch_out out,
                                                           int batch, int nstep, float gamma, int sampler,
                                                           uint64_t seed, uint64_t counter_val,
                                                           const uint64_t* counter_ptr, int vec16, StageOut stage) {
#pragma clang fp contract(off)
    if (stage.st && blockIdx.x == 0 && threadIdx.x == 0) step_begin_device(stage.st, 0);   // noise counter, Adam scalars of this step
    const uint64_t counter = counter_ptr ? *counter_ptr : counter_val;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= batch) return;
    int pos, idx;
    if (sampler == EXORL_SAMPLER_PHILOX) {
        uint32_t c[4] = {(uint32_t)b, (uint32_t)counter, (uint32_t)(counter >> 32), 0u};
        Philox::gen(c, seed);
        pos = (int)(((uint64_t)c[0] * (uint64_t)v.n_episodes) >> 32);
        const int span = v.len[pos] - nstep + 1;
        idx = (int)(((uint64_t)c[1] * (uint64_t)span) >> 32) + 1;
        if (lane == 0) { pairs_out[2 * b] = pos; pairs_out[2 * b + 1] = idx; }
    } else {
        pos = pairs_in[2 * b];
        idx = pairs_in[2 * b + 1];
    }
    const int64_t row = v.row0[pos] + idx;
    copy_row(v.obs + (row - 1) * v.obs_bytes, static_cast<unsigned char*>(out.obs) + (int64_t)b * out.obs_stride,
             v.obs_bytes, lane, vec16);
    copy_row(v.obs + (row + nstep - 1) * v.obs_bytes,
             static_cast<unsigned char*>(out.next_obs) + (int64_t)b * out.next_obs_stride, v.obs_bytes, lane, vec16);
    for (int j = lane; j < v.act_dim; j += 64) out.action[(int64_t)b * out.action_stride + j] = v.act[row * v.act_dim + j];
    if (out.meta)
        for (int j = lane; j < v.meta_dim; j += 64)
            out.meta[(int64_t)b * out.meta_stride + j] = v.meta[(row - 1) * v.meta_dim + j];
    if (stage.xa) {                                  // the agent's staged network inputs, straight from the arena rows
        const int O = stage.O, A = stage.A, W = O + A;
        const float* so = reinterpret_cast<const float*>(v.obs + (row - 1) * v.obs_bytes);
        const float* sn = reinterpret_cast<const float*>(v.obs + (row + nstep - 1) * v.obs_bytes);
        for (int j = lane; j < O; j += 64) {
            const float o = so[j], no = sn[j];
            stage.xa[(int64_t)b * O + j] = no;
            stage.xa[(int64_t)(stage.B + b) * O + j] = o;
            if (stage.has_critic) {
                stage.xc_cur[(int64_t)b * W + j] = o;
                stage.xc_next[(int64_t)b * W + j] = no;
                stage.xc_pi[(int64_t)b * W + j] = o;
            }
        }
        if (stage.has_critic)
            for (int j = lane; j < A; j += 64) stage.xc_cur[(int64_t)b * W + O + j] = v.act[row * v.act_dim + j];
    }
    if (lane == 0) {
        float R = 0.0f, D = 1.0f;
        for (int i = 0; i < nstep; ++i) {
            const float t = D * v.rew[row + i];        // replay_buffer.py:233  reward += discount * step_reward
            R = R + t;
            const float u = v.disc[row + i] * gamma;   // replay_buffer.py:234  discount *= ep_discount * gamma
            D = D * u;
        }
        out.reward[b] = R;
        out.discount[b] = D;
    }
}

static int upload_table(exorl_replay* r, hipStream_t s) {
    if (!r->table_dirty) return 0;
    const int n = (int)r->order.size();
    r->h_row0.resize(n);
    r->h_len.resize(n);
    int32_t mn = INT32_MAX;
    for (int i = 0; i < n; ++i) {
        const Slot& sl = r->slots[r->order[i]];
        r->h_row0[i] = sl.row0;
        r->h_len[i] = sl.rows - 1;
        mn = sl.rows - 1 < mn ? sl.rows - 1 : mn;
    }
    r->min_len = n ? mn : 0;
    if (n) {
        EXORL_CHECK_HIP(hipMemcpyAsync(r->d_row0, r->h_row0.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, s));
        EXORL_CHECK_HIP(hipMemcpyAsync(r->d_len, r->h_len.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, s));
        // pageable sources: the runtime stages them before returning, so the vectors may be reused
    }
    r->table_dirty = false;
    return 0;
}

// Moves live episodes down over evicted holes (bounce buffer handles overlapping ranges).
static int compact(exorl_replay* r) {
    EXORL_CHECK_HIP(hipDeviceSynchronize());
    std::vector<int> live;
    for (int i = 0; i < (int)r->slots.size(); ++i)
        if (r->slots[i].live) live.push_back(i);
    std::sort(live.begin(), live.end(), [&](int a, int b) { return r->slots[a].row0 < r->slots[b].row0; });
    const int64_t chunk_rows_cap = 4096;
    const int64_t widest = std::max<int64_t>(r->cfg.obs_bytes, std::max<int64_t>(r->cfg.act_dim, r->cfg.meta_dim) * 4);
    void* tmp = nullptr;
    EXORL_CHECK_HIP(hipMalloc(&tmp, chunk_rows_cap * std::max<int64_t>(widest, 4)));
    int64_t dst = 0;
    auto move = [&](unsigned char* base, int64_t rowbytes, int64_t from, int64_t to, int64_t rows) -> int {
        if (!base || rowbytes == 0 || from == to) return 0;
        for (int64_t done = 0; done < rows; done += chunk_rows_cap) {
            const int64_t n = std::min(chunk_rows_cap, rows - done);
            EXORL_CHECK_HIP(hipMemcpy(tmp, base + (from + done) * rowbytes, n * rowbytes, hipMemcpyDeviceToDevice));
            EXORL_CHECK_HIP(hipMemcpy(base + (to + done) * rowbytes, tmp, n * rowbytes, hipMemcpyDeviceToDevice));
        }
        return 0;
    };
    for (int i : live) {
        Slot& sl = r->slots[i];
        EXORL_TRY(move(r->obs, r->cfg.obs_bytes, sl.row0, dst, sl.rows));
        EXORL_TRY(move((unsigned char*)r->act, r->cfg.act_dim * 4, sl.row0, dst, sl.rows));
        EXORL_TRY(move((unsigned char*)r->rew, 4, sl.row0, dst, sl.rows));
        EXORL_TRY(move((unsigned char*)r->disc, 4, sl.row0, dst, sl.rows));
        EXORL_TRY(move((unsigned char*)r->meta, r->cfg.meta_dim * 4, sl.row0, dst, sl.rows));
        sl.row0 = dst;
        dst += sl.rows;
    }
    EXORL_CHECK_HIP(hipFree(tmp));
    r->used_rows = dst;
    r->table_dirty = true;
    return 0;
}

}  // namespace exorl

extern "C" {

int exorl_replay_create(const exorl_replay_cfg* cfg, exorl_replay_t** out) {
    EXORL_REQUIRE(cfg && out, "replay_create: null argument");
    EXORL_REQUIRE(cfg->obs_bytes > 0 && cfg->obs_bytes % 4 == 0, "replay_create: obs_bytes=%d must be a positive multiple of 4",
                  cfg->obs_bytes);
    EXORL_REQUIRE(cfg->act_dim > 0 && cfg->meta_dim >= 0 && cfg->capacity_rows > 0 && cfg->max_episodes > 0,
                  "replay_create: bad dims");
    auto* r = new exorl_replay();
    r->cfg = *cfg;
    const int64_t rows = cfg->capacity_rows;
    EXORL_CHECK_HIP(hipMalloc((void**)&r->obs, rows * cfg->obs_bytes));
    EXORL_CHECK_HIP(hipMalloc((void**)&r->act, rows * cfg->act_dim * 4));
    EXORL_CHECK_HIP(hipMalloc((void**)&r->rew, rows * 4));
    EXORL_CHECK_HIP(hipMalloc((void**)&r->disc, rows * 4));
    if (cfg->meta_dim) EXORL_CHECK_HIP(hipMalloc((void**)&r->meta, rows * cfg->meta_dim * 4));
    EXORL_CHECK_HIP(hipMalloc((void**)&r->d_row0, (size_t)cfg->max_episodes * sizeof(int64_t)));
    EXORL_CHECK_HIP(hipMalloc((void**)&r->d_len, (size_t)cfg->max_episodes * sizeof(int32_t)));
    *out = r;
    return 0;
}

int exorl_replay_destroy(exorl_replay_t* r) {
    if (!r) return 0;
    (void)hipFree(r->obs); (void)hipFree(r->act); (void)hipFree(r->rew); (void)hipFree(r->disc);
    if (r->meta) (void)hipFree(r->meta);
    (void)hipFree(r->d_row0); (void)hipFree(r->d_len);
    if (r->d_pairs) (void)hipFree(r->d_pairs);
    delete r;
    return 0;
}

int exorl_replay_append_episode(exorl_replay_t* r, const void* obs, const float* act, const float* rew,
                                const float* disc, const float* meta, int32_t rows, int32_t* slot_out) {
    EXORL_REQUIRE(r && obs && act && rew && disc && slot_out, "replay_append_episode: null argument");
    EXORL_REQUIRE(rows >= 2, "replay_append_episode: rows=%d (an episode is a dummy row + >=1 transition)", rows);
    EXORL_REQUIRE((r->cfg.meta_dim == 0) == (meta == nullptr), "replay_append_episode: meta pointer/meta_dim mismatch");
    if (r->used_rows + rows > r->cfg.capacity_rows) {
        EXORL_REQUIRE(r->live_rows + rows <= r->cfg.capacity_rows,
                      "replay_append_episode: arena full (%lld live + %d > capacity %lld rows)", (long long)r->live_rows, rows,
                      (long long)r->cfg.capacity_rows);
        EXORL_TRY(compact(r));
    }
    int slot = -1;
    for (int i = 0; i < (int)r->slots.size(); ++i)
        if (!r->slots[i].live) { slot = i; break; }
    if (slot < 0) {
        EXORL_REQUIRE((int)r->slots.size() < r->cfg.max_episodes, "replay_append_episode: more than max_episodes=%d resident",
                      r->cfg.max_episodes);
        r->slots.push_back(Slot{0, 0, false});
        slot = (int)r->slots.size() - 1;
    }
    const int64_t row0 = r->used_rows;
    EXORL_CHECK_HIP(hipMemcpy(r->obs + row0 * r->cfg.obs_bytes, obs, (size_t)rows * r->cfg.obs_bytes, hipMemcpyHostToDevice));
    EXORL_CHECK_HIP(hipMemcpy(r->act + row0 * r->cfg.act_dim, act, (size_t)rows * r->cfg.act_dim * 4, hipMemcpyHostToDevice));
    EXORL_CHECK_HIP(hipMemcpy(r->rew + row0, rew, (size_t)rows * 4, hipMemcpyHostToDevice));
    EXORL_CHECK_HIP(hipMemcpy(r->disc + row0, disc, (size_t)rows * 4, hipMemcpyHostToDevice));
    if (meta) EXORL_CHECK_HIP(hipMemcpy(r->meta + row0 * r->cfg.meta_dim, meta, (size_t)rows * r->cfg.meta_dim * 4, hipMemcpyHostToDevice));
    r->slots[slot] = Slot{row0, rows, true};
    r->used_rows += rows;
    r->live_rows += rows;
    *slot_out = slot;
    return 0;
}

int exorl_replay_evict(exorl_replay_t* r, int32_t slot) {
    EXORL_REQUIRE(r && slot >= 0 && slot < (int)r->slots.size() && r->slots[slot].live, "replay_evict: slot %d not resident", slot);
    r->slots[slot].live = false;
    r->live_rows -= r->slots[slot].rows;
    for (size_t i = 0; i < r->order.size(); ++i)
        if (r->order[i] == slot) { r->order.erase(r->order.begin() + i); break; }
    r->table_dirty = true;
    return 0;
}

int exorl_replay_set_order(exorl_replay_t* r, const int32_t* slots, int32_t n) {
    EXORL_REQUIRE(r && (slots || n == 0) && n >= 0 && n <= r->cfg.max_episodes, "replay_set_order: bad arguments");
    for (int i = 0; i < n; ++i)
        EXORL_REQUIRE(slots[i] >= 0 && slots[i] < (int)r->slots.size() && r->slots[slots[i]].live,
                      "replay_set_order: slot %d (position %d) not resident", slots[i], i);
    r->order.assign(slots, slots + n);
    r->table_dirty = true;
    return 0;
}

int exorl_replay_num_rows(exorl_replay_t* r, int64_t* live_rows, int64_t* used_rows) {
    EXORL_REQUIRE(r, "replay_num_rows: null handle");
    if (live_rows) *live_rows = r->live_rows;
    if (used_rows) *used_rows = r->used_rows;
    return 0;
}

int exorl_replay_seed_mt(exorl_replay_t* r, const uint32_t* py_key, int32_t py_pos, const uint32_t* np_key, int32_t np_pos) {
    EXORL_REQUIRE(r && py_key && np_key, "replay_seed_mt: null argument");
    EXORL_REQUIRE(py_pos >= 0 && py_pos <= 624 && np_pos >= 0 && np_pos <= 624, "replay_seed_mt: bad position");
    memcpy(r->py.mt, py_key, sizeof(r->py.mt)); r->py.pos = py_pos;
    memcpy(r->np.mt, np_key, sizeof(r->np.mt)); r->np.pos = np_pos;
    r->mt_seeded = true;
    return 0;
}

int exorl_replay_seed_mt_ints(exorl_replay_t* r, uint64_t py_seed, uint32_t np_seed) {
    EXORL_REQUIRE(r, "replay_seed_mt_ints: null handle");
    uint32_t key[2] = {(uint32_t)py_seed, (uint32_t)(py_seed >> 32)};
    r->py.init_by_array(key, key[1] ? 2 : 1);      // random.seed(int): 32-bit limbs of abs(seed)
    r->np.init_genrand(np_seed);                   // np.random.seed(int)
    r->mt_seeded = true;
    return 0;
}

int exorl_replay_get_mt(exorl_replay_t* r, uint32_t* py_key, int32_t* py_pos, uint32_t* np_key, int32_t* np_pos) {
    EXORL_REQUIRE(r && py_key && np_key && py_pos && np_pos, "replay_get_mt: null argument");
    memcpy(py_key, r->py.mt, sizeof(r->py.mt)); *py_pos = r->py.pos;
    memcpy(np_key, r->np.mt, sizeof(r->np.mt)); *np_pos = r->np.pos;
    return 0;
}

int exorl_replay_seed_philox(exorl_replay_t* r, uint64_t seed) {
    EXORL_REQUIRE(r, "replay_seed_philox: null handle");
    r->philox_seed = seed;
    r->philox_counter = 0;
    return 0;
}

}  // extern "C"

namespace exorl {
int replay_prepare(exorl_replay* r, int32_t batch, int32_t nstep, hipStream_t s) {
    EXORL_REQUIRE(r && batch > 0, "replay_prepare: bad arguments");
    EXORL_REQUIRE(!r->order.empty(), "replay_sample: no resident episodes (IndexError in random.choice, replay_buffer.py:169)");
    EXORL_TRY(upload_table(r, s));
    if (batch > r->pairs_cap) {
        if (r->d_pairs) EXORL_CHECK_HIP(hipFree(r->d_pairs));
        EXORL_CHECK_HIP(hipMalloc((void**)&r->d_pairs, (size_t)batch * 2 * sizeof(int32_t)));
        r->pairs_cap = batch;
    }
    if (nstep > 0)      // Philox draws a start inside every episode: all of them must hold nstep transitions
        EXORL_REQUIRE(r->min_len - nstep + 1 >= 1, "replay_sample: shortest episode (%d) shorter than nstep=%d", r->min_len, nstep);
    return 0;
}

// dev_counter != nullptr: Philox batch counter is read from device memory (graph-replayable); the host copy is
// still advanced so eager sampling continues the stream afterwards.
int replay_sample_impl(exorl_replay* r, int32_t batch, int32_t nstep, float gamma, int32_t sampler,
                       const int32_t* pairs_host, const exorl_batch_out* out, int32_t* pairs_out_host, hipStream_t s,
                       const uint64_t* dev_counter, const StageOut* stage) {
    EXORL_REQUIRE(r && out, "replay_sample: null argument");
    EXORL_REQUIRE(!stage || (r->cfg.obs_bytes == stage->O * 4 && r->cfg.act_dim == stage->A && stage->B == batch),
                  "replay_sample: staged outputs need fp32 state observations of the agent's dimensions");
    EXORL_REQUIRE(batch > 0 && nstep >= 1, "replay_sample: batch=%d nstep=%d", batch, nstep);
    EXORL_REQUIRE(out->obs && out->action && out->reward && out->discount && out->next_obs, "replay_sample: null output");
    EXORL_REQUIRE((r->cfg.meta_dim > 0) || out->meta == nullptr, "replay_sample: meta output without meta columns");
    const int n = (int)r->order.size();
    EXORL_TRY(replay_prepare(r, batch, sampler == EXORL_SAMPLER_PHILOX ? nstep : 0, s));
    if (sampler == EXORL_SAMPLER_MT19937 || sampler == EXORL_SAMPLER_GIVEN) {
        r->h_pairs.resize((size_t)batch * 2);
        if (sampler == EXORL_SAMPLER_MT19937) {
            EXORL_REQUIRE(r->mt_seeded, "replay_sample: MT19937 streams not seeded (exorl_replay_seed_mt)");
            for (int b = 0; b < batch; ++b) {
                const int pos = (int)r->py.py_randbelow((uint32_t)n);
                const int len = r->slots[r->order[pos]].rows - 1;
                EXORL_REQUIRE(len - nstep + 1 >= 1, "replay_sample: episode of length %d shorter than nstep=%d (ValueError in "
                              "np.random.randint, replay_buffer.py:222)", len, nstep);
                r->h_pairs[2 * b] = pos;
                r->h_pairs[2 * b + 1] = (int)r->np.np_randint0((uint32_t)(len - nstep + 1)) + 1;
            }
        } else {
            EXORL_REQUIRE(pairs_host, "replay_sample: EXORL_SAMPLER_GIVEN needs pairs_host");
            for (int b = 0; b < batch; ++b) {
                const int pos = pairs_host[2 * b], idx = pairs_host[2 * b + 1];
                EXORL_REQUIRE(pos >= 0 && pos < n, "replay_sample: pair %d: position %d out of range [0,%d)", b, pos, n);
                const int len = r->slots[r->order[pos]].rows - 1;
                EXORL_REQUIRE(idx >= 1 && idx + nstep - 1 <= len, "replay_sample: pair %d: idx %d out of range for len %d nstep %d",
                              b, idx, len, nstep);
                r->h_pairs[2 * b] = pos;
                r->h_pairs[2 * b + 1] = idx;
            }
        }
        EXORL_CHECK_HIP(hipMemcpyAsync(r->d_pairs, r->h_pairs.data(), (size_t)batch * 2 * sizeof(int32_t), hipMemcpyHostToDevice, s));
        if (pairs_out_host) memcpy(pairs_out_host, r->h_pairs.data(), (size_t)batch * 2 * sizeof(int32_t));
    } else if (sampler != EXORL_SAMPLER_PHILOX) {
        set_error("replay_sample: unknown sampler %d", sampler);
        return 2;
    }
    ReplayView v{r->obs, r->act, r->rew, r->disc, r->meta, r->d_row0, r->d_len, r->cfg.obs_bytes, r->cfg.act_dim, r->cfg.meta_dim, n};
    const int vec16 = (r->cfg.obs_bytes % 16 == 0) && (out->obs_stride % 16 == 0) && (out->next_obs_stride % 16 == 0) &&
                      ((uintptr_t)out->obs % 16 == 0) && ((uintptr_t)out->next_obs % 16 == 0);
    hipLaunchKernelGGL(gather_nstep_kernel, dim3(cdiv(batch, 4)), dim3(256), 0, s, v, r->d_pairs, r->d_pairs, *out, batch, nstep,
                       gamma, sampler, r->philox_seed, r->philox_counter, dev_counter, vec16, stage ? *stage : StageOut{});
    EXORL_LAUNCH_CHECK();
    if (sampler == EXORL_SAMPLER_PHILOX) r->philox_counter += 1;
    return 0;
}

uint64_t replay_philox_counter(exorl_replay* r) { return r->philox_counter; }
int replay_obs_bytes(exorl_replay* r) { return r->cfg.obs_bytes; }
void replay_advance_philox(exorl_replay* r, uint64_t n) { r->philox_counter += n; }
}  // namespace exorl

extern "C" {

int exorl_replay_sample(exorl_replay_t* r, int32_t batch, int32_t nstep, float gamma, int32_t sampler,
                        const int32_t* pairs_host, const exorl_batch_out* out, int32_t* pairs_out_host, void* stream) {
    return replay_sample_impl(r, batch, nstep, gamma, sampler, pairs_host, out, pairs_out_host, as_stream(stream), nullptr);
}

int exorl_replay_last_pairs(exorl_replay_t* r, int32_t batch, int32_t* pairs_host, void* stream) {
    EXORL_REQUIRE(r && pairs_host && batch > 0 && batch <= r->pairs_cap, "replay_last_pairs: bad arguments");
    hipStream_t s = as_stream(stream);
    EXORL_CHECK_HIP(hipMemcpyAsync(pairs_host, r->d_pairs, (size_t)batch * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    EXORL_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
