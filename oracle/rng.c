/*
 * oracle/rng.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * (1) The reference's thread-local RNG: simulator_util.cpp:38-73 uses
 *     std::default_random_engine (libstdc++: minstd_rand0), seeded from
 *     std::hash<std::string>(to_string(seed + ++n_threads)) truncated to int,
 *     std::uniform_int_distribution<int> and std::uniform_real_distribution<float>.
 *     libstdc++ is a third-party dependency of the reference that is not under
 *     /root/reference; its algorithms (bits/random.h, bits/random.tcc,
 *     bits/uniform_int_dist.h, libsupc++/hash_bytes.cc; GCC 4.x..11 identical
 *     for this engine) are restated here and pinned by the reference's own
 *     known-answer test tests/test_simulator_seed.cpp:22-50.
 * (2) Philox4x32-10, the build's own counter-based stream ("xwb-rng-v1").
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <string.h>

/* ---- minstd_rand0: x <- 16807 x mod (2^31 - 1) ------------------------- */
#define MINSTD_M 2147483647ULL
#define MINSTD_A 16807ULL

void orc_minstd_seed(orc_minstd *g, uint64_t s) {
    /* linear_congruential_engine::seed(result_type): c == 0, so s mod m == 0 -> 1 */
    uint64_t x = s % MINSTD_M;
    g->x = (uint32_t)(x == 0 ? 1 : x);
}

uint32_t orc_minstd_next(orc_minstd *g) {
    g->x = (uint32_t)((MINSTD_A * (uint64_t)g->x) % MINSTD_M);
    return g->x;
}

/* std::uniform_int_distribution<int>(0, size-1)(engine)  [bits/uniform_int_dist.h]
 * URNG range = max - min = 2147483645 > urange: classic down-scaling branch.      */
int orc_minstd_rand_ind(orc_minstd *g, int size) {
    const uint64_t urngrange = 2147483646ULL - 1ULL;
    const uint64_t uerange = (uint64_t)size;          /* urange + 1 */
    if (urngrange + 1 == uerange) return (int)(orc_minstd_next(g) - 1);
    const uint64_t scaling = urngrange / uerange;
    const uint64_t past = uerange * scaling;
    uint64_t ret;
    do {
        ret = (uint64_t)orc_minstd_next(g) - 1ULL;
    } while (ret >= past);
    return (int)(ret / scaling);
}

/* std::uniform_real_distribution<float>(0, upper)(engine)
 * generate_canonical<float, 24>: k = 1 call; sum = float(x - min); tmp = float(range) */
float orc_minstd_rand_range(orc_minstd *g, float upper) {
    const float tmp = (float)2147483646.0L;           /* max - min + 1, narrowed to float */
    float sum = (float)(orc_minstd_next(g) - 1u);
    float ret = sum / tmp;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret * (upper - 0.0f) + 0.0f;
}

/* libsupc++ hash_bytes.cc, 64-bit size_t variant (murmur2-like), seed 0xc70f6907 */
static uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

uint64_t orc_std_hash_string(const char *s, size_t len) {
    const uint64_t mul = (((uint64_t)0xc6a4a793UL) << 32) + (uint64_t)0x5bd1e995UL;
    const uint64_t seed = 0xc70f6907UL;
    const size_t len_aligned = len & ~(size_t)7;
    uint64_t hash = seed ^ (len * mul);
    for (size_t p = 0; p != len_aligned; p += 8) {
        uint64_t w;
        memcpy(&w, s + p, 8);
        uint64_t data = shift_mix(w * mul) * mul;
        hash ^= data;
        hash *= mul;
    }
    if ((len & 7) != 0) {
        int n = (int)(len & 7);
        uint64_t data = 0;
        --n;
        do {
            data = (data << 8) + (unsigned char)s[len_aligned + (size_t)n];
        } while (--n >= 0);
        hash ^= data;
        hash *= mul;
    }
    hash = shift_mix(hash) * mul;
    hash = shift_mix(hash);
    return hash;
}

/* ThreadCounter::ThreadCounter, simulator_util.cpp:40-50:
 *   int seed = std::hash<std::string>()(std::to_string(FLAGS_simulator_seed + (++__num_threads)));
 *   reng_.seed(seed);   // int -> unsigned long (sign extension) -> mod m            */
void orc_minstd_seed_thread(orc_minstd *g, int simulator_seed, int nth_thread) {
    char buf[32];
    int n = snprintf(buf, sizeof buf, "%d", simulator_seed + nth_thread);
    int seed = (int)orc_std_hash_string(buf, (size_t)n);
    orc_minstd_seed(g, (uint64_t)(int64_t)seed);
}

/* ---- Philox4x32-10 ------------------------------------------------------ */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* xwb-rng-v1: key = (seed, env_gid); ctr = (block, episode, stream_id, 0);
 * draws are the words of successive blocks in order. */
void orc_stream_init(orc_stream *s, uint32_t seed, uint32_t env_gid, uint32_t episode, uint32_t stream_id) {
    s->key[0] = seed; s->key[1] = env_gid;
    s->ctr[0] = 0; s->ctr[1] = episode; s->ctr[2] = stream_id; s->ctr[3] = 0;
    s->have = 0;
}

uint32_t orc_stream_u32(orc_stream *s) {
    if (s->have == 0) {
        orc_philox4x32(s->ctr, s->key, s->buf);
        s->ctr[0] += 1;
        s->have = 4;
    }
    uint32_t v = s->buf[4 - s->have];
    s->have -= 1;
    return v;
}

uint32_t orc_stream_below(orc_stream *s, uint32_t n) {
    /* always consumes exactly one draw (also for n <= 1): the number of draws a reset consumes is then a
     * function of the configuration only, never of the data */
    uint32_t v = orc_stream_u32(s);
    if (n <= 1) return 0;
    return (uint32_t)(((uint64_t)v * (uint64_t)n) >> 32);
}

float orc_stream_unit(orc_stream *s) {
    return (float)(orc_stream_u32(s) >> 8) * (1.0f / 16777216.0f);
}

/* random policy: key = (policy_seed, env_gid), ctr = (step, 0, 1, 0), word 0 */
int32_t orc_policy_action(uint32_t policy_seed, uint32_t env_gid, uint32_t step, int num_actions) {
    uint32_t key[2] = {policy_seed, env_gid};
    uint32_t ctr[4] = {step, 0u, 1u, 0u};
    uint32_t out[4];
    orc_philox4x32(ctr, key, out);
    return (int32_t)(((uint64_t)out[0] * (uint64_t)(uint32_t)num_actions) >> 32);
}

/* GameSimulator::decode_game_over_code, simulator.cpp:125-144 */
int orc_decode_game_over_code(int code, char *out, int cap) {
    char buf[64];
    buf[0] = 0;
    if (code == 0) {
        strcpy(buf, "alive");
    } else {
        if (code & ORC_MAX_STEP) strcat(buf, "max_step|");
        if (code & ORC_DEAD) strcat(buf, "dead|");
        if (code & ORC_SUCCESS) strcat(buf, "success|");
        if (code & ORC_LOST_LIFE) strcat(buf, "lost_life|");
        size_t n = strlen(buf);
        if (n > 0) buf[n - 1] = 0;
    }
    int n = (int)strlen(buf);
    if (out && cap > n) memcpy(out, buf, (size_t)n + 1);
    return n;
}

/* position-weighted observation checksum used by the large-batch parity tests */
uint64_t orc_obs_checksum(const void *obs, size_t n_bytes) {
    const uint8_t *b = (const uint8_t *)obs;
    uint64_t acc = 0;
    for (size_t i = 0; i < n_bytes; ++i) acc += (uint64_t)b[i] * ((uint64_t)(i + 1) * 0x9E3779B97F4A7C15ULL);
    return acc;
}
