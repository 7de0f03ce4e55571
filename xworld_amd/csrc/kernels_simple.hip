// kernels_simple.hip -- SimpleGame and SimpleRace as lock-step data-parallel HIP (gfx950).
//
// Replaces, for a whole batch of environments per launch:
//   SimpleGameEngine::reset_game/act/get_reward/get_screen   games/simple_game/simple_game_simulator.cpp:31-76
//   SimpleGame::take_action/game_over                        :92-103
//   RaceEngine::reset_game/act/get_reward/get_screen         games/simple_race/simple_race_simulator.cpp:267-341,386-430
//   BaseCar::move, StraightTrack::*, CircleTrack::*          :52-101,182-243
//   GameSimulator::take_actions/make_context_screens         simulator.cpp:51-108
//
// Layout: structure-of-arrays in HBM, env index fastest; one lane owns one env for the
// state transition, then the workgroup's 256 envs write their observation rows
// cooperatively so that consecutive lanes store consecutive 16-byte chunks.
//
// Build with -ffp-contract=off: SimpleRace's float state must see the same
// float/double rounding points as the reference (no FMA contraction).
#include "xwb_common.h"
#include "../../include/xwb_trig.h"
#include "../../include/xwb_minstd.h"

namespace xwb {

// Envs reset by this launch: each workgroup stores its own count (xwb_done_count adds them up) -- no atomics.  (One atomic
// per wavefront per step on a shared counter was 90 % of a fused SimpleRace launch, and one per wavefront per launch still
// made the reset_done pass of the example loop a 13 us kernel: under a random policy nearly every wavefront holds an env
// that ends at any given step, and L2 serialises same-address atomics.)  Every thread of the workgroup must call it.
__device__ __forceinline__ void store_reset_count(int32_t *partial, int n_reset) {
    __shared__ int s_cnt[4];
    if (!partial) return;                                   // (kernel argument: uniform)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_reset += __shfl_down(n_reset, off);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = n_reset;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ============================================================ SimpleGame ====
static constexpr float SG_MOVE_REWARD = -0.1f;   // simple_game_simulator.h:52
static constexpr float SG_DEST_REWARD = 4.0f;    // simple_game_simulator.h:53

// SimpleGameEngine::get_reward (cpp:69-76) with the two non-zero entries of `_rewards`
// (cpp:36-37: rewards[N-1] = 2 is written first, rewards[0] = 4 second) kept as two
// "already consumed" bits.
__device__ __forceinline__ float sg_get_reward(int pos, int A, uint32_t &flags) {
    float r = SG_MOVE_REWARD;
    if (pos == 0) {
        if (!(flags & 1u)) { r = SG_DEST_REWARD; flags |= 1u; }
    } else if (pos == A - 1) {
        if (!(flags & 2u)) { r = SG_DEST_REWARD / 2; flags |= 2u; }
    }
    return r;
}

__device__ __forceinline__ bool sg_over(int pos, int A) { return pos <= 0 || pos >= A - 1; }

template <int G> struct ChunkT;
template <> struct ChunkT<16> { using type = uint4; };
template <> struct ChunkT<4>  { using type = uint32_t; };
template <> struct ChunkT<1>  { using type = uint8_t; };

template <int G>
__device__ __forceinline__ typename ChunkT<G>::type sg_onehot_chunk(int off);
template <> __device__ __forceinline__ uint4 sg_onehot_chunk<16>(int off) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (off >= 0 && off < 16) {
        uint32_t w = 1u << ((off & 3) * 8);
        int q = off >> 2;
        v.x = q == 0 ? w : 0; v.y = q == 1 ? w : 0; v.z = q == 2 ? w : 0; v.w = q == 3 ? w : 0;
    }
    return v;
}
template <> __device__ __forceinline__ uint32_t sg_onehot_chunk<4>(int off) {
    return (off >= 0 && off < 4) ? (1u << (off * 8)) : 0u;
}
template <> __device__ __forceinline__ uint8_t sg_onehot_chunk<1>(int off) { return off == 0 ? 1 : 0; }

template <int G> __device__ __forceinline__ typename ChunkT<G>::type zero_chunk();
template <> __device__ __forceinline__ uint4 zero_chunk<16>() { return make_uint4(0, 0, 0, 0); }
template <> __device__ __forceinline__ uint32_t zero_chunk<4>() { return 0u; }
template <> __device__ __forceinline__ uint8_t zero_chunk<1>() { return 0; }

// One call = one SimulatorInterface::take_actions (or reset_game) for every env.
// FAST = a step call under the built-in policy with context 1: nothing in the body loads from memory, so the compiler
// has no reason to drain the outstanding stores (s_waitcnt vmcnt(0)) between the steps of a fused launch.
template <int G, bool FAST>
__device__ __forceinline__ void sg_body(const SgParams &p, uint32_t policy_step, int &pos, uint32_t &flags, int &steps,
                                        uint32_t &episode, bool &dirty, int &n_reset) {
    using chunk_t = typename ChunkT<G>::type;
    __shared__ int s_pos[256];      // -1: leave this env's observation untouched
    __shared__ uint8_t s_fresh[256];
    const int tid = threadIdx.x;
    const int e = blockIdx.x * 256 + tid;
    const int A = p.array_size;
    int obs_pos = -1;
    bool fresh = false;
    if (e < p.n) {
        bool do_reset = false;
        {                                                   // (the reset modes have a kernel of their own: sg_reset_kernel)
            int a = (FAST || !p.actions) ? policy_action(p.policy_seed, p.env_gid0 + (uint32_t)e, policy_step, 2) : p.actions[e];
            p.actions_out[e] = a;
            if (a == ACTION_SKIP) {
                // this env does not take part in the call
            } else if ((unsigned)a >= 2u) {                 // CHECK_LT(action_id, _legal_actions.size())
                atomicAdd(p.err_count, 1);
            } else {
                steps += 1;                                 // GameSimulator::take_actions: num_steps_++ once
                float r = 0.0f;
                for (int i = 0; i < p.act_rep; ++i) {
                    // SimpleGameEngine::act, cpp:44-63
                    if (!sg_over(pos, A)) pos += (a == 0) ? -1 : 1;
                    r += sg_get_reward(pos, A, flags);
                }
                float rr = 0.0f; rr += r;                   // SimulatorInterface::take_actions: r = 0; r += ...
                int code = ((p.max_steps > 0 && steps >= p.max_steps) ? MAX_STEP : ALIVE) |
                           (sg_over(pos, A) ? SUCCESS : ALIVE);
                p.reward[e] = rr;
                p.done[e] = (uint8_t)code;
                if (p.packed) p.packed[e] = make_float2(rr, (float)code);
                obs_pos = pos;
                if (p.auto_reset && code != ALIVE) do_reset = true;
            }
        }
        if (do_reset) {
            // SimpleGameEngine::reset_game cpp:31-38 ; GameSimulator::reset_game (the terminal code stays for the caller)
            pos = A / 2; flags = 0; steps = 0;
            episode += 1;
            obs_pos = pos; fresh = true;
        }
        n_reset += do_reset ? 1 : 0;
        if (obs_pos >= 0) dirty = true;
    }
    s_pos[tid] = obs_pos;
    s_fresh[tid] = fresh ? 1 : 0;
    __syncthreads();

    // observation: [env][context][A] bytes; work item = (env, chunk of one frame); the same lane
    // walks all context frames of its chunk so the ring shift needs no cross-lane ordering.
    // make_context_screens / shift_context (simulator.cpp:51-85): oldest frame first, newest last;
    // init_screen (:110-113): zeros, then one shift.
    const int cpf = A / G;                                   // chunks per frame
    const int base_env = blockIdx.x * 256;
    const int n_here = min(256, p.n - base_env);
    const int ctx = FAST ? 1 : p.context;
    for (int i = tid; i < n_here * cpf; i += 256) {
        int le = i / cpf, j = i - le * cpf;
        int pos = s_pos[le];
        if (pos < 0) continue;
        chunk_t *frame0 = reinterpret_cast<chunk_t *>(p.obs + ((size_t)(base_env + le) * ctx) * A) + j;
        if (s_fresh[le]) {
            for (int f = 0; f + 1 < ctx; ++f) frame0[(size_t)f * cpf] = zero_chunk<G>();
        } else {
            for (int f = 0; f + 1 < ctx; ++f) frame0[(size_t)f * cpf] = frame0[(size_t)(f + 1) * cpf];
        }
        frame0[(size_t)(ctx - 1) * cpf] = sg_onehot_chunk<G>(pos - j * G);
    }
}

// n_steps > 1 (xwb_step_n): consecutive steps under the built-in policy with in-kernel auto-reset, each one writing its
// reward / code / observation like a separate launch would -- one launch instead of n (a 6 MB step is launch-bound)
template <int G, bool FAST>
__global__ __launch_bounds__(256) void sg_kernel(SgParams p) {
    // the env's state stays in registers across the steps of one launch and is written back once
    const int e = blockIdx.x * 256 + threadIdx.x;
    int pos = 0, steps = 0;
    uint32_t flags = 0, episode = 0;
    bool dirty = false;
    int n_reset = 0;
    if (e < p.n) { pos = p.pos[e]; flags = p.flags[e]; steps = p.num_steps[e]; episode = p.episode[e]; }
    for (int it = 0; it < p.n_steps; ++it) {
        sg_body<G, FAST>(p, p.policy_step + (uint32_t)it, pos, flags, steps, episode, dirty, n_reset);   // (p stays in kernel-argument memory: never written)
        __syncthreads();                                   // the shared staging of this step is dead
    }
    if (e < p.n && dirty) { p.pos[e] = pos; p.flags[e] = (uint8_t)flags; p.num_steps[e] = steps; p.episode[e] = episode; }
    store_reset_count(p.reset_partial, n_reset);
}

// reset_game for the envs a mode selects (all / game over / mask), as its own kernel: a reset needs nothing of an env's old
// state but its episode counter, so the lanes load `done` (or the mask) and `episode` in ONE round trip, and lanes that
// do not reset touch nothing else -- the generic kernel in a reset mode loaded and wrote back the whole state of every
// env (the reset_done pass of the example loop then cost more than the step it follows).
template <int G>
__global__ __launch_bounds__(256) void sg_reset_kernel(SgParams p) {
    using chunk_t = typename ChunkT<G>::type;
    __shared__ uint8_t s_reset[256];
    const int tid = threadIdx.x;
    const int e = blockIdx.x * 256 + tid;
    const int A = p.array_size;
    bool do_reset = false;
    uint32_t episode = 0;
    if (e < p.n) {
        episode = p.episode[e];
        do_reset = p.mode == MODE_RESET_ALL || (p.mode == MODE_RESET_DONE ? p.done[e] != 0 : p.mask[e] != 0);
    }
    s_reset[tid] = do_reset ? 1 : 0;
    if (!__syncthreads_or(do_reset)) {                       // nobody in this workgroup: nothing to write
        if (tid == 0 && p.reset_partial) p.reset_partial[blockIdx.x] = 0;
        return;
    }
    const int pos = A / 2;                                   // SimpleGameEngine::reset_game cpp:31-38 ; GameSimulator::reset_game
    if (do_reset) {
        p.pos[e] = pos; p.flags[e] = 0; p.num_steps[e] = 0; p.episode[e] = episode + 1;
        p.done[e] = (uint8_t)(sg_over(pos, A) ? SUCCESS : ALIVE);     // over at once for array_size <= 2; num_steps_ == 0 < max_steps
    }
    // init_screen (simulator.cpp:110-113): zeros, then the first frame last
    const int cpf = A / G, ctx = p.context;
    const int base_env = blockIdx.x * 256;
    const int n_here = min(256, p.n - base_env);
    for (int i = tid; i < n_here * cpf; i += 256) {
        const int le = i / cpf, j = i - le * cpf;
        if (!s_reset[le]) continue;
        chunk_t *frame0 = reinterpret_cast<chunk_t *>(p.obs + ((size_t)(base_env + le) * ctx) * A) + j;
        for (int f = 0; f + 1 < ctx; ++f) frame0[(size_t)f * cpf] = zero_chunk<G>();
        frame0[(size_t)(ctx - 1) * cpf] = sg_onehot_chunk<G>(pos - j * G);
    }
    store_reset_count(p.reset_partial, do_reset ? 1 : 0);
}

hipError_t launch_simple_game(const SgParams &p, hipStream_t s) {
    dim3 grid((p.n + 255) / 256), block(256);
    const bool fast = p.mode == MODE_STEP && !p.actions && p.context == 1;
#define SG_LAUNCH(GV) do { if (p.mode != MODE_STEP) hipLaunchKernelGGL((sg_reset_kernel<GV>), grid, block, 0, s, p); \
                           else if (fast) hipLaunchKernelGGL((sg_kernel<GV, true>), grid, block, 0, s, p); \
                           else hipLaunchKernelGGL((sg_kernel<GV, false>), grid, block, 0, s, p); } while (0)
    if (p.array_size % 16 == 0) SG_LAUNCH(16);
    else if (p.array_size % 4 == 0) SG_LAUNCH(4);
    else SG_LAUNCH(1);
#undef SG_LAUNCH
    return hipGetLastError();
}

// ============================================================ SimpleRace ====
#define RACE_PI 3.1415926       // simple_race_simulator.h:39 (double literal)

struct RaceCar { float x, y, angle; };

// cv::norm(Point2f) -> double
__device__ __forceinline__ double race_norm(float x, float y) {
    return sqrt((double)x * x + (double)y * y);
}

// StraightTrack::out_of_bound cpp:182-186 ; CircleTrack::out_of_bound cpp:75-79
__device__ __forceinline__ bool race_oob(const RaceParams &p, float x, float y) {
    if (p.track_type == 1) {
        float r = (float)race_norm(x - p.center_x, y - p.center_y);
        return r < p.inner_radius || r > p.outer_radius;
    }
    return (x < p.mid_x - p.width / 2) || (x > p.mid_x + p.width / 2) || (y < p.start_y) || (y > p.end_y);
}

// race_finish: StraightTrack cpp:188-190 ; Track default false
__device__ __forceinline__ bool race_finish(const RaceParams &p, float y) {
    return p.track_type == 1 ? false : (y > p.end_y);
}

// horizontal_displacement: straight cpp:202-204, circle cpp:92-95
__device__ __forceinline__ float race_h_disp(const RaceParams &p, float x, float y) {
    if (p.track_type == 1)
        return (float)((2 * race_norm(x - p.center_x, y - p.center_y) - (double)p.inner_radius -
                        (double)p.outer_radius) / (double)p.width);
    return 2 * (x - p.mid_x) / p.width;
}

// vertical_displacement: straight cpp:210-212 ; Track default 0
__device__ __forceinline__ float race_v_disp(const RaceParams &p, float y) {
    return p.track_type == 1 ? 0.0f : 2 * (y - p.mid_y) / p.length;
}

// get_tangent_vec: straight cpp:218-220, circle cpp:101-104
__device__ __forceinline__ void race_tangent(const RaceParams &p, float x, float y, float &tx, float &ty) {
    if (p.track_type == 1) {
        float ux = p.center_y - y, uy = x - p.center_x;
        double s = 1 / race_norm(ux, uy);
        tx = (float)((double)ux * s);
        ty = (float)((double)uy * s);
    } else {
        tx = 0.0f; ty = 1.0f;
    }
}

// RaceEngine::get_screen, cpp:412-430.  (ca, sa) = cos / sin of the car's angle as doubles: the reference evaluates
// cos(angle) and sin(angle) twice here and twice more in BaseCar::move / get_reward, always of the same float angle.
__device__ __forceinline__ float4 race_screen(const RaceParams &p, const RaceCar &c, double ca, double sa) {
    float tx, ty;
    race_tangent(p, c.x, c.y, tx, ty);
    double d = (double)tx * ca + (double)ty * sa;
    float cos_theta = (float)fmax(-1.0, fmin(1.0, d));
    float sin_theta = (float)sqrt((double)(1 - cos_theta * cos_theta));
    if (ca * (double)ty + sa * (double)tx < 0) sin_theta = -sin_theta;
    return make_float4(cos_theta, sin_theta, race_h_disp(p, c.x, c.y), race_v_disp(p, c.y));
}

// RaceEngine::reset_game cpp:267-284 ; draws (random mode): track, start-pos #1, start-pos #2, angle
__device__ __forceinline__ void race_reset(const RaceParams &p, RaceCar &c, uint32_t gid, uint32_t episode, uint32_t *engine) {
    if (!p.random) {
        if (p.track_type == 1) {           // CircleTrack::get_start_pos cpp:81-85
            c.x = (p.inner_radius + p.width / 2) + p.center_x;
            c.y = 0.0f + p.center_y;
        } else {                           // StraightTrack::get_start_pos cpp:192-195
            c.x = p.start_x; c.y = p.start_y;
        }
        c.angle = (float)(RACE_PI / 2);    // BaseCar::set_angle(false)
        return;
    }
    float u_track, u_a, u_b, u_ang;
    if (engine) {
        // XWB_RNG_MINSTD: util::get_rand_range_val(1.0) four times from this env's engine, in the reference's call order
        uint32_t x = *engine;
        u_track = xwb_minstd_rand_range_state(&x, 1.0f);
        u_a = xwb_minstd_rand_range_state(&x, 1.0f);
        u_b = xwb_minstd_rand_range_state(&x, 1.0f);
        u_ang = xwb_minstd_rand_range_state(&x, 1.0f);
        *engine = x;
    } else {
        Stream s;
        s.init(p.seed, gid, episode, 0);
        u_track = s.unit();
        u_a = s.unit(); u_b = s.unit(); u_ang = s.unit();
    }
    (void)u_track;                         // one track in the pool -> index 0
    if (p.track_type == 1) {               // cpp:86-89
        float theta = (float)((double)(u_a * 2) * RACE_PI);
        float r = p.inner_radius + u_b * p.width;
        double ct, st;
        xwb_sincos((double)theta, &st, &ct);
        float qx = (float)((double)r * ct);
        float qy = (float)((double)r * st);
        c.x = qx + p.center_x; c.y = qy + p.center_y;
    } else {                               // cpp:196-199
        float dy = u_a * p.length / 2;
        float dx = (float)(((double)u_b - 0.5) * (double)p.width);
        c.x = dx + p.start_x; c.y = dy + p.start_y;
    }
    c.angle = (float)((double)(u_ang * 2) * RACE_PI);   // BaseCar::set_angle(true) cpp:237-243
}

// per-lane state of one env, kept in registers across the steps of one launch
struct RaceLane {
    RaceCar c;
    int steps;
    uint32_t episode;
    double ca, sa;           // cos / sin of c.angle, valid when `trig`
    bool trig, dirty;
    int n_reset;
};

// One SimulatorInterface::take_actions (or reset_game) for this lane's env.  Outputs (reward, code, action, frame) go to
// HBM here; the car itself stays in `L` (race_kernel writes it back once per launch).
template <bool FAST>        // see sg_body
__device__ __forceinline__ void race_body(const RaceParams &p, uint32_t policy_step, int e, RaceLane &L) {
    RaceCar &c = L.c;
    bool do_reset = false, touched = false;
    {                                                       // (the reset modes have a kernel of their own: race_reset_kernel)
        int a = (FAST || !p.actions) ? policy_action(p.policy_seed, p.env_gid0 + (uint32_t)e, policy_step, p.n_legal) : p.actions[e];
        p.actions_out[e] = a;
        if (a == ACTION_SKIP) {
            // this env does not take part in the call
        } else if ((unsigned)a >= (unsigned)p.n_legal) {
            atomicAdd(p.err_count, 1);
        } else {
            // _legal_actions[action_id], cpp:474.  The set is {4, 7} or 0..8 (race_setup): selected arithmetically rather than
            // by a dynamically indexed p.legal[a], which is a vector load from the kernel-argument buffer
            const int action = p.n_legal == 9 ? a : (a == 0 ? p.legal[0] : p.legal[1]);
            L.steps += 1;
            float reward = 0.0f;
            // RaceEngine::act cpp:290-341: the action decodes to the same (d_forward, d_turn) on every repeat
            float d_forward = 0.0f, d_turn = 0.0f;
            {
                int id = action, m = id % 3;
                if (m == 1) d_forward = p.delta_fwd; else if (m == 2) d_forward = -p.delta_fwd;
                id /= 3;
                m = id % 3;
                if (m == 1) d_turn = p.delta_ang; else if (m == 2) d_turn = -p.delta_ang;
            }
            for (int i = 0; i < p.act_rep; ++i) {
                // BaseCar::move cpp:227-235
                const float a0 = c.angle;
                c.angle += d_turn;
                if ((double)c.angle > 2 * RACE_PI) c.angle = (float)((double)c.angle - 2 * RACE_PI);
                else if (c.angle < 0) c.angle = (float)((double)c.angle + 2 * RACE_PI);
                if (!L.trig || c.angle != a0) xwb_sincos((double)c.angle, &L.sa, &L.ca);   // one evaluation per new angle
                L.trig = true;
                float dirx = (float)L.ca, diry = (float)L.sa;
                float sx = d_forward * dirx, sy = d_forward * diry;
                c.x += sx; c.y += sy;
                // RaceEngine::get_reward cpp:386-410
                float tx, ty;
                race_tangent(p, c.x, c.y, tx, ty);
                float vx = dirx, vy = diry;        // cos(angle), sin(angle) narrowed to float again
                float reward_speed = (vx * tx + vy * ty) * d_forward;
                float reward_finish = race_finish(p, c.y) ? 2.0f : 0.0f;
                float reward_boundary;
                if (!p.difficulty_hard) reward_boundary = (float)(-fabs((double)race_h_disp(p, c.x, c.y)));
                else reward_boundary = (race_oob(p, c.x, c.y) && !race_finish(p, c.y)) ? -2.0f : 0.0f;
                float rwd = reward_finish + reward_boundary + reward_speed;
                reward += (float)((double)rwd * p.reward_scale);
            }
            float rr = 0.0f; rr += reward;
            int code = ((p.max_steps > 0 && L.steps >= p.max_steps) ? MAX_STEP : ALIVE) |
                       (race_oob(p, c.x, c.y) ? DEAD : ALIVE);
            p.reward[e] = rr;
            p.done[e] = (uint8_t)code;
            if (p.packed) p.packed[e] = make_float2(rr, (float)code);
            touched = true;
            if (p.auto_reset && code != ALIVE) do_reset = true;
        }
    }
    if (do_reset) {
        L.episode += 1;
        race_reset(p, c, p.env_gid0 + (uint32_t)e, L.episode, (!FAST && p.minstd) ? p.minstd + e : nullptr);
        L.trig = false;
        L.steps = 0;
        touched = true;
    }
    L.n_reset += do_reset ? 1 : 0;
    if (!touched) return;
    L.dirty = true;
    if (!L.trig) { xwb_sincos((double)c.angle, &L.sa, &L.ca); L.trig = true; }
    // make_context_screens: [env][context][4] floats, 16 bytes per frame -> one float4 per lane
    const int ctx = FAST ? 1 : p.context;
    float4 *frames = reinterpret_cast<float4 *>(p.obs) + (size_t)e * ctx;
    if (do_reset) {
        for (int f = 0; f + 1 < ctx; ++f) frames[f] = make_float4(0, 0, 0, 0);
    } else {
        for (int f = 0; f + 1 < ctx; ++f) frames[f] = frames[f + 1];
    }
    frames[ctx - 1] = race_screen(p, c, L.ca, L.sa);
}

template <bool FAST>
__global__ __launch_bounds__(256) void race_kernel(RaceParams p) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool live = e < p.n;
    RaceLane L;
    L.trig = false; L.dirty = false; L.ca = L.sa = 0; L.n_reset = 0;
    L.c.x = L.c.y = L.c.angle = 0; L.steps = 0; L.episode = 0;
    if (live) { L.c.x = p.x[e]; L.c.y = p.y[e]; L.c.angle = p.angle[e]; L.steps = p.num_steps[e]; L.episode = p.episode[e]; }
    for (int it = 0; it < p.n_steps; ++it) {               // n_steps > 1: xwb_step_n, see sg_kernel
        if (live) race_body<FAST>(p, p.policy_step + (uint32_t)it, e, L);
    }
    if (live && L.dirty) { p.x[e] = L.c.x; p.y[e] = L.c.y; p.angle[e] = L.c.angle; p.num_steps[e] = L.steps; p.episode[e] = L.episode; }
    store_reset_count(p.reset_partial, L.n_reset);
}

// reset_game for the envs a mode selects, as its own kernel (see sg_reset_kernel): `done` / mask and `episode` in one round
// trip; lanes that do not reset leave at once, the others write the new car, counters, code and first frame.
__global__ __launch_bounds__(256) void race_reset_kernel(RaceParams p) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    bool do_reset = false;
    uint32_t episode = 0;
    if (e < p.n) {
        episode = p.episode[e];
        do_reset = p.mode == MODE_RESET_ALL || (p.mode == MODE_RESET_DONE ? p.done[e] != 0 : p.mask[e] != 0);
    }
    if (!__syncthreads_or(do_reset)) {                       // nobody in this workgroup
        if (threadIdx.x == 0 && p.reset_partial) p.reset_partial[blockIdx.x] = 0;
        return;
    }
    if (do_reset) {
        RaceCar c;
        episode += 1;
        race_reset(p, c, p.env_gid0 + (uint32_t)e, episode, p.minstd ? p.minstd + e : nullptr);
        p.x[e] = c.x; p.y[e] = c.y; p.angle[e] = c.angle; p.num_steps[e] = 0; p.episode[e] = episode;
        p.done[e] = (uint8_t)(race_oob(p, c.x, c.y) ? DEAD : ALIVE);        // game_over() right after the reset
        double sa, ca;
        xwb_sincos((double)c.angle, &sa, &ca);
        // init_screen: [env][context][4] floats; older frames zero, the first frame last
        float4 *frames = reinterpret_cast<float4 *>(p.obs) + (size_t)e * p.context;
        for (int f = 0; f + 1 < p.context; ++f) frames[f] = make_float4(0, 0, 0, 0);
        frames[p.context - 1] = race_screen(p, c, ca, sa);
    }
    store_reset_count(p.reset_partial, do_reset ? 1 : 0);
}

hipError_t launch_simple_race(const RaceParams &p, hipStream_t s) {
    dim3 grid((p.n + 255) / 256), block(256);
    if (p.mode != MODE_STEP) hipLaunchKernelGGL(race_reset_kernel, grid, block, 0, s, p);
    else if (!p.actions && p.context == 1 && !p.minstd) hipLaunchKernelGGL(race_kernel<true>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(race_kernel<false>, grid, block, 0, s, p);
    return hipGetLastError();
}

}  // namespace xwb
