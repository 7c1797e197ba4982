// hx_env.hip — batched pursuit-lock-launch env step for gfx950 (MI355X).
//
// One thread per env, struct-of-arrays fp32 state in HBM (word w of env i at state[w*stride + i]: every load and
// store of a wave is one contiguous 256-B segment), observation / replay-row tiles staged through LDS so that the
// row-major outputs leave the CU as full-line 16-B-per-lane stores, wave ballots for the store mask and the
// episode statistics, one ring-head atomic per workgroup.
//
// What it replaces (reference file:line):
//   HarfangEnv.step          hirl/environments/HarfangEnv_GYM.py:83-90   (E9)
//   _apply_action            :139-158 (E4)      scripted opponents :342-353, :412-421 (E10, E11)
//   UPDATE_SCENE tick        external Harfang simulator (E5) — re-derived model, docs/DYNAMICS.md
//   _get_observation         :193-268 (E6)      _get_reward :101-137 (E7)      _get_termination :160-169 (E8)
//   reset / random_reset     :34-81, :171-188, :374-406, :440-474 (E2, E3)
//   UniformMemory.store      hirl/utils/buffer.py:20-36 (U6), fused; episode rules train_all.py:341-361 (D1)
//   get_reward/get_termination for expert labelling :299-336 (E13)
//
// Numerics: this file is compiled with -ffp-contract=off; state-evolving arithmetic uses only + - * / sqrt in a
// fixed order, so masks are reproducible bit for bit.  asinf/atan2f/acosf appear only in the observation.
#include "hx_common.h"
#include <hip/hip_ext.h>

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

// ---- constants of the re-derived model (docs/DYNAMICS.md) ------------------------------------------------
constexpr float kDt = 1.0f / 60.0f;  // one UPDATE_SCENE: hirl/data/straight_line/ai_env.py:18
constexpr float kSlew = 0.05f;
constexpr float kRho0Half = 0.6125f;
constexpr float kLapse = 2.2558e-5f;
constexpr float kQHalf = 4000.0f;
constexpr float kPitchRate = 0.8f, kYawRate = 0.4f, kRollRate = 2.5f, kLeveller = 0.8f, kEasy = 0.2f;
constexpr float kCdX = 0.05f, kCdY = 0.02f, kCdZ = 5.2e-4f, kCl0 = 3.8e-4f, kTMax = 20.0f, kGrav = 9.8f;
constexpr float kCosLock = 0.9659258f, kLockMin = 100.0f, kLockMax = 3000.0f, kLockDelay = 1.0f;
constexpr float kMBoost = 50.0f, kMTurn = 0.15f, kMAcc = 300.0f, kMVmax = 1000.0f, kMLife = 20.0f;
constexpr float kMHit2 = 1600.0f, kMDamage = 0.3f;
constexpr float kPi = 3.14159265358979323846f, kRad2Deg = 57.29577951308232f;

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Plane {
    V3 p, v;
    float qw, qx, qy, qz;
    float lp, lr, ly;  // pitch, roll, yaw levels
};

struct Axes {
    V3 X, Y, Z;
};

__device__ __forceinline__ Axes quat_axes(float w, float x, float y, float z) {
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float wx = w * x, wy = w * y, wz = w * z;
    Axes a;
    a.X = {1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz), 2.0f * (xz - wy)};
    a.Y = {2.0f * (xy - wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx)};
    a.Z = {2.0f * (xz + wy), 2.0f * (yz - wx), 1.0f - 2.0f * (xx + yy)};
    return a;
}

__device__ __forceinline__ float slew(float level, float cmd) { return level + clampf(cmd - level, -kSlew, kSlew); }

// one tick of one aircraft; cmd = (pitch, roll, yaw) levels
__device__ __forceinline__ void plane_tick(Plane& P, float cp, float cr, float cy, float thrust) {
    P.lp = slew(P.lp, cp);
    P.lr = slew(P.lr, cr);
    P.ly = slew(P.ly, cy);
    const Axes A = quat_axes(P.qw, P.qx, P.qy, P.qz);
    const float h = clampf(P.p.y, 0.0f, 30000.0f);
    float x = 1.0f - kLapse * h;
    x = x < 0.05f ? 0.05f : x;
    const float x2 = x * x;
    const float sigma = (x2 * x2) * sqrtf(sqrtf(x));
    const float hr = kRho0Half * sigma;
    const float s2 = dot3(P.v, P.v);
    const float s = sqrtf(s2);
    const float qd = hr * s2;
    const float eff = qd / (qd + kQHalf);
    const float wx = (kPitchRate * P.lp) * eff;
    const float wy = (kYawRate * P.ly) * eff;
    const float wz = (kRollRate * P.lr) * eff - kLeveller * A.X.y;
    const float Om = -((kEasy * A.X.y) * eff);
    {
        const float qw = P.qw, qx = P.qx, qy = P.qy, qz = P.qz;
        float dw = -(qx * wx + qy * wy) - qz * wz;
        float dx = (qw * wx + qy * wz) - qz * wy;
        float dy = (qw * wy + qz * wx) - qx * wz;
        float dz = (qw * wz + qx * wy) - qy * wx;
        dw = dw - Om * qy;
        dx = dx + Om * qz;
        dy = dy + Om * qw;
        dz = dz - Om * qx;
        const float hdt = 0.5f * kDt;
        const float nw = qw + hdt * dw, nx = qx + hdt * dx, ny = qy + hdt * dy, nz = qz + hdt * dz;
        const float n = sqrtf(((nw * nw + nx * nx) + ny * ny) + nz * nz);
        P.qw = nw / n;
        P.qx = nx / n;
        P.qy = ny / n;
        P.qz = nz / n;
    }
    const float vbx = dot3(P.v, A.X), vby = dot3(P.v, A.Y), vbz = dot3(P.v, A.Z);
    const float k = hr * s;
    const float fx = -((kCdX * k) * vbx);
    const float fy = (kCl0 * hr) * (vbz * vbz) - (kCdY * k) * vby;
    const float fz = kTMax * thrust - (kCdZ * k) * vbz;
    const float ax = (A.X.x * fx + A.Y.x * fy) + A.Z.x * fz;
    float ay = (A.X.y * fx + A.Y.y * fy) + A.Z.y * fz;
    const float az = (A.X.z * fx + A.Y.z * fy) + A.Z.z * fz;
    ay = ay - kGrav;
    P.v.x = P.v.x + ax * kDt;
    P.v.y = P.v.y + ay * kDt;
    P.v.z = P.v.z + az * kDt;
    P.p.x = P.p.x + P.v.x * kDt;
    P.p.y = P.p.y + P.v.y * kDt;
    P.p.z = P.p.z + P.v.z * kDt;
}

struct Env {
    Plane ally, opp;
    V3 mp, mv;
    float health, lock_timer, missile_age;
    uint32_t flags, counters;
};

// State words are visited in order through a cursor whose base is UNIFORM (block base + w * stride lives in SGPRs, two scalar
// adds per word) and whose per-lane part is the 32-bit thread index: every access is `global_load/store v, v_tid, s[base]` — no
// 64-bit multiply-add per word per lane, which used to be a quarter of the kernel's instructions.
struct RCursor {
    const float* __restrict__ p;  // uniform
    int64_t stride;
    uint32_t lane;
    __device__ __forceinline__ float next() {
        const float v = p[lane];
        p += stride;
        return v;
    }
};
struct WCursor {
    float* __restrict__ p;  // uniform
    int64_t stride;
    uint32_t lane;
    __device__ __forceinline__ void put(float v) {
        p[lane] = v;
        p += stride;
    }
};
__device__ __forceinline__ void load_plane(Plane& P, RCursor& c) {
    P.p.x = c.next(); P.p.y = c.next(); P.p.z = c.next();
    P.v.x = c.next(); P.v.y = c.next(); P.v.z = c.next();
    P.qw = c.next(); P.qx = c.next(); P.qy = c.next(); P.qz = c.next();
    P.lp = c.next(); P.lr = c.next(); P.ly = c.next();
}
__device__ __forceinline__ void store_plane(const Plane& P, WCursor& c) {
    c.put(P.p.x); c.put(P.p.y); c.put(P.p.z);
    c.put(P.v.x); c.put(P.v.y); c.put(P.v.z);
    c.put(P.qw); c.put(P.qx); c.put(P.qy); c.put(P.qz);
    c.put(P.lp); c.put(P.lr); c.put(P.ly);
}
// i = i0 + lane with i0 uniform across the workgroup (blockIdx.x * kBlock)
__device__ __forceinline__ void load_env(Env& E, const float* __restrict__ s, int64_t stride, int64_t i0, uint32_t lane) {
    RCursor c{s + i0, stride, lane};
    load_plane(E.ally, c);
    load_plane(E.opp, c);
    E.mp.x = c.next(); E.mp.y = c.next(); E.mp.z = c.next();
    E.mv.x = c.next(); E.mv.y = c.next(); E.mv.z = c.next();
    E.health = c.next();
    E.lock_timer = c.next();
    E.missile_age = c.next();
    E.flags = __float_as_uint(c.next());
    E.counters = __float_as_uint(c.next());
}
__device__ __forceinline__ void store_env(const Env& E, float* __restrict__ s, int64_t stride, int64_t i0, uint32_t lane) {
    WCursor c{s + i0, stride, lane};
    store_plane(E.ally, c);
    store_plane(E.opp, c);
    c.put(E.mp.x); c.put(E.mp.y); c.put(E.mp.z);
    c.put(E.mv.x); c.put(E.mv.y); c.put(E.mv.z);
    c.put(E.health);
    c.put(E.lock_timer);
    c.put(E.missile_age);
    c.put(__uint_as_float(E.flags));
    c.put(__uint_as_float(E.counters));
}

// ---- Philox4x32-10 (Salmon et al., SC'11) ---------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0;
        c1 = l1;
        c2 = n2;
        c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}
// U{-100..100}: the support of random.randint(-100, 100), HarfangEnv_GYM.py:74
__device__ __forceinline__ float offset201(uint32_t u) { return (float)((int)__umulhi(u, 201u) - 100); }

__device__ __forceinline__ void plane_reset(Plane& P, float x, float y, float z, float speed) {
    P.p = {x, y, z};
    P.v = {0.0f, 0.0f, speed};
    P.qw = 1.0f;
    P.qx = P.qy = P.qz = 0.0f;
    P.lp = P.lr = P.ly = 0.0f;
}

// reset / random_reset: HarfangEnv_GYM.py:34-81 (+ :374-406 serpentine, :440-474 circular)
__device__ __forceinline__ void env_reset(Env& E, uint32_t scenario, bool randomize, uint64_t seed, uint32_t env_id,
                                          uint32_t episode) {
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    if (randomize) {
        uint32_t r[4];
        philox4x32_10(env_id, episode, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        ox = offset201(r[0]);
        oy = offset201(r[1]);
        oz = offset201(r[2]);
    }
    plane_reset(E.opp, 0.0f, 4200.0f, 0.0f, scenario == 2u ? 290.0f : 200.0f);  // :71-72,78 / :472-473
    plane_reset(E.ally, 0.0f + ox, 3500.0f + oy, -4000.0f + oz, 300.0f);        // :73-74,77
    E.mp = {0.0f, 0.0f, 0.0f};
    E.mv = {0.0f, 0.0f, 0.0f};
    E.health = 0.2f;  // :70
    E.lock_timer = 0.0f;
    E.missile_age = 0.0f;
    // latches as reset()'s own _get_observation leaves them: locked_prev = locked = False, slots True
    E.flags = HX_F_SLOT_PREV | HX_F_SLOT | HX_F_SIM_SLOT | (scenario << HX_F_SCEN_SHIFT);
    E.counters = 0u;
}

// what the wrapper reads back and packs: HarfangEnv_GYM.py:193-268.  Also returns distance, altitude and the
// normalised target angle the reward uses (:107-116).
struct Observed {
    float obs[HX_OBS_DIM];
    float loc_diff, target_angle, altitude;
};

__device__ __forceinline__ void euler_of(const Plane& P, float& pitch, float& heading, float& roll) {
    const Axes A = quat_axes(P.qw, P.qx, P.qy, P.qz);
    pitch = asinf(clampf(-A.Z.y, -1.0f, 1.0f));
    heading = atan2f(A.Z.x, A.Z.z);
    roll = atan2f(A.X.y, A.Y.y);
}

__device__ __forceinline__ void observe(const Env& E, Observed& O) {
    const V3 d = {E.ally.p.x - E.opp.p.x, E.ally.p.y - E.opp.p.y, E.ally.p.z - E.opp.p.z};
    O.obs[0] = d.x / 10000.0f;
    O.obs[1] = d.y / 10000.0f;
    O.obs[2] = d.z / 10000.0f;
    float p, h, r;
    euler_of(E.ally, p, h, r);
    O.obs[3] = p / kPi;
    O.obs[4] = h / kPi;
    O.obs[5] = r / kPi;
    const Axes A = quat_axes(E.ally.qw, E.ally.qx, E.ally.qy, E.ally.qz);
    const V3 to = {E.opp.p.x - E.ally.p.x, E.opp.p.y - E.ally.p.y, E.opp.p.z - E.ally.p.z};
    const float dist = sqrtf(dot3(to, to));
    const float cosang = clampf(dot3(A.Z, to) / dist, -1.0f, 1.0f);
    const float angle_deg = acosf(cosang) * kRad2Deg;
    O.target_angle = angle_deg / 180.0f;
    O.obs[6] = O.target_angle;
    O.obs[7] = (E.flags & HX_F_LOCKED) ? 1.0f : -1.0f;
    O.obs[8] = (E.flags & HX_F_SLOT) ? 1.0f : -1.0f;
    euler_of(E.opp, p, h, r);
    O.obs[9] = p / kPi;
    O.obs[10] = h / kPi;
    O.obs[11] = r / kPi;
    O.obs[12] = E.health;
    O.loc_diff = sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z);
    O.altitude = E.ally.p.y;
}

// E4 + E10/E11 + E5: apply the action, script the opponent, one simulator tick
__device__ __forceinline__ void sim_step(Env& E, float a0, float a1, float a2, bool fire) {
    // scripted opponent  HarfangEnv_GYM.py:145-147 / :342-353 / :412-421
    float op = 0.0f, orl = 0.0f, oy = 0.0f;
    {
        uint32_t script = E.counters >> 16;
        const uint32_t scen = (E.flags >> HX_F_SCEN_SHIFT) & 3u;
        if (scen == 1u) {
            script += 1u;
            const uint32_t duration = (E.flags & HX_F_SERP_LONG) ? 500u : 250u;
            if (script % duration == 0u) {
                script = 0u;
                E.flags ^= HX_F_SERP_POS;
                E.flags |= HX_F_SERP_LONG;
            }
            oy = (E.flags & HX_F_SERP_POS) ? 0.1f : -0.1f;
        } else if (scen == 2u) {
            if (script < 65535u) script += 1u;
            op = script < 100u ? -0.02f : -0.01f;
            orl = 0.28f;  // the 0.84 sent first never reaches a tick (:415-420)
        }
        E.counters = (E.counters & 0xFFFFu) | (script << 16);
    }
    E.flags = fire ? (E.flags | HX_F_FIRED) : (E.flags & ~HX_F_FIRED);  // now_missile_state :150-156
    // FIRE_MISSILE is handled before the tick, with the lock the simulator holds at that moment
    if (fire && (E.flags & HX_F_SIM_SLOT)) {
        const Axes A = quat_axes(E.ally.qw, E.ally.qx, E.ally.qy, E.ally.qz);
        E.flags &= ~HX_F_SIM_SLOT;
        E.flags |= HX_F_M_ACTIVE;
        E.flags = (E.lock_timer >= kLockDelay) ? (E.flags | HX_F_M_GUIDED) : (E.flags & ~HX_F_M_GUIDED);
        E.mp = E.ally.p;
        E.mv = {E.ally.v.x + A.Z.x * kMBoost, E.ally.v.y + A.Z.y * kMBoost, E.ally.v.z + A.Z.z * kMBoost};
        E.missile_age = 0.0f;
    }
    const uint32_t scen = (E.flags >> HX_F_SCEN_SHIFT) & 3u;
    plane_tick(E.ally, a0, a1, a2, 1.0f);
    plane_tick(E.opp, op, orl, oy, scen == 2u ? 0.8f : 0.6f);
    if (E.flags & HX_F_M_ACTIVE) {
        const V3 to = {E.opp.p.x - E.mp.x, E.opp.p.y - E.mp.y, E.opp.p.z - E.mp.z};
        const float ms = sqrtf(dot3(E.mv, E.mv));
        V3 dir = {E.mv.x / ms, E.mv.y / ms, E.mv.z / ms};
        if (E.flags & HX_F_M_GUIDED) {
            const float dist = sqrtf(dot3(to, to));
            const V3 nd = {dir.x + kMTurn * (to.x / dist - dir.x), dir.y + kMTurn * (to.y / dist - dir.y),
                           dir.z + kMTurn * (to.z / dist - dir.z)};
            const float nn = sqrtf(dot3(nd, nd));
            dir = {nd.x / nn, nd.y / nn, nd.z / nn};
        }
        float ms2 = ms + kMAcc * kDt;
        ms2 = ms2 > kMVmax ? kMVmax : ms2;
        E.mv = {dir.x * ms2, dir.y * ms2, dir.z * ms2};
        E.mp = {E.mp.x + E.mv.x * kDt, E.mp.y + E.mv.y * kDt, E.mp.z + E.mv.z * kDt};
        E.missile_age = E.missile_age + kDt;
        const V3 d = {E.opp.p.x - E.mp.x, E.opp.p.y - E.mp.y, E.opp.p.z - E.mp.z};
        if (dot3(d, d) < kMHit2) {
            const float hl = E.health - kMDamage;
            E.health = hl < 0.0f ? 0.0f : hl;
            E.flags &= ~HX_F_M_ACTIVE;
        } else if (E.missile_age > kMLife) {
            E.flags &= ~HX_F_M_ACTIVE;
        }
    }
    {
        const Axes A = quat_axes(E.ally.qw, E.ally.qx, E.ally.qy, E.ally.qz);
        const V3 d = {E.opp.p.x - E.ally.p.x, E.opp.p.y - E.ally.p.y, E.opp.p.z - E.ally.p.z};
        const float dist = sqrtf(dot3(d, d));
        const float cosang = dot3(A.Z, d) / dist;
        const bool in_cone = (cosang > kCosLock) && (dist > kLockMin) && (dist < kLockMax);
        E.lock_timer = in_cone ? E.lock_timer + kDt : 0.0f;
    }
}

// E6 latches + E7 reward + E8 termination on the post-tick state
__device__ __forceinline__ void wrap_step(Env& E, Observed& O, float& reward, int& success) {
    uint32_t f = E.flags;
    // Ally_target_locked <- n_Ally_target_locked ; n_Ally_target_locked <- read-back    :227-228
    f = (f & ~HX_F_LOCKED_PREV) | ((f & HX_F_LOCKED) ? HX_F_LOCKED_PREV : 0u);
    f = (f & ~HX_F_LOCKED) | ((E.lock_timer >= kLockDelay) ? HX_F_LOCKED : 0u);
    // missile1_state <- n_missile1_state ; n_missile1_state <- slots[0]                  :250-251
    f = (f & ~HX_F_SLOT_PREV) | ((f & HX_F_SLOT) ? HX_F_SLOT_PREV : 0u);
    f = (f & ~HX_F_SLOT) | ((f & HX_F_SIM_SLOT) ? HX_F_SLOT : 0u);
    E.flags = f;
    observe(E, O);
    float r = 0.0f;
    int s = 0;
    r = r - 0.0001f * O.loc_diff;                 // :107
    r = r - O.target_angle * 10.0f;               // :110
    if (O.altitude < 2000.0f) r = r - 4.0f;       // :112-113
    if (O.altitude > 7000.0f) r = r - 4.0f;       // :115-116
    if (f & HX_F_FIRED) {                         // :119-132 — flags latched BEFORE the action
        r = r - 8.0f;
        if ((f & HX_F_SLOT_PREV) && !(f & HX_F_LOCKED_PREV)) {
            s = -1;
        } else if ((f & HX_F_SLOT_PREV) && (f & HX_F_LOCKED_PREV)) {
            s = 1;
            f |= HX_F_FIRE_SUCCESS;
        }
    }
    // `health_level <= 0.1` on the float64 image of an fp32 value == `h < 0.1f` (0.1f rounds above 0.1)  :135
    if (E.health < 0.1f && (f & HX_F_FIRE_SUCCESS)) r = r + 600.0f;
    if (O.altitude < 500.0f || O.altitude > 10000.0f) f |= HX_F_DONE;          // :163-164
    if (E.health <= 0.0f) f |= HX_F_DONE | HX_F_EPISODE_SUCCESS;                // :165-167
    E.flags = f;
    reward = r;
    success = s;
}

struct StepArgs {
    float* state;
    int64_t n, stride;
    const float* actions;
    float* obs_io;
    float* reward;
    uint8_t* done;
    int8_t* success;
    HxStepOpts o;
};

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Cooperative, fully coalesced copy of `count` floats between a row-major global tile and LDS.
__device__ __forceinline__ void tile_load(float* __restrict__ lds, const float* __restrict__ g, int count) {
    for (int e = threadIdx.x; e < count; e += kBlock) lds[e] = g[e];
}
__device__ __forceinline__ void tile_store(float* __restrict__ g, const float* __restrict__ lds, int count) {
    for (int e = threadIdx.x; e < count; e += kBlock) g[e] = lds[e];
}

template <bool INSERT>
__global__ __launch_bounds__(kBlock) void env_step_kernel(StepArgs A) {
    // one LDS object: [obs tile 256*13][row tile 256*33 (INSERT)] + small scratch
    constexpr int kObsTile = kBlock * HX_OBS_DIM;
    constexpr int kRowPitch = HX_ROW_WORDS + 1;  // +1: conflict-free per-lane row writes
    __shared__ float lds[kObsTile + (INSERT ? kBlock * kRowPitch : 0)];
    __shared__ unsigned long long s_base;
    __shared__ int s_wcount[kWaves];
    __shared__ unsigned s_stat[kWaves][HX_STAT_COUNT];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * kBlock;
    const int64_t i = i0 + tid;
    const int nblk = (int)((A.n - i0) < kBlock ? (A.n - i0) : kBlock);
    const bool active = tid < nblk;
    float* s_obs = lds;
    float* s_row = lds + kObsTile;

    if (INSERT) tile_load(s_obs, A.obs_io + i0 * HX_OBS_DIM, nblk * HX_OBS_DIM);

    Env E;
    float4 act = {0.f, 0.f, 0.f, 0.f};
    bool trunc = false, store = false;
    if (active) {
        load_env(E, A.state, A.stride, i0, (uint32_t)tid);
        act = reinterpret_cast<const float4*>(A.actions)[i];
        uint32_t ep = E.counters & 0xFFFFu;
        ep = ep < 65535u ? ep + 1u : ep;
        trunc = A.o.max_step > 0 && (int)ep >= A.o.max_step;  // train_all.py:346-347
        store = INSERT && !trunc;
    }
    // ring slots: ballot -> per-wave rank -> one atomic per workgroup (issued before the arithmetic)
    int rank = 0, nstore = 0;
    if (INSERT) {
        const unsigned long long b = __ballot(store);
        rank = __popcll(b & ((1ull << lane_id()) - 1ull));
        if (lane_id() == 0) s_wcount[wave] = __popcll(b);
    }
    __syncthreads();
    if (INSERT) {
        int before = 0;
        for (int w = 0; w < kWaves; ++w) {
            before += (w < wave) ? s_wcount[w] : 0;
            nstore += s_wcount[w];
        }
        rank += before;
        if (tid == 0 && nstore > 0) s_base = atomicAdd((unsigned long long*)A.o.total, (unsigned long long)nstore);
    }

    float reward = 0.0f;
    int success = 0;
    bool done = false, ended = false;
    Observed O;
    unsigned st_kill = 0, st_fs = 0, st_tl = 0, st_fire = 0, st_good = 0, st_lock = 0;
    if (active) {
        const bool fire = act.w > 0.0f;  // float(action[3] > 0)  HarfangEnv_GYM.py:150
        sim_step(E, act.x, act.y, act.z, fire);
        wrap_step(E, O, reward, success);
        uint32_t ep = E.counters & 0xFFFFu;
        ep = ep < 65535u ? ep + 1u : ep;
        E.counters = (E.counters & 0xFFFF0000u) | ep;
        done = (E.flags & HX_F_DONE) != 0u;
        ended = A.o.auto_reset && (done || trunc);
        st_fire = (E.flags & HX_F_FIRED) ? 1u : 0u;
        st_good = success == 1 ? 1u : 0u;
        st_lock = (E.flags & HX_F_LOCKED) ? 1u : 0u;
        st_kill = (ended && (E.flags & HX_F_EPISODE_SUCCESS)) ? 1u : 0u;
        st_fs = (ended && (E.flags & HX_F_FIRE_SUCCESS)) ? 1u : 0u;
        st_tl = (ended && !done) ? 1u : 0u;
        A.reward[i] = reward;
        A.done[i] = done ? 1 : 0;
        A.success[i] = (int8_t)success;
    }
    if (INSERT) {
        if (store) {
            // row = s[13] a[4] s'[13] r done   (Transition, buffer.py:8; sample() drops step_success :48)
            float* row = s_row + rank * kRowPitch;
            const float* prev = s_obs + tid * HX_OBS_DIM;
#pragma unroll
            for (int j = 0; j < HX_OBS_DIM; ++j) row[j] = prev[j];
            row[13] = act.x;
            row[14] = act.y;
            row[15] = act.z;
            row[16] = act.w;
#pragma unroll
            for (int j = 0; j < HX_OBS_DIM; ++j) row[17 + j] = O.obs[j];
            row[30] = reward;
            row[31] = done ? 1.0f : 0.0f;
        }
    }
    __syncthreads();  // every lane has consumed its previous observation; rows complete; s_base visible
    if (INSERT && store && A.o.ring_success) {
        A.o.ring_success[(s_base + (unsigned long long)rank) % (unsigned long long)A.o.cap] = (int8_t)success;
    }
    if (active) {
        if (ended) {
            const uint32_t scen = (E.flags >> HX_F_SCEN_SHIFT) & 3u;
            const uint32_t epi = A.o.episode_ctr[i] + 1u;
            A.o.episode_ctr[i] = epi;
            env_reset(E, scen, A.o.randomize != 0, A.o.seed, A.o.env_id0 + (uint32_t)i, epi);
            observe(E, O);
        }
        store_env(E, A.state, A.stride, i0, (uint32_t)tid);
        float* out = s_obs + tid * HX_OBS_DIM;
#pragma unroll
        for (int j = 0; j < HX_OBS_DIM; ++j) out[j] = O.obs[j];
    }
    if (A.o.stats) {
        const unsigned vals[HX_STAT_COUNT] = {ended ? 1u : 0u, st_kill, st_fs, st_tl, st_fire, st_good, st_lock, active ? 1u : 0u};
#pragma unroll
        for (int k = 0; k < HX_STAT_COUNT; ++k) {
            const unsigned c = (unsigned)__popcll(__ballot(vals[k] != 0u));
            if (lane_id() == 0) s_stat[wave][k] = c;
        }
    }
    __syncthreads();
    tile_store(A.obs_io + i0 * HX_OBS_DIM, s_obs, nblk * HX_OBS_DIM);
    if (INSERT && nstore > 0) {
        // 16 B per lane, 1 KiB per wave-instruction, rows contiguous in the ring (modulo wrap)
        const unsigned long long base = s_base, cap = (unsigned long long)A.o.cap;
        float4* ring4 = reinterpret_cast<float4*>(A.o.ring);
        for (int e = tid; e < nstore * (HX_ROW_WORDS / 4); e += kBlock) {
            const int r = e >> 3, c = (e & 7) * 4;
            const float* src = s_row + r * kRowPitch + c;
            const float4 v = {src[0], src[1], src[2], src[3]};
            ring4[((base + (unsigned long long)r) % cap) * (HX_ROW_WORDS / 4) + (e & 7)] = v;
        }
    }
    if (A.o.stats && tid < HX_STAT_COUNT) {
        unsigned c = 0;
        for (int w = 0; w < kWaves; ++w) c += s_stat[w][tid];
        if (c) atomicAdd((unsigned long long*)&A.o.stats[tid], (unsigned long long)c);
    }
}

struct ResetArgs {
    float* state;
    int64_t n, stride;
    const uint8_t* mask;
    const int32_t* scenario;
    int32_t scenario_all, randomize;
    uint64_t seed;
    uint32_t env_id0;
    uint32_t* episode_ctr;
    float* obs;
};

__global__ __launch_bounds__(kBlock) void env_reset_kernel(ResetArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= A.n) return;
    if (A.mask && !A.mask[i]) return;
    Env E;
    const uint32_t scen = (uint32_t)(A.scenario ? A.scenario[i] : A.scenario_all);
    const uint32_t epi = A.episode_ctr ? A.episode_ctr[i] : 0u;
    env_reset(E, scen, A.randomize != 0, A.seed, A.env_id0 + (uint32_t)i, epi);
    store_env(E, A.state, A.stride, (int64_t)blockIdx.x * kBlock, threadIdx.x);
    if (A.obs) {
        Observed O;
        observe(E, O);
        for (int j = 0; j < HX_OBS_DIM; ++j) A.obs[i * HX_OBS_DIM + j] = O.obs[j];
    }
}

__global__ __launch_bounds__(kBlock) void env_rearm_kernel(float* state, int64_t n, int64_t stride, const uint8_t* mask) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n || (mask && !mask[i])) return;
    const uint32_t f = __float_as_uint(state[35 * stride + i]) | HX_F_SIM_SLOT;
    state[35 * stride + i] = __uint_as_float(f);
}

// get_reward / get_termination  HarfangEnv_GYM.py:299-336
__global__ __launch_bounds__(kBlock) void label_kernel(const float* __restrict__ s, const float* __restrict__ a,
                                                       const float* __restrict__ ns, int64_t n, float* reward,
                                                       int8_t* success, uint8_t* done) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* S = s + i * HX_OBS_DIM;
    const float* N = ns + i * HX_OBS_DIM;
    const float x = N[0] * 10000.0f, y = N[1] * 10000.0f, z = N[2] * 10000.0f;
    const float loc = sqrtf((x * x + y * y) + z * z);
    float r = 0.0f;
    int sc = 0;
    r = r - 0.0001f * loc;
    r = r - N[6] * 10.0f;
    if (a[i * HX_ACT_DIM + 3] > 0.0f) {
        r = r - 8.0f;
        if (S[8] > 0.0f && S[7] < 0.0f) sc = -1;
        else if (S[8] > 0.0f && S[7] > 0.0f) sc = 1;
    }
    // float64 comparisons against 0.1 on fp32 data: `< 0.1` and `<= 0.1` both equal `h < 0.1f`
    const bool low = N[12] < 0.1f;
    if (low) r = r + 600.0f;
    reward[i] = r;
    success[i] = (int8_t)sc;
    done[i] = low ? 1 : 0;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

extern "C" {

int hx_env_reset(float* state, int64_t n, int64_t stride, const uint8_t* mask, const int32_t* scenario,
                 int32_t scenario_all, int32_t randomize, uint64_t seed, uint32_t env_id0, uint32_t* episode_ctr,
                 float* obs, void* stream) {
    HX_REQUIRE(state && n > 0 && stride >= n, "hx_env_reset: bad state/n/stride");
    HX_REQUIRE(scenario || (scenario_all >= 0 && scenario_all <= 2), "hx_env_reset: scenario must be 0..2");
    ResetArgs A{state, n, stride, mask, scenario, scenario_all, randomize, seed, env_id0, episode_ctr, obs};
    hipLaunchKernelGGL(env_reset_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, A);
    HX_CHECK_LAUNCH("hx_env_reset");
    return 0;
}

int hx_env_step(float* state, int64_t n, int64_t stride, const float* actions, float* obs_io, float* reward,
                uint8_t* done, int8_t* success, const HxStepOpts* opts, void* stream) {
    HX_REQUIRE(state && actions && obs_io && reward && done && success, "hx_env_step: null buffer");
    HX_REQUIRE(n > 0 && stride >= n, "hx_env_step: bad n/stride");
    HX_REQUIRE((reinterpret_cast<uintptr_t>(actions) & 15u) == 0, "hx_env_step: actions must be 16-byte aligned");
    StepArgs A{state, n, stride, actions, obs_io, reward, done, success, HxStepOpts{}};
    if (opts) A.o = *opts;
    HX_REQUIRE(!A.o.auto_reset || A.o.episode_ctr, "hx_env_step: auto_reset needs episode_ctr");
    if (A.o.ring) {
        HX_REQUIRE(A.o.cap > 0 && A.o.total, "hx_env_step: ring needs cap and total");
        HX_REQUIRE((reinterpret_cast<uintptr_t>(A.o.ring) & 15u) == 0, "hx_env_step: ring must be 16-byte aligned");
        if (A.o.ev_start && A.o.ev_stop)
            hipExtLaunchKernelGGL(env_step_kernel<true>, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                                  (hipEvent_t)A.o.ev_start, (hipEvent_t)A.o.ev_stop, 0, A);
        else
            hipLaunchKernelGGL(env_step_kernel<true>, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, A);
    } else {
        if (A.o.ev_start && A.o.ev_stop)
            hipExtLaunchKernelGGL(env_step_kernel<false>, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                                  (hipEvent_t)A.o.ev_start, (hipEvent_t)A.o.ev_stop, 0, A);
        else
            hipLaunchKernelGGL(env_step_kernel<false>, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, A);
    }
    HX_CHECK_LAUNCH("hx_env_step");
    return 0;
}

int hx_env_rearm(float* state, int64_t n, int64_t stride, const uint8_t* mask, void* stream) {
    HX_REQUIRE(state && n > 0 && stride >= n, "hx_env_rearm: bad state/n/stride");
    hipLaunchKernelGGL(env_rearm_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, state, n, stride, mask);
    HX_CHECK_LAUNCH("hx_env_rearm");
    return 0;
}

int hx_label_transitions(const float* s, const float* a, const float* ns, int64_t n, float* reward, int8_t* success,
                         uint8_t* done, void* stream) {
    HX_REQUIRE(s && a && ns && reward && success && done && n > 0, "hx_label_transitions: bad arguments");
    hipLaunchKernelGGL(label_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, s, a, ns, n, reward, success, done);
    HX_CHECK_LAUNCH("hx_label_transitions");
    return 0;
}

}  // extern "C"
