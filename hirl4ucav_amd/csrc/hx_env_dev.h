// hx_env_dev.h — device functions of the pursuit-lock-launch env (one env per lane): state load/store cursors, reset, the
// re-derived simulator tick (docs/DYNAMICS.md), the wrapper's latches / observation / reward / termination.
// Shared by hx_env.hip (env_step_kernel & co) and hx_update.hip (the act + env step fused launch).
//
// Numerics: everything in here is compiled with floating-point contraction OFF, whatever the including translation unit uses:
// state-evolving arithmetic is + - * / sqrt in a fixed order, so state words and masks are reproducible bit for bit on any
// IEEE-754 fp32 implementation (the CPU restatement the tests check against).  asinf/atan2f/acosf appear only in the observation.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/hirl4ucav.h"

#pragma clang fp contract(off)
namespace hxenv {

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

// E10/E11: the scripted opponent's commanded levels for this tick  HarfangEnv_GYM.py:145-147 / :342-353 / :412-421
__device__ __forceinline__ void script_opponent(Env& E, float& op, float& orl, float& oy) {
    op = 0.0f; orl = 0.0f; oy = 0.0f;
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
}

// E5: one simulator tick with the commanded levels of BOTH aircraft given (what UPDATE_SCENE does after the SET_PLANE_* / FIRE_MISSILE
// calls of the wire protocol, dogfight_client.py): missile launch, two aircraft ticks, missile flight / hit, targeting device
__device__ __forceinline__ void sim_core(Env& E, float a0, float a1, float a2, float op, float orl, float oy, bool fire) {
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

// E4 + E10/E11 + E5: apply the action, script the opponent, one simulator tick
__device__ __forceinline__ void sim_step(Env& E, float a0, float a1, float a2, bool fire) {
    float op, orl, oy;
    script_opponent(E, op, orl, oy);
    E.flags = fire ? (E.flags | HX_F_FIRED) : (E.flags & ~HX_F_FIRED);  // now_missile_state :150-156
    sim_core(E, a0, a1, a2, op, orl, oy, fire);
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

}  // namespace hxenv
#pragma clang fp contract(fast)
