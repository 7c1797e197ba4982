// hx_env_dev.h — device functions of the pursuit-lock-launch env: state cursors, reset, the re-derived simulator tick
// (docs/DYNAMICS.md, model v2), the wrapper's latches / observation / reward / termination, and the two lane layouts that
// step an env:
//   Solo  one lane per env (throughput shape: ≥ 64k envs per launch)
//   Pair  two adjacent lanes per env — lane 2e ("env lane") owns the ally aircraft, the missile, the targeting device and the
//         wrapper; lane 2e+1 owns the opponent aircraft.  Both run the SAME aircraft tick and Euler extraction on their own
//         plane, so the env's dependent chain is one aircraft long instead of two; the opponent's new position crosses over
//         with one DPP quad_perm per word.  Values are bit-identical to Solo (same functions on the same inputs).
// Shared by hx_env.hip (env_step_kernel & co) and hx_act.hip (the act + env step fused launch).
//
// Numerics: fp32.  Every operation below is spelled out: `fm(a, b, c)` is ONE fused multiply-add (v_fma_f32 = C fmaf, both
// correctly rounded), everything else rounds once per operator in the order written; the file is compiled with contraction
// OFF whatever the including translation unit uses, so the compiler fuses nothing on its own.  Only + - * / sqrt fma touch
// state AND observation (asin / acos / atan2 are the polynomials below, not libm), so a scalar CPU restatement of
// docs/DYNAMICS.md (the one the tests check against) reproduces every state word, mask, observation and reward bit for bit.
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
constexpr float kPi = 3.14159265358979323846f, kHalfPi = 1.57079632679489661923f, kRad2Deg = 57.29577951308232f;
constexpr float kInvPi = 0.31830988618379067154f;   // Euler angles / pi      constants.py NormStates, HarfangEnv_GYM.py:199-201
constexpr float kInv180 = 1.0f / 180.0f;            // target_angle / 180     HarfangEnv_GYM.py:234
constexpr float kInv1e4 = 1.0e-4f;                  // positions / 10000      HarfangEnv_GYM.py:196-198

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ float fm(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return fm(a.z, b.z, fm(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- inverse trigonometry: polynomials on + - * / sqrt fma (max error 1e-7 rad) --------------------------------
// asin core on a = |x|: a <= 0.5: a + a z P(z), z = a^2;  a > 0.5: pi/2 - 2 (s + s z P(z)), z = (1 - a)/2, s = sqrt(z)
__device__ __forceinline__ float asin_poly(float z) {
    float p = 0.038328301161527634f;
    p = fm(p, z, 0.02643335610628128f);
    p = fm(p, z, 0.045020218938589096f);
    p = fm(p, z, 0.07498761266469955f);
    p = fm(p, z, 0.16666673123836517f);
    return p * z;
}
__device__ __forceinline__ float hx_asin(float x) {
    const float a = fabsf(x);
    const bool big = a > 0.5f;
    const float z = big ? fm(-0.5f, a, 0.5f) : a * a;
    const float s = big ? sqrtf(z) : a;
    const float t = fm(s, asin_poly(z), s);
    const float r = big ? fm(-2.0f, t, kHalfPi) : t;
    return x < 0.0f ? -r : r;
}
__device__ __forceinline__ float hx_acos(float x) {
    const float a = fabsf(x);
    const bool big = a > 0.5f;
    const float z = big ? fm(-0.5f, a, 0.5f) : a * a;
    const float s = big ? sqrtf(z) : a;
    const float t = fm(s, asin_poly(z), s);
    const float t2 = t + t;
    const float rb = x < 0.0f ? kPi - t2 : t2;
    const float rs = x < 0.0f ? kHalfPi + t : kHalfPi - t;
    return big ? rb : rs;
}
// atan2(y, x): a = min(|x|,|y|) / max(|x|,|y|) in [0, 1], atan(a) = a P(a^2), octant fix-ups
__device__ __forceinline__ float hx_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const float q = mn / mx;
    const float a = mx > 0.0f ? q : 0.0f;
    const float t = a * a;
    float p = 0.0029034700710326433f;
    p = fm(p, t, -0.01628268137574196f);
    p = fm(p, t, 0.04303884133696556f);
    p = fm(p, t, -0.07533632218837738f);
    p = fm(p, t, 0.10654657334089279f);
    p = fm(p, t, -0.14207129180431366f);
    p = fm(p, t, 0.19993053376674652f);
    p = fm(p, t, -0.3333309292793274f);
    p = fm(p, t, 1.0f);
    float r = a * p;
    r = ay > ax ? kHalfPi - r : r;
    r = x < 0.0f ? kPi - r : r;
    return y < 0.0f ? -r : r;
}

struct Plane {
    V3 p, v;
    float qw, qx, qy, qz;
    float lp, lr, ly;  // pitch, roll, yaw levels
};

struct Axes {
    V3 X, Y, Z;
};

// rotation-matrix columns of a unit quaternion: 3 doublings, 3 products, 11 fma
__device__ __forceinline__ Axes quat_axes(float w, float x, float y, float z) {
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xy2 = x2 * y, xz2 = x2 * z, yz2 = y2 * z;
    const float ax = fm(-x2, x, 1.0f), ay = fm(-y2, y, 1.0f);
    Axes a;
    a.X = {fm(-z2, z, ay), fm(w, z2, xy2), fm(-w, y2, xz2)};
    a.Y = {fm(-w, z2, xy2), fm(-z2, z, ax), fm(w, x2, yz2)};
    a.Z = {fm(w, y2, xz2), fm(-w, x2, yz2), fm(-y2, y, ax)};
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
    float x = fm(-kLapse, h, 1.0f);
    x = x < 0.05f ? 0.05f : x;
    const float x2 = x * x;
    const float sigma = (x2 * x2) * fm(0.25f, x, 0.75f);  // x^4.25 ~ x^4 (3 + x)/4 (first order about sea level)
    const float hr = kRho0Half * sigma;
    const float s2 = dot3(P.v, P.v);
    const float s = sqrtf(s2);
    const float qd = hr * s2;
    const float eff = qd / (qd + kQHalf);
    const float wx = (kPitchRate * P.lp) * eff;
    const float wy = (kYawRate * P.ly) * eff;
    const float wz = fm(kRollRate * P.lr, eff, -(kLeveller * A.X.y));
    const float Om = -((kEasy * A.X.y) * eff);
    {
        const float qw = P.qw, qx = P.qx, qy = P.qy, qz = P.qz;
        const float dw = fm(-Om, qy, -fm(qz, wz, fm(qy, wy, qx * wx)));
        const float dx = fm(Om, qz, fm(-qz, wy, fm(qy, wz, qw * wx)));
        const float dy = fm(Om, qw, fm(-qx, wz, fm(qz, wx, qw * wy)));
        const float dz = fm(-Om, qx, fm(-qy, wx, fm(qx, wy, qw * wz)));
        const float hdt = 0.5f * kDt;
        const float nw = fm(hdt, dw, qw), nx = fm(hdt, dx, qx), ny = fm(hdt, dy, qy), nz = fm(hdt, dz, qz);
        const float inv = 1.0f / sqrtf(fm(nz, nz, fm(ny, ny, fm(nx, nx, nw * nw))));
        P.qw = nw * inv;
        P.qx = nx * inv;
        P.qy = ny * inv;
        P.qz = nz * inv;
    }
    const float vbx = dot3(P.v, A.X), vby = dot3(P.v, A.Y), vbz = dot3(P.v, A.Z);
    const float k = hr * s;
    const float fx = -((kCdX * k) * vbx);
    const float fy = fm(kCl0 * hr, vbz * vbz, -((kCdY * k) * vby));
    const float fz = fm(-(kCdZ * k), vbz, kTMax * thrust);
    const float ax = fm(A.Z.x, fz, fm(A.Y.x, fy, A.X.x * fx));
    const float ay = fm(A.Z.y, fz, fm(A.Y.y, fy, A.X.y * fx)) - kGrav;
    const float az = fm(A.Z.z, fz, fm(A.Y.z, fy, A.X.z * fx));
    P.v.x = fm(ax, kDt, P.v.x);
    P.v.y = fm(ay, kDt, P.v.y);
    P.v.z = fm(az, kDt, P.v.z);
    P.p.x = fm(P.v.x, kDt, P.p.x);
    P.p.y = fm(P.v.y, kDt, P.p.y);
    P.p.z = fm(P.v.z, kDt, P.p.z);
}

// what is left of an env once the two aircraft are taken out (11 state words)
struct Shared {
    V3 mp, mv;
    float health, lock_timer, missile_age;
    uint32_t flags, counters;
};
struct Env {
    Plane ally, opp;
    Shared s;
};

// State words are visited in order through a cursor whose base is UNIFORM (block base + w * stride lives in SGPRs, two scalar
// adds per word) and whose per-lane part is a 32-bit BYTE offset: every access is `global_load/store v, v_off, s[base]`, no
// 64-bit address arithmetic per lane.
struct RCursor {
    const char* __restrict__ p;  // uniform
    int64_t stride_b;            // bytes between consecutive words of one env
    uint32_t off;                // this lane's byte offset
    __device__ __forceinline__ RCursor(const float* base, int64_t stride, uint32_t index)
        : p(reinterpret_cast<const char*>(base)), stride_b(stride * 4), off(index * 4u) {}
    __device__ __forceinline__ float next() {
        const float v = *reinterpret_cast<const float*>(p + off);
        p += stride_b;
        return v;
    }
    __device__ __forceinline__ void skip(int words) { p += stride_b * words; }
};
struct WCursor {
    char* __restrict__ p;  // uniform
    int64_t stride_b;
    uint32_t off;
    __device__ __forceinline__ WCursor(float* base, int64_t stride, uint32_t index)
        : p(reinterpret_cast<char*>(base)), stride_b(stride * 4), off(index * 4u) {}
    __device__ __forceinline__ void put(float v) {
#if defined(HX_ENV_NT) && HX_ENV_NT >= 2
        __builtin_nontemporal_store(v, reinterpret_cast<float*>(p + off));
#else
        *reinterpret_cast<float*>(p + off) = v;
#endif
        p += stride_b;
    }
    __device__ __forceinline__ void skip(int words) { p += stride_b * words; }
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
__device__ __forceinline__ void load_shared(Shared& S, RCursor& c) {
    S.mp.x = c.next(); S.mp.y = c.next(); S.mp.z = c.next();
    S.mv.x = c.next(); S.mv.y = c.next(); S.mv.z = c.next();
    S.health = c.next();
    S.lock_timer = c.next();
    S.missile_age = c.next();
    S.flags = __float_as_uint(c.next());
    S.counters = __float_as_uint(c.next());
}
__device__ __forceinline__ void store_shared(const Shared& S, WCursor& c) {
    c.put(S.mp.x); c.put(S.mp.y); c.put(S.mp.z);
    c.put(S.mv.x); c.put(S.mv.y); c.put(S.mv.z);
    c.put(S.health);
    c.put(S.lock_timer);
    c.put(S.missile_age);
    c.put(__uint_as_float(S.flags));
    c.put(__uint_as_float(S.counters));
}
// i = i0 + lane with i0 uniform across the workgroup
__device__ __forceinline__ void load_env(Env& E, const float* __restrict__ s, int64_t stride, int64_t i0, uint32_t lane) {
    RCursor c(s + i0, stride, lane);
    load_plane(E.ally, c);
    load_plane(E.opp, c);
    load_shared(E.s, c);
}
__device__ __forceinline__ void store_env(const Env& E, float* __restrict__ s, int64_t stride, int64_t i0, uint32_t lane) {
    WCursor c(s + i0, stride, lane);
    store_plane(E.ally, c);
    store_plane(E.opp, c);
    store_shared(E.s, c);
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
// reset / random_reset: HarfangEnv_GYM.py:34-81 (+ :374-406 serpentine, :440-474 circular), one piece per owner
__device__ __forceinline__ void ally_reset(Plane& P, bool randomize, uint64_t seed, uint32_t env_id, uint32_t episode) {
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    if (randomize) {
        uint32_t r[4];
        philox4x32_10(env_id, episode, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        ox = offset201(r[0]);
        oy = offset201(r[1]);
        oz = offset201(r[2]);
    }
    plane_reset(P, 0.0f + ox, 3500.0f + oy, -4000.0f + oz, 300.0f);  // :73-74,77
}
__device__ __forceinline__ void opp_reset(Plane& P, uint32_t scenario) {
    plane_reset(P, 0.0f, 4200.0f, 0.0f, scenario == 2u ? 290.0f : 200.0f);  // :71-72,78 / :472-473
}
__device__ __forceinline__ void shared_reset(Shared& S, uint32_t scenario) {
    S.mp = {0.0f, 0.0f, 0.0f};
    S.mv = {0.0f, 0.0f, 0.0f};
    S.health = 0.2f;  // :70
    S.lock_timer = 0.0f;
    S.missile_age = 0.0f;
    // latches as reset()'s own _get_observation leaves them: locked_prev = locked = False, slots True
    S.flags = HX_F_SLOT_PREV | HX_F_SLOT | HX_F_SIM_SLOT | (scenario << HX_F_SCEN_SHIFT);
    S.counters = 0u;
}
__device__ __forceinline__ void env_reset(Env& E, uint32_t scenario, bool randomize, uint64_t seed, uint32_t env_id,
                                          uint32_t episode) {
    opp_reset(E.opp, scenario);
    ally_reset(E.ally, randomize, seed, env_id, episode);
    shared_reset(E.s, scenario);
}

// ---- read-back pieces (what the wrapper reads from the simulator, HarfangEnv_GYM.py:193-268) --------------------------
// Euler angles of one aircraft, already divided by pi: (pitch, heading, roll)   :199-201,216-218
__device__ __forceinline__ V3 euler_rad(const Axes& A) {
    return {hx_asin(clampf(-A.Z.y, -1.0f, 1.0f)), hx_atan2(A.Z.x, A.Z.z), hx_atan2(A.X.y, A.Y.y)};
}
__device__ __forceinline__ V3 euler_norm(const Axes& A) {
    const V3 e = euler_rad(A);
    return {e.x * kInvPi, e.y * kInvPi, e.z * kInvPi};
}
// ally -> opponent geometry shared by the targeting device and the read-back
struct Geo {
    V3 rel;  // p_ally - p_opp: what the observation carries (HarfangEnv_GYM.py:237); NOT -(p_opp - p_ally): 0 - 0 is +0 either way
    float dist, cosang;
};
__device__ __forceinline__ Geo geometry(const V3& pa, const V3& aZ, const V3& po) {
    Geo g;
    const V3 d = {po.x - pa.x, po.y - pa.y, po.z - pa.z};
    g.rel = {pa.x - po.x, pa.y - po.y, pa.z - po.z};
    g.dist = sqrtf(dot3(d, d));
    g.cosang = dot3(aZ, d) / g.dist;
    return g;
}
__device__ __forceinline__ float target_angle_deg(float cosang) { return hx_acos(clampf(cosang, -1.0f, 1.0f)) * kRad2Deg; }

// E10/E11: the scripted opponent's commanded levels for this tick  HarfangEnv_GYM.py:145-147 / :342-353 / :412-421
__device__ __forceinline__ void script_opponent(uint32_t& flags, uint32_t& counters, float& op, float& orl, float& oy) {
    op = 0.0f; orl = 0.0f; oy = 0.0f;
    uint32_t script = counters >> 16;
    const uint32_t scen = (flags >> HX_F_SCEN_SHIFT) & 3u;
    if (scen == 1u) {
        script += 1u;
        // script % duration == 0 with duration 250 (first leg) or 500: two constant moduli, no integer division
        if (((flags & HX_F_SERP_LONG) ? script % 500u : script % 250u) == 0u) {
            script = 0u;
            flags ^= HX_F_SERP_POS;
            flags |= HX_F_SERP_LONG;
        }
        oy = (flags & HX_F_SERP_POS) ? 0.1f : -0.1f;
    } else if (scen == 2u) {
        if (script < 65535u) script += 1u;
        op = script < 100u ? -0.02f : -0.01f;
        orl = 0.28f;  // the 0.84 sent first never reaches a tick (:415-420)
    }
    counters = (counters & 0xFFFFu) | (script << 16);
}

// FIRE_MISSILE is handled before the tick, with the lock the simulator holds at that moment and the ally's pre-tick state
__device__ __forceinline__ void missile_launch(Shared& S, const Plane& ally, bool fire) {
    if (fire && (S.flags & HX_F_SIM_SLOT)) {
        const Axes A = quat_axes(ally.qw, ally.qx, ally.qy, ally.qz);
        S.flags &= ~HX_F_SIM_SLOT;
        S.flags |= HX_F_M_ACTIVE;
        S.flags = (S.lock_timer >= kLockDelay) ? (S.flags | HX_F_M_GUIDED) : (S.flags & ~HX_F_M_GUIDED);
        S.mp = ally.p;
        S.mv = {fm(A.Z.x, kMBoost, ally.v.x), fm(A.Z.y, kMBoost, ally.v.y), fm(A.Z.z, kMBoost, ally.v.z)};
        S.missile_age = 0.0f;
    }
}
// missile flight / hit against the opponent's post-tick position
__device__ __forceinline__ void missile_tick(Shared& S, const V3& po) {
    if (S.flags & HX_F_M_ACTIVE) {
        const float ms = sqrtf(dot3(S.mv, S.mv));
        const float ims = 1.0f / ms;
        V3 dir = {S.mv.x * ims, S.mv.y * ims, S.mv.z * ims};
        if (S.flags & HX_F_M_GUIDED) {
            const V3 to = {po.x - S.mp.x, po.y - S.mp.y, po.z - S.mp.z};
            const float idist = 1.0f / sqrtf(dot3(to, to));
            const V3 nd = {fm(kMTurn, fm(to.x, idist, -dir.x), dir.x), fm(kMTurn, fm(to.y, idist, -dir.y), dir.y),
                           fm(kMTurn, fm(to.z, idist, -dir.z), dir.z)};
            const float inn = 1.0f / sqrtf(dot3(nd, nd));
            dir = {nd.x * inn, nd.y * inn, nd.z * inn};
        }
        float ms2 = fm(kMAcc, kDt, ms);
        ms2 = ms2 > kMVmax ? kMVmax : ms2;
        S.mv = {dir.x * ms2, dir.y * ms2, dir.z * ms2};
        S.mp = {fm(S.mv.x, kDt, S.mp.x), fm(S.mv.y, kDt, S.mp.y), fm(S.mv.z, kDt, S.mp.z)};
        S.missile_age = S.missile_age + kDt;
        const V3 d = {po.x - S.mp.x, po.y - S.mp.y, po.z - S.mp.z};
        if (dot3(d, d) < kMHit2) {
            const float hl = S.health - kMDamage;
            S.health = hl < 0.0f ? 0.0f : hl;
            S.flags &= ~HX_F_M_ACTIVE;
        } else if (S.missile_age > kMLife) {
            S.flags &= ~HX_F_M_ACTIVE;
        }
    }
}
__device__ __forceinline__ void lock_update(Shared& S, const Geo& g) {
    const bool in_cone = (g.cosang > kCosLock) && (g.dist > kLockMin) && (g.dist < kLockMax);
    S.lock_timer = in_cone ? S.lock_timer + kDt : 0.0f;
}

// E5 for the wire-protocol server: one simulator tick with the commanded levels of BOTH aircraft given (what UPDATE_SCENE does
// after the SET_PLANE_* / FIRE_MISSILE calls, dogfight_client.py)
__device__ __forceinline__ void sim_core(Env& E, float a0, float a1, float a2, float op, float orl, float oy, bool fire) {
    missile_launch(E.s, E.ally, fire);
    const uint32_t scen = (E.s.flags >> HX_F_SCEN_SHIFT) & 3u;
    plane_tick(E.ally, a0, a1, a2, 1.0f);
    plane_tick(E.opp, op, orl, oy, scen == 2u ? 0.8f : 0.6f);
    missile_tick(E.s, E.opp.p);
    const Axes A = quat_axes(E.ally.qw, E.ally.qx, E.ally.qy, E.ally.qz);
    lock_update(E.s, geometry(E.ally.p, A.Z, E.opp.p));
}

// ---- wrapper: E6 latches, E7 reward, E8 termination on the post-tick read-back ---------------------------------------
struct Wrapped {
    float o0, o1, o2, o6, o7, o8, o12;  // the observation entries the env lane owns (the six Euler entries come from euler_norm)
    float reward;
    int success;
    bool done;
};
// g: post-tick geometry (ally -> opponent), altitude = ally y
__device__ __forceinline__ void wrap_step(Shared& S, const Geo& g, float altitude, Wrapped& W) {
    uint32_t f = S.flags;
    // Ally_target_locked <- n_Ally_target_locked ; n_Ally_target_locked <- read-back    :227-228
    f = (f & ~HX_F_LOCKED_PREV) | ((f & HX_F_LOCKED) ? HX_F_LOCKED_PREV : 0u);
    f = (f & ~HX_F_LOCKED) | ((S.lock_timer >= kLockDelay) ? HX_F_LOCKED : 0u);
    // missile1_state <- n_missile1_state ; n_missile1_state <- slots[0]                  :250-251
    f = (f & ~HX_F_SLOT_PREV) | ((f & HX_F_SLOT) ? HX_F_SLOT_PREV : 0u);
    f = (f & ~HX_F_SLOT) | ((f & HX_F_SIM_SLOT) ? HX_F_SLOT : 0u);
    const float ta = target_angle_deg(g.cosang) * kInv180;  // :234
    W.o0 = g.rel.x * kInv1e4;  // (p_ally - p_oppo) / 10000  :237,262
    W.o1 = g.rel.y * kInv1e4;
    W.o2 = g.rel.z * kInv1e4;
    W.o6 = ta;
    W.o7 = (f & HX_F_LOCKED) ? 1.0f : -1.0f;
    W.o8 = (f & HX_F_SLOT) ? 1.0f : -1.0f;
    W.o12 = S.health;
    float r = 0.0f;
    int s = 0;
    r = r - 0.0001f * g.dist;                     // :107  (|p_ally - p_oppo| = |p_oppo - p_ally| bit for bit)
    r = r - ta * 10.0f;                           // :110
    if (altitude < 2000.0f) r = r - 4.0f;         // :112-113
    if (altitude > 7000.0f) r = r - 4.0f;         // :115-116
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
    if (S.health < 0.1f && (f & HX_F_FIRE_SUCCESS)) r = r + 600.0f;
    if (altitude < 500.0f || altitude > 10000.0f) f |= HX_F_DONE;              // :163-164
    if (S.health <= 0.0f) f |= HX_F_DONE | HX_F_EPISODE_SUCCESS;                // :165-167
    S.flags = f;
    W.reward = r;
    W.success = s;
    W.done = (f & HX_F_DONE) != 0u;
}
// the env lane's observation entries of the CURRENT state without touching the latches (right after a reset)
__device__ __forceinline__ void observe_shared(const Shared& S, const Geo& g, Wrapped& W) {
    W.o0 = g.rel.x * kInv1e4;
    W.o1 = g.rel.y * kInv1e4;
    W.o2 = g.rel.z * kInv1e4;
    W.o6 = target_angle_deg(g.cosang) * kInv180;
    W.o7 = (S.flags & HX_F_LOCKED) ? 1.0f : -1.0f;
    W.o8 = (S.flags & HX_F_SLOT) ? 1.0f : -1.0f;
    W.o12 = S.health;
}

// full 13-vector of an env held by one lane (reset kernel, read-back kernel)
__device__ __forceinline__ void observe(const Env& E, float (&obs)[HX_OBS_DIM], float* angle_deg = nullptr) {
    const Axes A = quat_axes(E.ally.qw, E.ally.qx, E.ally.qy, E.ally.qz);
    const Axes B = quat_axes(E.opp.qw, E.opp.qx, E.opp.qy, E.opp.qz);
    const Geo g = geometry(E.ally.p, A.Z, E.opp.p);
    Wrapped W;
    observe_shared(E.s, g, W);
    const V3 ea = euler_norm(A), eo = euler_norm(B);
    obs[0] = W.o0; obs[1] = W.o1; obs[2] = W.o2;
    obs[3] = ea.x; obs[4] = ea.y; obs[5] = ea.z;
    obs[6] = W.o6; obs[7] = W.o7; obs[8] = W.o8;
    obs[9] = eo.x; obs[10] = eo.y; obs[11] = eo.z;
    obs[12] = W.o12;
    if (angle_deg) *angle_deg = target_angle_deg(g.cosang);
}

// =====================================================================================================================
// One env step with the vectorised driver's episode rules (train_all.py:341-361), fused replay insert and statistics, for
// a group of lanes of ONE wave (or several waves, see the barrier hooks).  Used by env_step_kernel (hx_env.hip) and by
// the tail of act_fused_kernel (hx_act.hip).
//
//   PAIR = false: lane l steps env l.        PAIR = true: lanes 2e / 2e+1 step env e (ally+shared / opponent).
//
// LDS tiles (caller provides): s_obs [envs][13] — in: previous observations (INSERT), out: next observations;
//                              s_row [envs][33] — replay rows in ring order (INSERT).
// =====================================================================================================================
constexpr int kRowPitch = HX_ROW_WORDS + 1;  // +1: conflict-free per-lane row writes

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float swap1(float v) { return dpp_f<0xB1>(v); }  // quad_perm [1,0,3,2]: the other lane of the pair
__device__ __forceinline__ uint32_t swap1u(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }

// total % cap without a 64-bit division (the software routine is ~150 instructions on a lane every other lane then waits for):
// quotient estimate in fp64 (exact for total < 2^52), one correction step either way.  cap < 2^31.
__device__ __forceinline__ unsigned ring_slot(unsigned long long total, unsigned long long cap, double inv_cap) {
    const unsigned long long q = (unsigned long long)((double)total * inv_cap);
    long long r = (long long)(total - q * cap);
    if (r < 0) r += (long long)cap;
    if (r >= (long long)cap) r -= (long long)cap;
    return (unsigned)r;
}
__device__ __forceinline__ unsigned wrap_slot(unsigned slot, unsigned cap) { return slot >= cap ? slot - cap : slot; }

// A NaN / Inf action component (a diverged policy) is taken as 0 before it can reach the state; returns whether any was replaced.
__device__ __forceinline__ bool sanitize_action(float4& a) {
    const bool bx = !(fabsf(a.x) <= 3.0e38f), by = !(fabsf(a.y) <= 3.0e38f), bz = !(fabsf(a.z) <= 3.0e38f), bw = !(fabsf(a.w) <= 3.0e38f);
    a.x = bx ? 0.0f : a.x;
    a.y = by ? 0.0f : a.y;
    a.z = bz ? 0.0f : a.z;
    a.w = bw ? 0.0f : a.w;
    return bx || by || bz || bw;
}

// The arithmetic of one env step for one lane.  Solo: the lane holds the whole env.  Pair: the lane holds ITS aircraft (`mine`)
// and a copy of the 11 shared words; both lanes of a pair run the same instruction stream — the opponent lane's missile /
// targeting / wrapper results are computed on meaningless operands and simply never stored (the lanes share a wave: masking
// them off would cost the same cycles plus the exec-mask bookkeeping).
template <bool PAIR>
struct Stepper {
    Plane mine;   // Pair: this lane's aircraft; Solo: the ally
    Plane other;  // Solo only: the opponent
    Shared S;

    // e_local: env inside the tile (i0 = first env of the tile); is_opp: this lane owns the opponent aircraft (Pair)
    __device__ __forceinline__ void load(const float* __restrict__ state, int64_t stride, int64_t i0, uint32_t e_local, bool is_opp) {
        // Pair: the opponent lane reads 13 words further on — a per-lane 32-bit offset, the base stays uniform (26 * stride * 4 < 2^32)
        RCursor c(state + i0, stride, PAIR ? e_local + (is_opp ? 13u * (uint32_t)stride : 0u) : e_local);
        load_plane(mine, c);
        if (!PAIR) load_plane(other, c);
        RCursor cs(state + i0 + stride * 26, stride, e_local);
        load_shared(S, cs);
    }
    __device__ __forceinline__ void store(float* __restrict__ state, int64_t stride, int64_t i0, uint32_t e_local, bool is_opp) const {
        WCursor c(state + i0, stride, PAIR ? e_local + (is_opp ? 13u * (uint32_t)stride : 0u) : e_local);
        store_plane(mine, c);
        if (!PAIR) store_plane(other, c);
        if (!is_opp) {
            WCursor cs(state + i0 + stride * 26, stride, e_local);
            store_shared(S, cs);
        }
    }
    __device__ __forceinline__ uint32_t episode_step() const { return S.counters & 0xFFFFu; }

    // E4 + E10/E11 + E5 + E6..E8.  eu: the Euler entries of `mine`; eu_other: of `other` (Solo); W: meaningful on the env lane.
    __device__ __forceinline__ void step(const float4& act, bool is_opp, V3& eu, V3& eu_other, Wrapped& W) {
        float op, orl, oy;
        script_opponent(S.flags, S.counters, op, orl, oy);
        const bool fire = act.w > 0.0f;  // float(action[3] > 0)  HarfangEnv_GYM.py:150
        const uint32_t scen = (S.flags >> HX_F_SCEN_SHIFT) & 3u;
        const float thr_opp = scen == 2u ? 0.8f : 0.6f;
        S.flags = fire ? (S.flags | HX_F_FIRED) : (S.flags & ~HX_F_FIRED);  // now_missile_state :150-156
        missile_launch(S, mine, fire);
        V3 po;
        if (!PAIR) {
            plane_tick(mine, act.x, act.y, act.z, 1.0f);
            plane_tick(other, op, orl, oy, thr_opp);
            po = other.p;
        } else {
            plane_tick(mine, is_opp ? op : act.x, is_opp ? orl : act.y, is_opp ? oy : act.z, is_opp ? thr_opp : 1.0f);
            po = {swap1(mine.p.x), swap1(mine.p.y), swap1(mine.p.z)};  // env lane: the opponent's new position
        }
        missile_tick(S, po);
        const Axes A = quat_axes(mine.qw, mine.qx, mine.qy, mine.qz);
        const Geo g = geometry(mine.p, A.Z, po);
        lock_update(S, g);
        wrap_step(S, g, mine.p.y, W);
        eu = euler_norm(A);
        if (!PAIR) eu_other = euler_norm(quat_axes(other.qw, other.qx, other.qy, other.qz));
        uint32_t ep = S.counters & 0xFFFFu;
        ep = ep < 65535u ? ep + 1u : ep;
        S.counters = (S.counters & 0xFFFF0000u) | ep;
    }

    // auto-reset in place: new state + the reset observation (vectorised counterpart of train_all.py:320-323)
    __device__ __forceinline__ void reset(bool is_opp, bool randomize, uint64_t seed, uint32_t env_id, uint32_t episode, V3& eu, V3& eu_other,
                                          Wrapped& W) {
        const uint32_t scen = (S.flags >> HX_F_SCEN_SHIFT) & 3u;
        Plane fresh_opp;
        opp_reset(fresh_opp, scen);
        ally_reset(mine, randomize, seed, env_id, episode);
        shared_reset(S, scen);
        if (PAIR) {
            const V3 pa = mine.p;  // geometry wants ally -> opponent on the env lane; the opponent lane's copy is never stored
            if (is_opp) mine = fresh_opp;
            const Axes A = quat_axes(mine.qw, mine.qx, mine.qy, mine.qz);
            observe_shared(S, geometry(pa, A.Z, fresh_opp.p), W);
            eu = euler_norm(A);
        } else {
            other = fresh_opp;
            const Axes A = quat_axes(mine.qw, mine.qx, mine.qy, mine.qz);
            observe_shared(S, geometry(mine.p, A.Z, other.p), W);
            eu = euler_norm(A);
            eu_other = euler_norm(quat_axes(other.qw, other.qx, other.qy, other.qz));
        }
    }
};

}  // namespace hxenv
#pragma clang fp contract(fast)
