/*
 * oracle/env_oracle.c — CPU ORACLE for the pursuit-lock-launch env step.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
 * product (hirl4ucav_amd/) never links, imports or calls it.
 *
 * Two layers, deliberately kept apart:
 *
 *  (W) WRAPPER layer — a scalar restatement of what the reference's Python env wrapper computes from
 *      the simulator read-back.  PINNED: checked against golden vectors produced by running the
 *      reference wrapper itself (hirl/environments/HarfangEnv_GYM.py) over a fake dogfight_client
 *      (tests/golden/gen_env_golden.py -> tests/golden/env_wrapper_*.npz).
 *        ox_wrap_reset      <- HarfangEnv.reset / random_reset latch part  HarfangEnv_GYM.py:34-66
 *        ox_wrap_observe    <- HarfangEnv._get_observation                 HarfangEnv_GYM.py:193-268
 *        ox_wrap_reward     <- HarfangEnv._get_reward                      HarfangEnv_GYM.py:101-137
 *        ox_wrap_terminate  <- HarfangEnv._get_termination                 HarfangEnv_GYM.py:160-169
 *        ox_script_opponent <- set_ennemy_yaw (serpentine / circular)      HarfangEnv_GYM.py:342-353,412-421
 *        ox_get_reward / ox_get_termination <- expert labelling            HarfangEnv_GYM.py:299-336
 *
 *  (S) SIMULATOR layer — the 6-DoF tick.  The reference has NO dynamics source (it talks to the external
 *      Harfang dogfight-sandbox-hg2 process over TCP, dogfight_client.py:49,215).  The model below is this
 *      project's own, specified in docs/DYNAMICS.md.  **Dynamics parity with Harfang: UNPINNED.**
 *      What is checked for the simulator is GPU == this scalar restatement on identical (state, action).
 *
 * Numerics (model v2, docs/DYNAMICS.md): fp32 throughout, compiled with -ffp-contract=off so the compiler
 * fuses nothing on its own; where the model says "fma" the code calls fmaf() — ONE correctly rounded
 * operation on the host exactly as v_fma_f32 is on gfx950.  Everything, the inverse trigonometry of the
 * read-back included (ox_asin / ox_acos / ox_atan2 below: polynomials, not libm), is built from
 * + - * / sqrt fma, all correctly rounded on both sides, so state words, masks, observations and rewards
 * can be compared bit for bit.
 *
 * Layout here is array-of-structs on purpose (the product is struct-of-arrays): the oracle is an
 * independent restatement, not a copy of the kernel.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* state                                                                                            */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    float pos[3];
    float vel[3];
    float q[4];   /* w x y z, body->world */
    float lvl[3]; /* smoothed control levels: pitch, roll, yaw */
} OxPlane;        /* 13 words */

typedef struct {
    OxPlane ally;       /* words 0..12  */
    OxPlane opp;        /* words 13..25 */
    float mpos[3];      /* 26..28 */
    float mvel[3];      /* 29..31 */
    float health;       /* 32 opponent health_level */
    float lock_timer;   /* 33 seconds the target has been inside the lock cone */
    float missile_age;  /* 34 */
    uint32_t flags;     /* 35 */
    uint32_t counters;  /* 36: lo16 episode step, hi16 opponent script step */
} OxEnv;                /* 37 words = 148 B */

enum {
    F_LOCKED_PREV = 1u << 0, /* Ally_target_locked   (before the action) HarfangEnv_GYM.py:227 */
    F_LOCKED = 1u << 1,      /* n_Ally_target_locked (after the action)  HarfangEnv_GYM.py:228 */
    F_SLOT_PREV = 1u << 2,   /* missile1_state       HarfangEnv_GYM.py:250 */
    F_SLOT = 1u << 3,        /* n_missile1_state     HarfangEnv_GYM.py:251 */
    F_FIRED = 1u << 4,       /* now_missile_state    HarfangEnv_GYM.py:150-156 */
    F_FIRE_SUCCESS = 1u << 5,
    F_EPISODE_SUCCESS = 1u << 6,
    F_DONE = 1u << 7,
    F_SCEN_SHIFT = 8, /* bits 8..9: 0 straight_line, 1 serpentine, 2 circular */
    F_SERP_POS = 1u << 10,  /* oppo_yaw > 0 */
    F_SERP_LONG = 1u << 11, /* duration == 500 (else 250) */
    F_M_ACTIVE = 1u << 12,  /* missile in flight */
    F_M_GUIDED = 1u << 13,  /* missile launched with a lock */
    F_SIM_SLOT = 1u << 14,  /* simulator side: missile still on the rail */
};

/* simulator read-back, the subset of get_plane_state / get_health / get_missiles_device_slots_state the
 * wrapper consumes (HarfangEnv_GYM.py:195-251) */
typedef struct {
    float ally_pos[3], ally_euler[3];
    float opp_pos[3], opp_euler[3];
    float target_angle_deg;
    float health;
    int32_t target_locked;
    int32_t slot0;
} OxReadback;

/* ------------------------------------------------------------------------------------------------ */
/* constants of the re-derived model (docs/DYNAMICS.md)                                             */
/* ------------------------------------------------------------------------------------------------ */
#define DT (1.0f / 60.0f) /* one UPDATE_SCENE = 1/60 s: hirl/data/straight_line/ai_env.py:18 */
#define SLEW 0.05f        /* level change per tick (3 /s) */
#define RHO0_HALF 0.6125f /* 0.5 * 1.225 kg/m^3 */
#define LAPSE 2.2558e-5f
#define Q_HALF 4000.0f
#define K_PITCH 0.8f
#define K_YAW 0.4f
#define K_ROLL 2.5f
#define K_LEVEL 0.8f
#define K_EASY 0.2f
#define CD_X 0.05f
#define CD_Y 0.02f
#define CD_Z 5.2e-4f
#define CL_0 3.8e-4f
#define T_MAX 20.0f
#define GRAV 9.8f
#define COS_LOCK 0.9659258f /* cos 15 deg */
#define LOCK_MIN 100.0f
#define LOCK_MAX 3000.0f
#define LOCK_DELAY 1.0f
#define M_BOOST 50.0f
#define M_TURN 0.15f
#define M_ACC 300.0f
#define M_VMAX 1000.0f
#define M_LIFE 20.0f
#define M_HIT2 1600.0f /* (40 m)^2 */
#define M_DAMAGE 0.3f
#define PI_F 3.14159265358979323846f
#define RAD2DEG 57.29577951308232f

#define HALF_PI_F 1.57079632679489661923f
#define INV_PI_F 0.31830988618379067154f /* Euler angles / pi     constants.py NormStates, HarfangEnv_GYM.py:199-201 */
#define INV_180_F (1.0f / 180.0f)        /* target_angle / 180    HarfangEnv_GYM.py:234 */
#define INV_1E4_F 1.0e-4f                /* positions / 10000     HarfangEnv_GYM.py:196-198 */

static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* a.b accumulated from the x component: fma(az, bz, fma(ay, by, ax*bx)) */
static inline float dot3(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }

/* ---- inverse trigonometry of the read-back (docs/DYNAMICS.md "Angles"): max error 1e-7 rad ---------------- */
static const float ASIN_C[5] = {0.16666673123836517f, 0.07498761266469955f, 0.045020218938589096f, 0.02643335610628128f,
                                0.038328301161527634f};
static const float ATAN_C[9] = {1.0f, -0.3333309292793274f, 0.19993053376674652f, -0.14207129180431366f, 0.10654657334089279f,
                                -0.07533632218837738f, 0.04303884133696556f, -0.01628268137574196f, 0.0029034700710326433f};

/* z P(z), Horner from the highest coefficient */
static float asin_tail(float z) {
    float p = ASIN_C[4];
    for (int k = 3; k >= 0; --k) p = fmaf(p, z, ASIN_C[k]);
    return p * z;
}
/* shared core: |x| <= 1/2: z = x^2, s = |x|;  else z = (1 - |x|)/2, s = sqrt(z);  t = s + s z P(z) */
static float asin_core(float a, int* big) {
    *big = a > 0.5f;
    float z = *big ? fmaf(-0.5f, a, 0.5f) : a * a;
    float s = *big ? sqrtf(z) : a;
    return fmaf(s, asin_tail(z), s);
}
float ox_asin(float x) {
    int big;
    float t = asin_core(fabsf(x), &big);
    float r = big ? fmaf(-2.0f, t, HALF_PI_F) : t;
    return x < 0.0f ? -r : r;
}
float ox_acos(float x) {
    int big;
    float t = asin_core(fabsf(x), &big);
    if (big) {
        float t2 = t + t;
        return x < 0.0f ? PI_F - t2 : t2;
    }
    return x < 0.0f ? HALF_PI_F + t : HALF_PI_F - t;
}
float ox_atan2(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    float a = mx > 0.0f ? mn / mx : 0.0f;
    float t = a * a;
    float p = ATAN_C[8];
    for (int k = 7; k >= 0; --k) p = fmaf(p, t, ATAN_C[k]);
    float r = a * p;
    if (ay > ax) r = HALF_PI_F - r;
    if (x < 0.0f) r = PI_F - r;
    return y < 0.0f ? -r : r;
}

/* rotation matrix columns (images of body X, Y, Z) of a unit quaternion:
 *   X = (1 - 2yy - 2zz, 2xy + 2wz, 2xz - 2wy)   Y = (2xy - 2wz, 1 - 2xx - 2zz, 2yz + 2wx)   Z = (2xz + 2wy, 2yz - 2wx, 1 - 2xx - 2yy)
 * with the doublings done first (exact) and one fma per remaining product-sum */
static void quat_axes(const float* q, float* aX, float* aY, float* aZ) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xy2 = x2 * y, xz2 = x2 * z, yz2 = y2 * z;
    const float ax = fmaf(-x2, x, 1.0f), ay = fmaf(-y2, y, 1.0f);
    aX[0] = fmaf(-z2, z, ay); aX[1] = fmaf(w, z2, xy2);  aX[2] = fmaf(-w, y2, xz2);
    aY[0] = fmaf(-w, z2, xy2); aY[1] = fmaf(-z2, z, ax); aY[2] = fmaf(w, x2, yz2);
    aZ[0] = fmaf(w, y2, xz2);  aZ[1] = fmaf(-w, x2, yz2); aZ[2] = fmaf(-y2, y, ax);
}

/* Euler angles (pitch about X, heading about Y, roll about Z) of R = Ry(h) Rx(p) Rz(r); pitch > 0 is nose
 * down, as the comments at HarfangEnv_GYM.py:199-204 describe for the simulator's read-back. */
static void quat_euler(const float* q, float* e) {
    float aX[3], aY[3], aZ[3];
    quat_axes(q, aX, aY, aZ);
    e[0] = ox_asin(clampf(-aZ[1], -1.0f, 1.0f));
    e[1] = ox_atan2(aZ[0], aZ[2]);
    e[2] = ox_atan2(aX[1], aY[1]);
}

/* ------------------------------------------------------------------------------------------------ */
/* (S) simulator layer                                                                              */
/* ------------------------------------------------------------------------------------------------ */
static void plane_tick(OxPlane* P, const float* cmd, float thrust) {
    /* a. control levels slew toward the command */
    for (int k = 0; k < 3; ++k) {
        float d = clampf(cmd[k] - P->lvl[k], -SLEW, SLEW);
        P->lvl[k] = P->lvl[k] + d;
    }
    /* b. body axes */
    float aX[3], aY[3], aZ[3];
    quat_axes(P->q, aX, aY, aZ);
    /* c. atmosphere, dynamic pressure, control effectiveness */
    float h = clampf(P->pos[1], 0.0f, 30000.0f);
    float x = fmaf(-LAPSE, h, 1.0f);
    x = x < 0.05f ? 0.05f : x;
    float x2 = x * x;
    float sigma = (x2 * x2) * fmaf(0.25f, x, 0.75f);
    float hr = RHO0_HALF * sigma;
    float s2 = dot3(P->vel, P->vel);
    float s = sqrtf(s2);
    float qd = hr * s2;
    float eff = qd / (qd + Q_HALF);
    /* d. angular velocity: body rates from the levels, wing-leveller, "easy steering" world yaw */
    float wx = (K_PITCH * P->lvl[0]) * eff;
    float wy = (K_YAW * P->lvl[2]) * eff;
    float wz = fmaf(K_ROLL * P->lvl[1], eff, -(K_LEVEL * aX[1]));
    float Om = -((K_EASY * aX[1]) * eff);
    /* e. attitude: q += dt/2 * ( q (x) (0,w_body) + (0,0,Om,0) (x) q ), renormalise with ONE reciprocal */
    {
        const float qw = P->q[0], qx = P->q[1], qy = P->q[2], qz = P->q[3];
        float dq[4];
        dq[0] = fmaf(-Om, qy, -fmaf(qz, wz, fmaf(qy, wy, qx * wx)));
        dq[1] = fmaf(Om, qz, fmaf(-qz, wy, fmaf(qy, wz, qw * wx)));
        dq[2] = fmaf(Om, qw, fmaf(-qx, wz, fmaf(qz, wx, qw * wy)));
        dq[3] = fmaf(-Om, qx, fmaf(-qy, wx, fmaf(qx, wy, qw * wz)));
        const float hdt = 0.5f * DT;
        float nq[4];
        for (int c = 0; c < 4; ++c) nq[c] = fmaf(hdt, dq[c], P->q[c]);
        float n2 = nq[0] * nq[0];
        for (int c = 1; c < 4; ++c) n2 = fmaf(nq[c], nq[c], n2);
        float inv = 1.0f / sqrtf(n2);
        for (int c = 0; c < 4; ++c) P->q[c] = nq[c] * inv;
    }
    /* f. specific forces in body axes (pre-rotation axes) */
    float vbx = dot3(P->vel, aX), vby = dot3(P->vel, aY), vbz = dot3(P->vel, aZ);
    float k = hr * s;
    float fx = -((CD_X * k) * vbx);
    float fy = fmaf(CL_0 * hr, vbz * vbz, -((CD_Y * k) * vby));
    float fz = fmaf(-(CD_Z * k), vbz, T_MAX * thrust);
    float acc[3];
    for (int c = 0; c < 3; ++c) acc[c] = fmaf(aZ[c], fz, fmaf(aY[c], fy, aX[c] * fx));
    acc[1] = acc[1] - GRAV;
    /* g. semi-implicit Euler */
    for (int c = 0; c < 3; ++c) {
        P->vel[c] = fmaf(acc[c], DT, P->vel[c]);
        P->pos[c] = fmaf(P->vel[c], DT, P->pos[c]);
    }
}

static float scenario_thrust_opp(uint32_t flags) { return ((flags >> F_SCEN_SHIFT) & 3u) == 2u ? 0.8f : 0.6f; }

/* ally -> opponent geometry: distance and cos(target angle); shared by the targeting device and the read-back */
static void target_geometry(const OxEnv* E, float* dist, float* cosang) {
    float aX[3], aY[3], aZ[3], d[3];
    quat_axes(E->ally.q, aX, aY, aZ);
    for (int c = 0; c < 3; ++c) d[c] = E->opp.pos[c] - E->ally.pos[c];
    *dist = sqrtf(dot3(d, d));
    *cosang = dot3(aZ, d) / *dist;
}

/* One UPDATE_SCENE.  ally_cmd / opp_cmd = (pitch, roll, yaw) levels, fire = FIRE_MISSILE was sent before the tick. */
void ox_sim_tick(OxEnv* E, const float* ally_cmd, const float* opp_cmd, int fire) {
    /* missile leaves the rail before the tick, with the lock state the simulator holds at that moment */
    if (fire && (E->flags & F_SIM_SLOT)) {
        float aX[3], aY[3], aZ[3];
        quat_axes(E->ally.q, aX, aY, aZ);
        E->flags &= ~F_SIM_SLOT;
        E->flags |= F_M_ACTIVE;
        if (E->lock_timer >= LOCK_DELAY) E->flags |= F_M_GUIDED; else E->flags &= ~F_M_GUIDED;
        for (int c = 0; c < 3; ++c) {
            E->mpos[c] = E->ally.pos[c];
            E->mvel[c] = fmaf(aZ[c], M_BOOST, E->ally.vel[c]);
        }
        E->missile_age = 0.0f;
    }
    plane_tick(&E->ally, ally_cmd, 1.0f);
    plane_tick(&E->opp, opp_cmd, scenario_thrust_opp(E->flags));
    /* missile */
    if (E->flags & F_M_ACTIVE) {
        float ms = sqrtf(dot3(E->mvel, E->mvel));
        float ims = 1.0f / ms;
        float dir[3];
        for (int c = 0; c < 3; ++c) dir[c] = E->mvel[c] * ims;
        if (E->flags & F_M_GUIDED) {
            float to[3];
            for (int c = 0; c < 3; ++c) to[c] = E->opp.pos[c] - E->mpos[c];
            float idist = 1.0f / sqrtf(dot3(to, to));
            float nd[3];
            for (int c = 0; c < 3; ++c) nd[c] = fmaf(M_TURN, fmaf(to[c], idist, -dir[c]), dir[c]);
            float inn = 1.0f / sqrtf(dot3(nd, nd));
            for (int c = 0; c < 3; ++c) dir[c] = nd[c] * inn;
        }
        float ms2 = fmaf(M_ACC, DT, ms);
        ms2 = ms2 > M_VMAX ? M_VMAX : ms2;
        for (int c = 0; c < 3; ++c) {
            E->mvel[c] = dir[c] * ms2;
            E->mpos[c] = fmaf(E->mvel[c], DT, E->mpos[c]);
        }
        E->missile_age = E->missile_age + DT;
        float d[3];
        for (int c = 0; c < 3; ++c) d[c] = E->opp.pos[c] - E->mpos[c];
        if (dot3(d, d) < M_HIT2) {
            float hl = E->health - M_DAMAGE;
            E->health = hl < 0.0f ? 0.0f : hl;
            E->flags &= ~F_M_ACTIVE;
        } else if (E->missile_age > M_LIFE) {
            E->flags &= ~F_M_ACTIVE;
        }
    }
    /* targeting device of the ally */
    {
        float dist, cosang;
        target_geometry(E, &dist, &cosang);
        int in_cone = (cosang > COS_LOCK) && (dist > LOCK_MIN) && (dist < LOCK_MAX);
        E->lock_timer = in_cone ? E->lock_timer + DT : 0.0f;
    }
}

void ox_sim_readback(const OxEnv* E, OxReadback* rb) {
    for (int c = 0; c < 3; ++c) { rb->ally_pos[c] = E->ally.pos[c]; rb->opp_pos[c] = E->opp.pos[c]; }
    quat_euler(E->ally.q, rb->ally_euler);
    quat_euler(E->opp.q, rb->opp_euler);
    float dist, cosang;
    target_geometry(E, &dist, &cosang);
    rb->target_angle_deg = ox_acos(clampf(cosang, -1.0f, 1.0f)) * RAD2DEG;
    rb->health = E->health;
    rb->target_locked = E->lock_timer >= LOCK_DELAY;
    rb->slot0 = (E->flags & F_SIM_SLOT) != 0;
}

/* ---- Philox4x32-10 (Salmon et al., SC'11), counter-based so host and device agree ---------------- */
void ox_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* integer offset in [-100, 100], the distribution of random.randint(-100, 100) HarfangEnv_GYM.py:74 */
static int32_t offset201(uint32_t u) { return (int32_t)(((uint64_t)u * 201u) >> 32) - 100; }

static void plane_reset(OxPlane* P, float x, float y, float z, float speed) {
    P->pos[0] = x; P->pos[1] = y; P->pos[2] = z;
    P->vel[0] = 0.0f; P->vel[1] = 0.0f; P->vel[2] = speed; /* rot (0,0,0): nose along +Z */
    P->q[0] = 1.0f; P->q[1] = 0.0f; P->q[2] = 0.0f; P->q[3] = 0.0f;
    P->lvl[0] = P->lvl[1] = P->lvl[2] = 0.0f;
}

/* reset / random_reset: HarfangEnv_GYM.py:34-81,171-188 (+ serpentine :374-406, circular :440-474).
 * scenario 0 straight_line, 1 serpentine, 2 circular.  randomize: ally position + U{-100..100} per axis,
 * drawn from Philox(key = seed, counter = (env_id, episode, 0, 0)). */
void ox_env_reset(OxEnv* E, int scenario, int randomize, uint64_t seed, uint32_t env_id, uint32_t episode) {
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    if (randomize) {
        uint32_t ctr[4] = {env_id, episode, 0u, 0u}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
        ox_philox4x32_10(ctr, key, r);
        ox = (float)offset201(r[0]); oy = (float)offset201(r[1]); oz = (float)offset201(r[2]);
    }
    plane_reset(&E->opp, 0.0f, 4200.0f, 0.0f, scenario == 2 ? 290.0f : 200.0f);
    plane_reset(&E->ally, 0.0f + ox, 3500.0f + oy, -4000.0f + oz, 300.0f);
    for (int c = 0; c < 3; ++c) { E->mpos[c] = 0.0f; E->mvel[c] = 0.0f; }
    E->health = 0.2f;
    E->lock_timer = 0.0f;
    E->missile_age = 0.0f;
    /* latches after reset()'s own _get_observation: locked_prev = locked = False, slot_prev = slot = True */
    E->flags = F_SLOT_PREV | F_SLOT | F_SIM_SLOT | ((uint32_t)scenario << F_SCEN_SHIFT);
    E->counters = 0u;
}

/* ------------------------------------------------------------------------------------------------ */
/* (W) wrapper layer                                                                                */
/* ------------------------------------------------------------------------------------------------ */

/* _get_observation: latches + the 13-vector.  HarfangEnv_GYM.py:193-268 */
void ox_wrap_observe(uint32_t* flags, const OxReadback* rb, float* obs) {
    uint32_t f = *flags;
    /* Ally_target_locked <- n_Ally_target_locked ; n_Ally_target_locked <- read-back   :227-228 */
    f = (f & ~F_LOCKED_PREV) | ((f & F_LOCKED) ? F_LOCKED_PREV : 0u);
    f = (f & ~F_LOCKED) | (rb->target_locked ? F_LOCKED : 0u);
    /* missile1_state <- n_missile1_state ; n_missile1_state <- slots[0]                 :250-251 */
    f = (f & ~F_SLOT_PREV) | ((f & F_SLOT) ? F_SLOT_PREV : 0u);
    f = (f & ~F_SLOT) | (rb->slot0 ? F_SLOT : 0u);
    *flags = f;
    for (int c = 0; c < 3; ++c) obs[c] = (rb->ally_pos[c] - rb->opp_pos[c]) * INV_1E4_F; /* / 10000  :196-198,213-215,237 */
    for (int c = 0; c < 3; ++c) obs[3 + c] = rb->ally_euler[c] * INV_PI_F;              /* / pi  :199-201 */
    obs[6] = rb->target_angle_deg * INV_180_F;                                           /* / 180  :234 */
    obs[7] = (f & F_LOCKED) ? 1.0f : -1.0f;                                              /* :229-232 */
    obs[8] = (f & F_SLOT) ? 1.0f : -1.0f;                                                /* :252-255 */
    for (int c = 0; c < 3; ++c) obs[9 + c] = rb->opp_euler[c] * INV_PI_F;               /* :216-218 */
    obs[12] = rb->health;                                                                /* :241 */
}

/* _get_reward: HarfangEnv_GYM.py:101-137.  Uses the PRE-action latches (locked_prev, slot_prev) and the
 * post-action distance, angle and altitude. */
float ox_wrap_reward(uint32_t* flags, const OxReadback* rb, int8_t* success) {
    uint32_t f = *flags;
    float d[3];
    for (int c = 0; c < 3; ++c) d[c] = rb->ally_pos[c] - rb->opp_pos[c];
    float loc_diff = sqrtf(dot3(d, d)); /* :190-191 */
    float r = 0.0f;
    int8_t s = 0;
    r = r - 0.0001f * loc_diff;                       /* :107 */
    r = r - (rb->target_angle_deg * INV_180_F) * 10.0f; /* :110 */
    if (rb->ally_pos[1] < 2000.0f) r = r - 4.0f;      /* :112-113 */
    if (rb->ally_pos[1] > 7000.0f) r = r - 4.0f;      /* :115-116 */
    if (f & F_FIRED) {                                /* :119 */
        r = r - 8.0f;
        if ((f & F_SLOT_PREV) && !(f & F_LOCKED_PREV)) s = -1;           /* :121-124 */
        else if ((f & F_SLOT_PREV) && (f & F_LOCKED_PREV)) { s = 1; f |= F_FIRE_SUCCESS; } /* :125-129 */
    }
    /* the reference compares the float64 image of the read-back with the double 0.1 */
    if ((double)rb->health <= 0.1 && (f & F_FIRE_SUCCESS)) r = r + 600.0f; /* :135-136 */
    *flags = f;
    *success = s;
    return r;
}

/* _get_termination: HarfangEnv_GYM.py:160-169 */
void ox_wrap_terminate(uint32_t* flags, const OxReadback* rb) {
    uint32_t f = *flags;
    if (rb->ally_pos[1] < 500.0f || rb->ally_pos[1] > 10000.0f) f |= F_DONE;
    if (rb->health <= 0.0f) f |= F_DONE | F_EPISODE_SUCCESS;
    *flags = f;
}

/* set_ennemy_yaw: serpentine HarfangEnv_GYM.py:342-353 (+ reset_ennemy :394-399), circular :412-421.
 * Straight line sends zeros (:145-147). */
void ox_script_opponent(uint32_t* flags, uint32_t* counters, float* cmd) {
    uint32_t f = *flags, script = *counters >> 16;
    uint32_t scen = (f >> F_SCEN_SHIFT) & 3u;
    cmd[0] = 0.0f; cmd[1] = 0.0f; cmd[2] = 0.0f;
    if (scen == 1u) {
        script += 1u;
        uint32_t duration = (f & F_SERP_LONG) ? 500u : 250u;
        if (script % duration == 0u) {
            script = 0u;
            f ^= F_SERP_POS;  /* 0.1 * (-1 if oppo_yaw > 0 else 1) */
            f |= F_SERP_LONG; /* duration = 500 */
        }
        cmd[2] = (f & F_SERP_POS) ? 0.1f : -0.1f;
    } else if (scen == 2u) {
        if (script < 65535u) script += 1u;
        cmd[0] = script < 100u ? -0.02f : -0.01f;
        cmd[1] = 0.28f; /* the 0.84 sent first is overwritten before the tick (:415-420) */
    }
    *flags = f;
    *counters = (*counters & 0xFFFFu) | (script << 16);
}

/* observation of the current state WITHOUT touching the latches (used right after reset, where the latch
 * values are set by ox_env_reset to what reset()'s own _get_observation leaves behind) */
void ox_env_observe(const OxEnv* E, float* obs) {
    OxReadback rb;
    ox_sim_readback(E, &rb);
    uint32_t f = E->flags;
    for (int c = 0; c < 3; ++c) obs[c] = (rb.ally_pos[c] - rb.opp_pos[c]) * INV_1E4_F;
    for (int c = 0; c < 3; ++c) obs[3 + c] = rb.ally_euler[c] * INV_PI_F;
    obs[6] = rb.target_angle_deg * INV_180_F;
    obs[7] = (f & F_LOCKED) ? 1.0f : -1.0f;
    obs[8] = (f & F_SLOT) ? 1.0f : -1.0f;
    for (int c = 0; c < 3; ++c) obs[9 + c] = rb.opp_euler[c] * INV_PI_F;
    obs[12] = rb.health;
}

/* A NaN / Inf action component (a diverged policy) is taken as 0 before it can reach the state (the build's stand-in for failure
 * detection).  Returns whether any component was replaced. */
int ox_sanitize_action(const float* in, float* out) {
    int bad = 0;
    for (int c = 0; c < 4; ++c) {
        int b = !(fabsf(in[c]) <= 3.0e38f);
        out[c] = b ? 0.0f : in[c];
        bad |= b;
    }
    return bad;
}

/* HarfangEnv.step: E4 -> E5 -> E6 -> E7 -> E8   HarfangEnv_GYM.py:83-90 (action: already sanitized) */
void ox_env_step(OxEnv* E, const float* action, float* obs, float* reward, uint8_t* done, int8_t* success) {
    float ally_cmd[3] = {action[0], action[1], action[2]}; /* pitch, roll, yaw :140-142 */
    float opp_cmd[3];
    ox_script_opponent(&E->flags, &E->counters, opp_cmd);
    int fire = action[3] > 0.0f; /* :150 */
    if (fire) E->flags |= F_FIRED; else E->flags &= ~F_FIRED;
    ox_sim_tick(E, ally_cmd, opp_cmd, fire);
    OxReadback rb;
    ox_sim_readback(E, &rb);
    ox_wrap_observe(&E->flags, &rb, obs);
    *reward = ox_wrap_reward(&E->flags, &rb, success);
    ox_wrap_terminate(&E->flags, &rb);
    *done = (E->flags & F_DONE) ? 1 : 0;
    uint32_t ep = E->counters & 0xFFFFu;
    if (ep < 65535u) ep += 1u;
    E->counters = (E->counters & 0xFFFF0000u) | ep;
}

/* rearm_machine: HarfangSerpentineInfiniteEnv.step_test HarfangEnv_GYM.py:484-486 */
void ox_env_rearm(OxEnv* E) { E->flags |= F_SIM_SLOT; }

/* Batched step with the vectorised driver's episode handling (train_all.py:341-361 restated per env):
 *   - the transition (obs_prev, a, obs_next, r, done) is stored unless this was step max_step of the
 *     episode (train_all.py:346-347: executed, not stored, episode ends without done);
 *   - an episode that ended (done or time limit) is reset in place and obs_io receives the reset obs.
 * obs_io [n][13] in: previous observation, out: next observation for the policy.
 * ring [cap][32] rows (s13 a4 s'13 r done), *total = transitions ever stored (slot = total % cap).
 * stats[9] += {episodes, kills(episode_success), fire_success episodes, time-limit ends, fires, good fires,
 *               locked steps, env steps, steps with a non-finite action}. */
void ox_env_step_batch(OxEnv* envs, int64_t n, const float* actions, float* obs_io, float* reward, uint8_t* done,
                       int8_t* success, int max_step, int auto_reset, int randomize, uint64_t seed,
                       uint32_t env_id0, uint32_t* episode_ctr, float* ring, int8_t* ring_succ, int64_t cap,
                       uint64_t* total, uint64_t* stats) {
    for (int64_t i = 0; i < n; ++i) {
        OxEnv* E = &envs[i];
        float prev[13], nobs[13];
        float act[4];
        const int bad_act = ox_sanitize_action(actions + i * 4, act);
        memcpy(prev, obs_io + i * 13, sizeof prev);
        ox_env_step(E, act, nobs, &reward[i], &done[i], &success[i]);
        uint32_t ep = E->counters & 0xFFFFu;
        int trunc = max_step > 0 && (int)ep >= max_step;
        if (ring && !trunc) {
            float* row = ring + (int64_t)(*total % (uint64_t)cap) * 32;
            memcpy(row, prev, 13 * 4);
            memcpy(row + 13, act, 4 * 4);
            memcpy(row + 17, nobs, 13 * 4);
            row[30] = reward[i];
            row[31] = done[i] ? 1.0f : 0.0f;
            if (ring_succ) ring_succ[*total % (uint64_t)cap] = success[i];
            *total += 1;
        }
        if (stats) {
            if (E->flags & F_FIRED) stats[4] += 1;
            if (success[i] == 1) stats[5] += 1;
            if (E->flags & F_LOCKED) stats[6] += 1;
            stats[7] += 1;
            if (bad_act) stats[8] += 1;
        }
        if (auto_reset && (done[i] || trunc)) {
            if (stats) {
                stats[0] += 1;
                if (E->flags & F_EPISODE_SUCCESS) stats[1] += 1;
                if (E->flags & F_FIRE_SUCCESS) stats[2] += 1;
                if (!done[i]) stats[3] += 1;
            }
            int scen = (int)((E->flags >> F_SCEN_SHIFT) & 3u);
            episode_ctr[i] += 1u;
            ox_env_reset(E, scen, randomize, seed, env_id0 + (uint32_t)i, episode_ctr[i]);
            ox_env_observe(E, nobs);
        }
        memcpy(obs_io + i * 13, nobs, sizeof nobs);
    }
}

/* ---- expert labelling: obs-only reward / termination   HarfangEnv_GYM.py:299-336 ----------------- */
float ox_get_reward(const float* s, const float* a, const float* ns, int8_t* success) {
    float x = ns[0] * 10000.0f, y = ns[1] * 10000.0f, z = ns[2] * 10000.0f;
    float loc_diff = sqrtf((x * x + y * y) + z * z); /* :299-301 */
    float r = 0.0f;
    int8_t sc = 0;
    r = r - 0.0001f * loc_diff;  /* :309 */
    r = r - ns[6] * 10.0f;       /* :312 */
    if (a[3] > 0.0f) {           /* :315 */
        r = r - 8.0f;
        if (s[8] > 0.0f && s[7] < 0.0f) sc = -1;
        else if (s[8] > 0.0f && s[7] > 0.0f) sc = 1;
    }
    if ((double)ns[12] < 0.1) r = r + 600.0f; /* :327-328 */
    *success = sc;
    return r;
}
int ox_get_termination(const float* ns) { return (double)ns[12] <= 0.1; } /* :332-336 */

int ox_sizeof_env(void) { return (int)sizeof(OxEnv); }
