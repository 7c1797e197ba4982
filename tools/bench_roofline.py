"""Roofline records of bench.py.

The rule: `frac` is always a fraction of a roof the kernel actually HAS, so 0 < frac < 1.
  * fp32-MFMA kernels (update launches; acting below `x9_rows` rows outside the front launch): algorithmic fp32 FLOPs against the 157.3 TFLOP/s
    dense fp32 matrix peak.
  * kernels whose 256 -> 512 product runs in the EXACT-SPLIT format (six bf16 MFMAs per fp32 product): the FLOPs the matrix cores EXECUTE against
    the 2,500 TFLOP/s dense bf16 peak.  The fp32 arithmetic the launch delivers is reported beside it as `fp32_equivalent` with a `ratio_to_peak`
    against 157.3 — a ratio, not a fraction: the split lets it exceed what v_mfma_f32_16x16x4_f32 could do (1.04 at 131,072 envs in round 5).
  * bf16 kernels: algorithmic FLOPs against the bf16 peak.
HBM roofs: algorithmic bytes (550 B per env step with the fused insert, + the five networks of launches A / B for the front launch) against 8 TB/s;
`traffic` = the committed rocprofv3 --pmc passes of that launch (profiles/pmc_env_traffic.json), never invented.
"""
import ctypes
import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP32_MATRIX_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 dense peak, same guide
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, same guide
ENV_BYTES_FUSED = 550            # algorithmic bytes per env-step with the fused replay insert
ENV_BYTES_PLAIN = 370
ACTOR_FLOP = 272896              # forward FLOPs per sample (2 * MAC, GEMMs only), SURVEY.md 8d
LEARN_FLOP_PER_SAMPLE = 3810816  # HIRL-soft learn(), averaged over the actor-every-2nd alternation, SURVEY.md 8d
POLICY_FLOP_SAC = 2 * (13 * 256 + 256 * 512 + 512 * 8)  # GaussianPolicy forward, GEMMs only
PRODUCT_FLOP = 2 * 256 * 512     # the 256 -> 512 product per row: what the exact-split format executes SIX times
X9_TERMS = 6
FRONT_NET_BYTES = 4 * (138756 + 4 * 138244)  # target actor, critic x 2, target critic x 2: read once by launches A and B


def stamped_us(lib, pairs):
    """elapsed us of (begin, end) hipExtLaunchKernelGGL event pairs"""
    out = []
    for a, b in pairs:
        us = ctypes.c_float()
        lib.call("hx_event_elapsed_us", a, b, ctypes.byref(us))
        out.append(us.value)
    return out


def stamped_env_us(env, actions, launches):
    """The env-step kernel's OWN duration (begin/end stamps of hipExtLaunchKernelGGL, what rocprofv3 reports) over `launches` launches."""
    import torch

    from hirl4ucav_amd import _lib

    L = _lib.load()
    evs = [(ctypes.c_void_p(L.hx_event_create()), ctypes.c_void_p(L.hx_event_create())) for _ in range(launches)]
    for s, e in evs:
        env.time_next_steps(s, e)
        env.step(actions)
    env.time_next_steps(None, None)
    torch.cuda.synchronize()
    us = stamped_us(_lib, evs)
    for s, e in evs:
        L.hx_event_destroy(s)
        L.hx_event_destroy(e)
    return us


def env_sweep(device):
    """The env-step kernel with the fused insert over 4k..4M envs per launch: the kernel's own duration, algorithmic 550 B per env-step."""
    import torch

    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay

    out = []
    for n in (4096, 65536, 1 << 20, 1 << 22):
        rep = DeviceReplay(max(2 * n, 1 << 20), device)
        env = BatchedHarfangEnv(n, scenario="straight_line", device=device, seed=0, max_step=1500, replay=rep)
        env.reset()
        a = torch.rand(n, 4, device=device) * 2 - 1
        for _ in range(3):
            env.step(a)
        us = float(np.median(stamped_env_us(env, a, 16)))
        out.append({"envs_per_launch": n, "us": round(us, 2), "GBps": round(ENV_BYTES_FUSED * n / us / 1e3, 1),
                    "frac": round(ENV_BYTES_FUSED * n / us / 1e3 / HBM_PEAK_GBPS, 4)})
        del env, rep
        torch.cuda.empty_cache()
    return out


def profile_traffic(key):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, gfx950 FETCH x2
    calibration: tools/pmc_env.py).  Not measured by this run: reported under its own key with the file it came from."""
    f = os.path.join(REPO, "profiles", "pmc_env_traffic.json")
    if not os.path.exists(f):
        return None
    with open(f) as fh:
        doc = json.load(fh)
    pmc = doc.get(str(key))
    if not pmc:
        return None
    return {"bytes": pmc["traffic_bytes"], "fetch_bytes": pmc["fetch_bytes"], "write_bytes": pmc["write_bytes"], "ratio_to_algorithmic": pmc["ratio"],
            "source": "profiles/pmc_env_traffic.json", "kernel_build": doc.get("kernel_build"), "measured_at_commit": doc.get("measured_at_commit"),
            "note": "separate rocprofv3 --pmc passes of tools/pmc_env.py at this size; a profile artefact, not a measurement of this run"}


def env_kernel_roof(envs, kern_us):
    """the stand-alone env-step kernel (HBM-bound): mean of its stamped launches, like the rocprofv3 --stats average it must agree with"""
    us = float(np.mean(kern_us))
    pmc = profile_traffic(envs)
    gb = ENV_BYTES_FUSED * envs / us / 1e3
    return {"kernel": "env_step_kernel<PAIR, INSERT, EPB> (hx_env.hip)", "bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(gb / HBM_PEAK_GBPS, 4), "traffic": (pmc or {}).get("bytes"), "bytes_per_launch": ENV_BYTES_FUSED * envs,
            "us_per_launch": round(us, 2), "launches_timed": len(kern_us),
            "timing": "the kernel's own begin/end stamps (hipExtLaunchKernelGGL events) on the launch stream, mean over the launches of "
                      "the second pass in which act and env step are issued as two launches",
            "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc passes of this kernel at this size (not collectable inside this "
                            "process): see traffic_from_profiles; null when no pass exists for the size",
            "traffic_from_profiles": pmc}


def matrix_roof(flop_alg, product_rows, us, fmt):
    """The MFMA side of a launch.  fmt: "f32" (fp32 MFMA), "bf16", or "x9" (fp32 operands, the 256 -> 512 product of `product_rows` rows as SIX
    bf16 MFMAs per fp32 product).  -> the record, whose `frac` is a fraction of the peak of the instruction the product really runs on."""
    if fmt == "x9":
        executed = flop_alg + (X9_TERMS - 1) * PRODUCT_FLOP * product_rows
        tf, eq = executed / us / 1e6, flop_alg / us / 1e6
        return {"bound": "mfma", "achieved": round(tf, 2), "peak": BF16_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / BF16_MATRIX_PEAK_TFLOPS, 4),
                "flop_per_launch": executed,
                "what": "EXECUTED FLOPs against the dense bf16 matrix peak: the acting rows' 256 -> 512 product runs as six bf16 MFMAs per fp32 product "
                        "(exact hi | mid | lo split of both operands; the three partial products below fp32 resolution are not formed), everything else "
                        "(layer 1, the update's forward passes) counted once",
                "fp32_equivalent": {"flop_per_launch": flop_alg, "achieved": round(eq, 2), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "ratio_to_peak": round(eq / FP32_MATRIX_PEAK_TFLOPS, 4),
                                    "what": "the fp32 arithmetic the launch delivers against the fp32 matrix peak — a ratio, not a roofline fraction: "
                                            "the product does not run on v_mfma_f32_16x16x4_f32, so it may exceed 1"}}
    peak = BF16_MATRIX_PEAK_TFLOPS if fmt == "bf16" else FP32_MATRIX_PEAK_TFLOPS
    tf = flop_alg / us / 1e6
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "flop_per_launch": flop_alg}


def acting_format(args, loop):
    """"x9" / "bf16" / "f32": the matrix-core format of the acting rows' 256 -> 512 product in the launch the timed loop runs"""
    eng = loop.eng
    if args.agent == "sac":
        return "x9" if (getattr(eng, "x9_rows", None) is not None and args.envs >= eng.x9_rows) else "f32"
    if args.dtype in ("bf16", "bf16_policy"):
        return "bf16"
    if args.dtype == "f32x9":
        return "x9"
    if loop.front and eng.front_x9:
        return "x9"
    return "x9" if (eng.x9_rows is not None and args.envs >= eng.x9_rows) else "f32"


FRONT_KERNELS = {
    "sac": "actps_sac_front_kernel<MODE> (hx_front.hip): the persistent streaming acting workgroups (Gaussian policy + env step + fused replay insert) "
           "with the first forward launch of learn() (policy(s'), policy(s), Q1/Q2(s, a)) behind them",
    "tile": "act_front_kernel<RELU, X3, BF16> (hx_front.hip): the acting workgroups (32 rows each: policy inference + env step + fused replay insert) on "
            "half of the CUs, launches A and B of learn() (target actor, critics; target critics) on the other half",
    "persistent": "actp_front_kernel<RELU> / actps_front_kernel<RELU> (hx_front.hip): persistent acting workgroups (bf16: weight-stationary, two thirds of "
                  "the CUs; exact split: one 64-row pass each) with the env step + fused replay insert in their tail, launches A and B of learn() on the "
                  "CUs they leave",
}


def launch_roofline(args, loop, fused_us):
    """`roofline` of the JSON line: the launch the timed loop RUNS (policy inference + env step + replay insert in one launch; in the front loop +
    launches A and B of learn()).  Both roofs are quoted; `bound` names the nearer."""
    us = float(np.mean(fused_us))
    sac, front = args.agent == "sac", loop.front
    fmt = acting_format(args, loop)
    if front:  # (the front launch has PMC passes of its own)
        pmc_key = f"front_{args.envs}" if args.dtype == "f32" else f"front_bf16_{args.envs}" if args.dtype == "bf16" else None
    else:
        pmc_key = f"fused_{args.envs}" if args.dtype == "f32" else f"fused_bf16_{args.envs}" if args.dtype == "bf16" else None
    pmc = profile_traffic(pmc_key) if (pmc_key and not sac) else None
    flop = (POLICY_FLOP_SAC if sac else ACTOR_FLOP) * args.envs
    if front and not sac:  # + the forward passes of launches A and B over the minibatch: 3 + 2 nets on a critic-only call, 4 + 4 on an actor call
        flop += int(6.5 * args.batch * ACTOR_FLOP)
    nbytes = ENV_BYTES_FUSED * args.envs + (FRONT_NET_BYTES if (front and not sac) else 0)
    gb = nbytes / us / 1e3
    hbm = {"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBPS, 4), "bytes_per_launch": nbytes}
    mfma = matrix_roof(flop, args.envs, us, fmt)
    first, second = (mfma, hbm) if mfma["frac"] >= hbm["frac"] else (hbm, mfma)
    persistent = args.envs > 8192
    if front and sac:
        name = FRONT_KERNELS["sac"]
    elif front:
        name = FRONT_KERNELS["tile" if args.envs <= (4096 if args.dtype == "bf16" else 8192) else "persistent"]
        name += "; FLOPs: the policy's over the envs + the 6.5 forward passes (average) of launches A and B over the minibatch"
    elif persistent:
        name = ("act_persist_*_kernel<..., ENV = true> (hx_actp.hip): persistent workgroups (one per CU) looping over their row tiles, env step + "
                "fused replay insert in the launch's tail")
    else:
        name = "act_fused_kernel<NRT, GAUSS, ENV = true, ...> (hx_act.hip): policy inference + env step + fused replay insert"
    stamped = ("2 of every 4 steps: every 4th issues act, env step and learn() as separate launches, and the front launch behind it is not stamped"
               if front else "3 of every 4 steps")
    note = ("bound by CU time: 128 acting workgroups of 32 rows beside 320-448 update workgroups on the other 128 CUs; neither roof is near "
            "(DESIGN.md section 4 K5)" if front else
            "vector-issue / LDS bound tile loop (LayerNorm + head per row), DESIGN.md section 4" if persistent else
            "latency-bound at this size: 256 workgroups, one round; neither roof is near (DESIGN.md section 4)")
    return {"kernel": name + ", the dominant kernel of the timed loop", **first, "acting_format": fmt, "traffic": (pmc or {}).get("bytes"),
            "other_roof": second, "us_per_launch": round(us, 2), "launches_timed": len(fused_us),
            "timing": "the kernel's own begin/end stamps (hipExtLaunchKernelGGL events) on the launch stream, mean over the fused launches of the "
                      f"second pass ({stamped})",
            "note": note,
            "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc passes of this launch (HIRL, this policy format, this size; "
                            "FETCH_SIZE calibrated x2, WRITE_SIZE: tools/pmc_env_passes.sh); each of the 8 XCDs pulls the policy's weights into its own "
                            "L2 once per launch, hence a few x the env's 550 B/env-step at small sizes; null where no pass exists",
            "traffic_from_profiles": pmc}


def update_roof(args, learn_us):
    """learn() as the second pass sees it (torch events around sample + learn, median): fp32 / bf16 MFMA, algorithmic FLOPs"""
    peak = BF16_MATRIX_PEAK_TFLOPS if args.dtype == "bf16" else FP32_MATRIX_PEAK_TFLOPS
    tf = LEARN_FLOP_PER_SAMPLE * args.batch / learn_us / 1e6
    return {"kernels": "fwd_l2/bwd_l2/wgrad(+adam) (one learn, minibatch draw included)", "bound": "mfma", "unit": "TFLOP/s", "achieved": round(tf, 3),
            "peak": peak, "frac": round(tf / peak, 5), "us_per_learn": round(learn_us, 2), "samples_per_s": round(args.batch / learn_us * 1e6, 1),
            "timing": "torch events around learn() in the second pass (median)"}


def act_roof(args, loop, act_us):
    """the acting kernel as its own launch (ENV = false; every 4th step of the second pass, outside the front launch: the engine's size rule picks
    the format)"""
    sac = args.agent == "sac"
    eng = loop.eng
    if sac:
        fmt = "x9" if (getattr(eng, "x9_rows", None) is not None and args.envs >= eng.x9_rows) else "f32"
    elif args.dtype in ("bf16", "bf16_policy"):
        fmt = "bf16"
    else:
        fmt = "x9" if (args.dtype == "f32x9" or (eng.x9_rows is not None and args.envs >= eng.x9_rows)) else "f32"
    r = matrix_roof((POLICY_FLOP_SAC if sac else ACTOR_FLOP) * args.envs, args.envs, act_us, fmt)
    r.update({"kernels": "the acting kernel (ENV = false) as its own launch (every 4th step of the second pass)", "acting_format": fmt, "us": round(act_us, 2),
              "timing": "torch events (median)"})
    return r
