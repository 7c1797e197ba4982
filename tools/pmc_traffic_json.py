"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_env.py (one counter per pass, CSV) -> profiles/pmc_env_traffic.json, the file
bench.py reports as `traffic_from_profiles`.
    python tools/pmc_traffic_json.py <dir with pmc_<COUNTER>_env_<n>.csv> <commit> > profiles/pmc_env_traffic.json
FETCH_SIZE is reported in KiB and reads 1/2 of the bytes on gfx950 for this access shape: the factor is CALIBRATED in the same run with the
512 MiB dword copy (MI355X_MICROARCH.md's gfx950 correction); WRITE_SIZE is taken as reported (the copy reads back 1.000)."""
import csv
import glob
import json
import os
import re
import sys

d, commit = sys.argv[1], sys.argv[2]
ALGO = {"env_step_kernel": 550, "act_fused_kernel": 550, "act_persist": 550, "act_front_kernel": 550, "actps_front_kernel": 550, "actp_front_kernel": 550}  # algorithmic bytes per env-step with the fused insert


# the front launch also reads the five networks of launches A and B once (target actor, critic x 2, target critic x 2) beside the acting policy's:
# 4 B x (138,756 + 4 x 138,244) = 2.77 MB; the counter sees every XCD's L2 pull its own copy of all of them
extra = {k: 4 * (138756 + 4 * 138244) for k in ("act_front_kernel", "actps_front_kernel", "actp_front_kernel")}  # (actps_ / actp_: the streaming / persistent acting roles)


def rows(path):
    with open(path) as fh:
        return list(csv.DictReader(fh))


out = {"_comment": "HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_env.py; FETCH_SIZE calibrated with the "
                   "512 MiB dword copy of the same run, WRITE_SIZE as reported). Keys: envs per launch -> the stand-alone env_step_kernel; 'fused_<n>' -> "
                   "the fused act + env launch (policy inference + env step + insert, the kernel of bench.py's timed loop: act_fused_kernel<..., ENV> up to 8,192 envs, "
                   "the persistent act_persist_*_kernel<..., ENV> beyond), 'fused_bf16_<n>' the same with the bf16 policy, 'front_<n>' the front launch (act_front_kernel: the same "
                   "plus launches A and B of learn() on the minibatch; algorithmic bytes = the env step's only); their reads include the policy's W2 per "
                   "workgroup out of L2, which the counter does not see.",
       "kernel_build": "round 5: fp32 acting in the exact-split format from 4,096 rows on (six partial products), bf16 acting kernels with LayerNorm 2 + final layer from the accumulators; env kernel as in round 3", "measured_at_commit": commit}
for f in sorted(glob.glob(os.path.join(d, "pmc_FETCH_SIZE_env_*.csv"))):
    mt = re.search(r"env_(\d+)(_bf16)?\.csv", f)
    n, dt = int(mt.group(1)), (mt.group(2) or "")
    w = f.replace("FETCH_SIZE", "WRITE_SIZE")
    fr, wr = rows(f), rows(w)
    cal_f = [float(r["Counter_Value"]) for r in fr if "calib_copy_dword" in r["Kernel_Name"]]
    cal_w = [float(r["Counter_Value"]) for r in wr if "calib_copy_dword" in r["Kernel_Name"]]
    kib = 128 * 1024 * 1024 * 4 / 1024  # the copy reads and writes 512 MiB
    ff, wf = kib / (sum(cal_f) / len(cal_f)), kib / (sum(cal_w) / len(cal_w))
    for kern, key in (("env_step_kernel", str(n)), ("act_fused_kernel", f"fused{dt}_{n}"), ("act_persist", f"fused{dt}_{n}"), ("act_front_kernel", f"front{dt}_{n}"),
                      ("actps_front_kernel", f"front{dt}_{n}"), ("actp_front_kernel", f"front{dt}_{n}")):
        if kern == "env_step_kernel" and (dt or str(n) in out):
            continue
        fv = [float(r["Counter_Value"]) for r in fr if kern in r["Kernel_Name"]]
        wv = [float(r["Counter_Value"]) for r in wr if kern in r["Kernel_Name"]]
        if not fv or not wv:
            continue
        fv, wv = fv[len(fv) // 2:], wv[len(wv) // 2:]  # the later launches: steady state
        fb, wb = sum(fv) / len(fv) * 1024 * ff, sum(wv) / len(wv) * 1024 * wf
        out[key] = {"fetch_KiB_reported": round(sum(fv) / len(fv), 2), "write_KiB_reported": round(sum(wv) / len(wv), 2), "fetch_factor_calibrated": round(ff, 4),
                    "write_factor_calibrated": round(wf, 4), "fetch_bytes": int(fb), "write_bytes": int(wb), "traffic_bytes": int(fb) + int(wb),
                    "algorithmic_bytes": ALGO[kern] * n + extra.get(kern, 0), "ratio": round((int(fb) + int(wb)) / (ALGO[kern] * n + extra.get(kern, 0)), 4), "launches": len(fv)}
print(json.dumps(out, indent=1))
