"""Markdown tables of a round's measurement set (tools/measure_round.sh <tag> -> gpurun_out/<tag>/ or profiles/<tag>_*): the lines DESIGN.md section 8 quotes.
    python tools/results_table.py profiles/r05_            (a file-name prefix: <prefix>bench_default_line_with_sweep_and_cpu.json, ...)"""
import csv
import json
import os
import sys

pre = sys.argv[1]


def load(name):
    p = pre + name
    return json.load(open(p)) if os.path.exists(p) else None


def line(d):
    return f"{d['value'] / 1e6:.1f} M env steps/s, {d['ms_per_step'] * 1e3:.2f} µs, {d.get('update_steps_per_s', 0):,.0f} `learn()`/s"


d = load("bench_default_line_with_sweep_and_cpu.json")
dd = load("bench_driver_form_steps20_warmup5.json")
print("| workload (one GPU, one `learn(B=128)` per vector step) | reference order | **front loop** |")
print("|---|---|---|")
if d:
    print(f"| **configs[1]** 4,096 straight_line, HIRL-soft fp32 — default run (3 × 20,000 steps) | {line(d['reference_order'])} | **{line(d)}** |")
if dd:
    print(f"| … the driver's form `--steps 20 --warmup 5` (3 × 20 steps; repetitions {', '.join('%.1f' % (x * 1e3) for x in dd['repetitions']['ms_per_step'])} µs) | {line(dd['reference_order'])} | **{line(dd)}** |")
rows = {}
p = pre + "bench_other_configs.jsonl"
if os.path.exists(p):
    for ln in open(p):
        try:
            x = json.loads(ln)
        except Exception:
            continue
        c = x["config"]
        key = c["workload"].split(", 1 learn")[0]
        if c.get("batch", 128) != 128:
            key += f" [learn(B = {c['batch']})]"
        if "rccl_ranks" in x or "staged" in json.dumps(c):
            key += " [sharded rank's sequence]"
        if c.get("actions") == "uniform" or "uniform" in c["workload"]:
            key += " [uniform actions]"
        rows.setdefault(key, {})[c.get("loop")] = x
for k, v in rows.items():
    f, r = v.get("front"), v.get("reference order")
    print(f"| {k} | {line(r) if r else '—'} | {('**' + line(f) + '**') if f else '—'} |")
print()
print("| configuration | dominant launch (rocprofv3 average, µs) | share of GPU time | the other launches (average µs × calls) |")
print("|---|---|---|---|")
for tag in ("f32", "f32_reference_order", "bf16", "staged_f32", "sac16k_serpentine", "circ8192_f32", "mixed16k_f32", "mixed16k_bf16", "circ65536_f32", "mixed131072_bf16"):
    p = pre + f"bench_{tag}_kernel_stats.csv"
    if not os.path.exists(p):
        continue
    rs = [r for r in csv.DictReader(open(p)) if "hx" in r["Name"] or "act_" in r["Name"] or "_l2" in r["Name"] or "wgrad" in r["Name"] or "kernel" in r["Name"]]
    rs = [r for r in rs if "elementwise" not in r["Name"] and "rocclr" not in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rs)
    rs.sort(key=lambda r: -float(r["TotalDurationNs"]))

    def nm(r):
        return r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    top = rs[0]
    rest = "; ".join(f"`{nm(r)}` {float(r['AverageNs']) / 1e3:.2f} × {int(r['Calls'])}" for r in rs[1:7])
    print(f"| {tag} | `{nm(top)}` **{float(top['AverageNs']) / 1e3:.2f}** × {int(top['Calls'])} | {float(top['TotalDurationNs']) / tot:.0%} | {rest} |")
if d:
    print()
    r = d["roofline"]
    print(f"roofline (default run): {r['bound']} {r['achieved']} / {r['peak']} {r['unit']} = {r['frac']}; {r['us_per_launch']} µs per launch (live stamp); traffic {r['traffic']}; other roof {r['other_roof']['achieved']} GB/s = {r['other_roof']['frac']}")
    if "fp32_equivalent" in r:  # [r6] frac = executed bf16 FLOPs / bf16 peak; the fp32 arithmetic delivered is a RATIO beside it
        q = r["fp32_equivalent"]
        print(f"fp32-equivalent: {q['achieved']} TFLOP/s = {q['ratio_to_peak']} x the fp32 matrix peak (a ratio, not a roofline fraction)")
    print("every line of the set: roofline.frac / other_roof.frac / roofline_act.frac / roofline_update.frac")
    every = [("default", d)] + ([("driver form", dd)] if dd else []) + [(k + " / " + lp, x) for k, v in rows.items() for lp, x in v.items()]
    worst = 0.0
    for name, x in every:
        fr = [x["roofline"]["frac"], x["roofline"].get("other_roof", {}).get("frac"), x.get("roofline_act", {}).get("frac"), x.get("roofline_update", {}).get("frac")]
        worst = max([worst] + [f for f in fr if f is not None])
        print(f"  {name[:90]:90s} " + " / ".join("-" if f is None else f"{f:.4f}" for f in fr))
    print(f"largest frac of the set: {worst:.4f} (must be < 1)")
    print("env sweep:", " | ".join(f"{x['envs_per_launch']:,}: {x['us']} µs = {x['frac']}" for x in d["roofline_env_sweep"]))
    c = d["cpu_baseline"]
    print(f"cpu baseline: {c['value']:,.0f} env steps/s on {c['cores']} cores ({c.get('sample', '')[:120]}); B0 {c['b0_reference_plumbing']['value']}, B1 {c['b1_batched_cpu']['value']:,.0f}, B2 {c['b2_eager_rocm_learn']['value']}")
