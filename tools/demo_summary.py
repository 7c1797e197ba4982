"""Summary of tools/demo_all.sh: per scenario, seed and agent the LAST validation (success, fire success, mean reward) and the best one."""
import glob
import os
import re
import sys

root = sys.argv[1]
pat = re.compile(r"Validation (\d+): avg reward ([-\d.]+) \(std [-\d.]+\) success ([\d.]+) fire success ([\d.]+)")
print("| scenario | seed | agent | validations | last: success / fire success / reward | best success | env steps/s (last episode line) |")
print("|---|---|---|---|---|---|---|")
agg = {}
for env in ("straight_line", "serpentine", "circular"):
    for sd in sorted(glob.glob(os.path.join(root, env, "seed*"))):
        for agent, f in (("BC", "bc.log"), ("HIRL-soft", "hirl_soft.log"), ("TD3", "td3.log")):
            p = os.path.join(sd, f)
            if not os.path.exists(p):
                continue
            txt = open(p, errors="ignore").read()
            v = pat.findall(txt)
            rate = re.findall(r"\| ([\d,]+) env steps/s", txt)
            if not v:
                print(f"| {env} | {os.path.basename(sd)[4:]} | {agent} | 0 | - | - | - |")
                continue
            last, best = v[-1], max(float(x[2]) for x in v)
            print(f"| {env} | {os.path.basename(sd)[4:]} | {agent} | {len(v)} | {last[2]} / {last[3]} / {last[1]} | {best:.2f} | {rate[-1] if rate else '-'} |")
            agg.setdefault((env, agent), []).append((float(last[2]), best))
print()
print("| scenario | agent | mean last success over seeds | mean best success over seeds |")
print("|---|---|---|---|")
for (env, agent), xs in agg.items():
    print(f"| {env} | {agent} | {sum(a for a, _ in xs) / len(xs):.2f} | {sum(b for _, b in xs) / len(xs):.2f} |")
