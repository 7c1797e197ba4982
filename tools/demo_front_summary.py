"""Summary of tools/demo_front_vs_reference.sh: per scenario, seed, agent and LOOP (front | reference) the validations (last, best, mean over the
run's validations), the training kill rate over the run's last quarter (env episodes that ended with a kill / env episodes that ended, from the
driver's per-episode counters), and per scenario x agent the two loops side by side with the seed-to-seed spread."""
import glob
import os
import re
import sys

root = sys.argv[1]
val = re.compile(r"Validation (\d+): avg reward ([-\d.]+) \(std [-\d.]+\) success ([\d.]+) fire success ([\d.]+)")
epi = re.compile(r"Episode (\d+): .*\| episodes (\d+) kills (\d+) fire-success (\d+) \| ([\d,]+) env steps/s")


def read(p):
    txt = open(p, errors="ignore").read()
    v = [(float(r), float(s), float(f)) for _, r, s, f in val.findall(txt)]
    e = [(int(a), int(b), int(c)) for a, b, c, _, _ in epi.findall(txt)]
    rate = epi.findall(txt)[-1][4] if e else "-"
    kill_rate = None
    if len(e) >= 8:
        q = e[len(e) - 1 - len(e) // 4]  # counters at the start of the last quarter
        kill_rate = (e[-1][2] - q[2]) / max(e[-1][1] - q[1], 1)
    loop = re.findall(r"vector loop: (\w+)", txt)
    return v, kill_rate, rate, (loop[0] if loop else "-")


print("| scenario | seed | agent | loop | validations | last: success / fire success / reward | best success | mean success over validations | "
      "training kill rate, last quarter | env steps/s |")
print("|---|---|---|---|---|---|---|---|---|---|")
agg = {}
for env in ("straight_line", "serpentine", "circular"):
    for sd in sorted(glob.glob(os.path.join(root, env, "seed*"))):
        seed = os.path.basename(sd)[4:]
        for agent, stem in (("BC", "bc"), ("HIRL-soft", "hirl_soft"), ("TD3", "td3")):
            for loop in (("-",) if agent == "BC" else ("front", "reference")):
                p = os.path.join(sd, stem + (".log" if agent == "BC" else f".{loop}.log"))
                if not os.path.exists(p):
                    continue
                v, kr, rate, said = read(p)
                if agent != "BC" and said != loop:
                    print(f"| {env} | {seed} | {agent} | {loop} | LOG SAYS '{said}' | - | - | - | - | - |")
                    continue
                if not v:
                    print(f"| {env} | {seed} | {agent} | {loop} | 0 | - | - | - | - | - |")
                    continue
                last, best, mean = v[-1], max(x[1] for x in v), sum(x[1] for x in v) / len(v)
                print(f"| {env} | {seed} | {agent} | {loop} | {len(v)} | {last[1]:.2f} / {last[2]:.2f} / {last[0]:.1f} | {best:.2f} | {mean:.2f} | "
                      f"{'-' if kr is None else f'{kr:.3f}'} | {rate} |")
                agg.setdefault((env, agent, loop), []).append((last[1], best, mean, kr if kr is not None else float('nan'), last[0]))


def col(xs, i):
    ys = [x[i] for x in xs]
    return f"{sum(ys) / len(ys):.2f} ({min(ys):.2f}..{max(ys):.2f})"


print()
print("Per scenario x agent x loop, mean over seeds (min..max over seeds):")
print()
print("| scenario | agent | loop | seeds | last success | best success | mean success over validations | training kill rate, last quarter | last avg reward |")
print("|---|---|---|---|---|---|---|---|---|")
for (env, agent, loop), xs in agg.items():
    print(f"| {env} | {agent} | {loop} | {len(xs)} | {col(xs, 0)} | {col(xs, 1)} | {col(xs, 2)} | {col(xs, 3)} | {col(xs, 4)} |")
print()
print("Front minus reference (mean over seeds) against the seed-to-seed spread of the reference-order runs (max - min):")
print()
print("| scenario | agent | d(mean validation success) | spread | d(training kill rate) | spread | within the spread? |")
print("|---|---|---|---|---|---|---|")
for (env, agent, loop), xs in agg.items():
    if loop != "front" or (env, agent, "reference") not in agg:
        continue
    ys = agg[(env, agent, "reference")]
    row, ok = [], True
    for i in (2, 3):
        a, b = [x[i] for x in xs], [y[i] for y in ys]
        d = sum(a) / len(a) - sum(b) / len(b)
        spread = max(max(b) - min(b), max(a) - min(a))
        row += [f"{d:+.3f}", f"{spread:.3f}"]
        ok = ok and abs(d) <= spread + 0.02
    print(f"| {env} | {agent} | {row[0]} | {row[1]} | {row[2]} | {row[3]} | {'yes' if ok else 'NO'} |")
