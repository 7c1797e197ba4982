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
agg, runs = {}, {}
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
                runs[(env, seed, agent, loop)] = (mean, kr if kr is not None else float('nan'))


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

# the same comparison PAIRED: front and reference runs of one (scenario, seed) share the expert set, the BC checkpoint and every seed
print()
print("Paired by (scenario, seed): front minus reference, mean over the pairs +- its standard error (sd / sqrt(pairs)); |mean| <= 2 standard errors = no difference shown:")
print()
print("| agent | pairs | d(mean validation success) | d(training kill rate) | difference shown? |")
print("|---|---|---|---|---|")
for agent in ("HIRL-soft", "TD3"):
    pairs = [(runs[k], runs[(k[0], k[1], agent, "reference")]) for k in runs if k[2] == agent and k[3] == "front" and (k[0], k[1], agent, "reference") in runs]
    if len(pairs) < 2:
        continue
    cells, shown = [], False
    for i in (0, 1):
        d = [a[i] - b[i] for a, b in pairs if a[i] == a[i] and b[i] == b[i]]
        m = sum(d) / len(d)
        se = (sum((x - m) ** 2 for x in d) / (len(d) - 1)) ** 0.5 / len(d) ** 0.5
        cells.append(f"{m:+.3f} +- {se:.3f}")
        shown = shown or abs(m) > 2 * se + 1e-12
    print(f"| {agent} | {len(pairs)} | {cells[0]} | {cells[1]} | {'YES' if shown else 'no'} |")
