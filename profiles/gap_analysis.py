"""Launch boundaries inside the captured step, from a rocprofv3 kernel trace (Start / End timestamps of every dispatch):
usage: python profiles/gap_analysis.py <..._kernel_trace.csv> [out.txt]
Steps are cut at `adam_kernel` (the last launch of a step); the replayed steps are those with the most common kernel count.  Per step: sum of kernel
durations, sum of the idle gaps between one kernel's end and the next one's start, and how the gaps are distributed."""
import csv
import sys
from collections import Counter


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], []
    for r in rows:
        cur.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
        if "adam_kernel" in r["Kernel_Name"]:
            steps.append(cur)
            cur = []
    if not steps:
        raise SystemExit("no adam_kernel in the trace")
    n_mode = Counter(len(s) for s in steps).most_common(1)[0][0]
    reps = [s for s in steps if len(s) == n_mode]
    # keep the steps that ran back to back as graph replays: short ones (eager warm-up steps are longer)
    spans = sorted((s[-1][1] - s[0][0]) for s in reps)
    med = spans[len(spans) // 2]
    reps = [s for s in reps if (s[-1][1] - s[0][0]) <= 1.1 * med]
    out = []
    out.append(f"{len(reps)} replayed steps of {n_mode} kernels each (of {len(steps)} steps in the trace)")
    tot_k, tot_g, gaps_all, overl = 0.0, 0.0, [], 0
    for s in reps:
        k = sum(e - b for b, e, _ in s)
        g = 0
        for (b0, e0, _), (b1, e1, _) in zip(s, s[1:]):
            d = b1 - e0
            if d >= 0:
                g += d
                gaps_all.append(d)
            else:
                overl += 1
        tot_k += k; tot_g += g
    n = len(reps)
    span = sum(s[-1][1] - s[0][0] for s in reps) / n
    out.append(f"per step: span first start -> last end {span / 1e3:8.1f} us, sum of kernel durations {tot_k / n / 1e3:8.1f} us, sum of gaps {tot_g / n / 1e3:7.1f} us "
               f"({tot_g / n / (n_mode - 1):.0f} ns per boundary over {n_mode - 1} boundaries; {overl / n:.1f} boundaries per step where the next kernel started before the previous ended)")
    gaps_all.sort()
    q = lambda p: gaps_all[min(len(gaps_all) - 1, int(p * len(gaps_all)))]
    out.append(f"gap distribution (ns): min {gaps_all[0]}, 10 % {q(0.10)}, median {q(0.50)}, 90 % {q(0.90)}, 99 % {q(0.99)}, max {gaps_all[-1]}")
    # which kernels are followed by the largest gaps (mean over the replayed steps)
    after = Counter(); cnt = Counter()
    for s in reps:
        for (b0, e0, n0), (b1, e1, n1) in zip(s, s[1:]):
            if b1 - e0 >= 0:
                key = n1.split("(")[0].replace("void ", "")[:60]
                after[key] += b1 - e0; cnt[key] += 1
    out.append("mean gap in FRONT of a kernel, by kernel (ns; launches per step):")
    for key, v in sorted(after.items(), key=lambda kv: -kv[1])[:14]:
        out.append(f"  {key:62s} {v / cnt[key]:7.0f}   {cnt[key] / n:5.1f}")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
