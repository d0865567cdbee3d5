"""Per-step timeline of the training loop from a rocprofv3 kernel trace (tools/train_loop_profile.py under
`rocprofv3 --kernel-trace`): per kernel, in launch order, launches / time / idle gap after it, averaged over steps.
usage: python tools/step_timeline.py <..._kernel_trace.csv> [first_step] [last_step]"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 250)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|naqs::|void ", "", n)
    return n.split("(")[0][:64]
names = [short(r["Kernel_Name"]) for r in rows]
# a step starts at the sampler's head launch (round 5: the finish job rides in the forward launch, so it is no kernel of its own)
fin = [i for i, n in enumerate(names) if "sample_head" in n]
segs = [(fin[i], fin[i + 1]) for i in range(lo, min(hi, len(fin) - 1))]
tot, cnt, gap = collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(float)
wall = 0
for a, b in segs:
    wall += int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    for j in range(a, b):
        n = names[j]
        tot[n] += int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])
        cnt[n] += 1
        gap[n] += int(rows[j + 1]["Start_Timestamp"]) - int(rows[j]["End_Timestamp"])
ns = len(segs)
print(f"{ns} steps: {wall / ns / 1e3:.1f} us/step on the GPU's clock, kernels {sum(tot.values()) / ns / 1e3:.1f} us in "
      f"{sum(cnt.values()) / ns:.1f} launches, idle {sum(gap.values()) / ns / 1e3:.1f} us")
seen = []
a, b = segs[0]
for j in range(a, b):
    if names[j] not in seen:
        seen.append(names[j])
for n in seen:
    print(f"  {n:64s} n={cnt[n] / ns:5.2f}  t={tot[n] / ns / 1e3:7.2f} us  idle after={gap[n] / ns / 1e3:6.2f} us")
