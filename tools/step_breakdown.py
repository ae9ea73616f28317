"""Per-kernel time of the LAST complete train step of a `rocprofv3 --kernel-trace` run (best read from a ONE-stream run:
RR_WGRAD_STREAM=0, where kernel durations do not overlap and sum to the step): totals by kernel name, by (name, grid) for
the matrix kernels, and the idle time between kernels.
  python tools/step_breakdown.py <dir with *_kernel_trace.csv> [marker kernel, default adam_kernel]"""
import collections
import csv
import glob
import re
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    grid = (int(r.get("Grid_Size_X", 0)) // max(int(r.get("Workgroup_Size_X", 1)), 1), int(r.get("Grid_Size_Y", 1)), int(r.get("Grid_Size_Z", 1)))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid))
rows.sort()
ends = [i for i, r in enumerate(rows) if marker in r[2]]
step = rows[ends[-2] + 1:ends[-1] + 1]
t0, t1 = step[0][0], max(r[1] for r in step)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]


tot = collections.defaultdict(lambda: [0, 0])
by_grid = collections.defaultdict(lambda: [0, 0])
for s, e, n, g in step:
    k = short(n)
    tot[k][0] += 1; tot[k][1] += e - s
    if "conv_igemm" in k or "conv_wgrad" in k or "conv_dgrad" in k:
        by_grid[(k, g)][0] += 1; by_grid[(k, g)][1] += e - s
busy, cur = 0, t0
for s, e, n, g in step:
    if e > cur:
        busy += e - max(s, cur); cur = e
span = t1 - t0
ksum = sum(v[1] for v in tot.values())
print("step span %.2f ms; kernel time summed %.2f ms; some kernel running %.2f ms; idle %.2f ms; %d launches" %
      (span / 1e6, ksum / 1e6, busy / 1e6, (span - busy) / 1e6, len(step)))
print("--- by kernel")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%9.3f ms %5d x %8.1f us  %s" % (t / 1e6, c, t / c / 1e3, k))
print("--- matrix kernels by grid (workgroups x, y, z)")
for (k, g), (c, t) in sorted(by_grid.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%9.3f ms %5d x %8.1f us  %-60s %s" % (t / 1e6, c, t / c / 1e3, k, g))
