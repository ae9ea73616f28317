"""Idle time of the GPU inside the timed steps of a `rocprofv3 --kernel-trace` run: the union of all kernel intervals
(all streams) against the wall span, the largest gaps and the kernels that precede them.
  python tools/trace_gaps.py <dir with *_kernel_trace.csv> [first-kernel-substring of a step, default adam_kernel]"""
import collections
import csv
import glob
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if marker in r[2]]
if len(ends) < 2:
    sys.exit("fewer than two %s launches" % marker)
lo, hi = ends[-2] + 1, ends[-1]                  # the last complete step: after one Adam up to the next
step = rows[lo:hi + 1]
t0, t1 = step[0][0], max(r[1] for r in step)
busy, cur_end, gaps = 0, t0, []
last = None
for s, e, name in step:
    if s > cur_end:
        gaps.append((s - cur_end, last, name))
        busy += 0
        cur_end = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        last = name
span = t1 - t0
print("step span %.2f ms, some kernel running %.2f ms (%.1f %%), idle %.2f ms in %d gaps" %
      (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
hist = collections.Counter()
for g, _, _ in gaps:
    hist[min(int(g / 1000) // 5 * 5, 100)] += g
print("idle by gap length (us bucket -> ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
by_prev = collections.Counter()
for g, prev, nxt in gaps:
    by_prev[(prev or "")[:50] + " -> " + nxt[:50]] += g
for k, v in by_prev.most_common(12):
    print("%8.2f ms  %s" % (v / 1e6, k))
# time with NO matrix kernel (conv_igemm / conv_wgrad / dcn GEMMs) running, by the kernel that is running instead
ev = []
for s, e, name in step:
    mm = any(k in name for k in ("conv_igemm", "conv_wgrad", "dcn_fprop", "dcn_dgrad", "dcn_wgrad", "head_tail", "rows_gemm"))
    ev.append((s, 1, mm, name)); ev.append((e, -1, mm, name))
ev.sort(key=lambda x: (x[0], x[1]))
n_mm, running, prev_t, no_mm = 0, collections.Counter(), t0, collections.Counter()
for tt, d, mm, name in ev:
    if n_mm == 0 and tt > prev_t:
        share = [k for k, v in running.items() if v > 0]
        for k in share:
            no_mm[k[:60]] += (tt - prev_t) / len(share)
        if not share:
            no_mm["(idle)"] += tt - prev_t
    prev_t = tt
    if mm:
        n_mm += d
    else:
        running[name] += d
tot = sum(no_mm.values())
print("no matrix kernel running: %.2f ms of %.2f" % (tot / 1e6, span / 1e6))
for k, v in no_mm.most_common(14):
    print("%8.2f ms  %s" % (v / 1e6, k))
