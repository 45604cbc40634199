#!/usr/bin/env python3
"""Timeline of ONE step out of a rocprofv3 --kernel-trace csv: every dispatch between the last two launches of a marker
kernel (default: embed_kernel, the first kernel of a forward), with its duration and the gap to the previous dispatch.

    python tools/step_trace.py <kernel_trace.csv> [--marker embed_kernel] [--summary]
"""
import argparse, csv, re, collections
ap = argparse.ArgumentParser()
ap.add_argument("trace"); ap.add_argument("--marker", default=r"(^|::| )embed_kernel<", help="regex of the kernel that starts a step"); ap.add_argument("--summary", action="store_true")
ap.add_argument("--back", type=int, default=0, help="take the step that ends N markers before the last one (bench.py --mode train ends with `steps` event-timed replays on one stream)")
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"gemm_nt_kernel<(\d), (\w+), (\w+), (\w+), (\w+), (\w+), (\w+)(?:, (\w+))?>", n)
    if m:
        act, st, ta, tb, add, sk, ws, grp = m.groups()
        kind = "dW" if ta == "true" else ("dX" if tb == "true" else "fwd")
        return f"gemm[{kind}{' act' + act if act != '0' else ''}{' +R' if add == 'true' else ''}{' SK' if sk == 'true' else ''}{' GRP' if grp == 'true' else ''}]"
    return n.split("(")[0][:48]
marks = [i for i, r in enumerate(rows) if re.search(a.marker, r["Kernel_Name"])]
lo, hi = (marks[-2 - a.back], marks[-1 - a.back]) if len(marks) >= 2 + a.back else (0, len(rows))
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); prev_end = t0
tot = collections.OrderedDict(); gaps = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = short(r["Kernel_Name"])
    gap = s - prev_end
    gaps += max(gap, 0)
    if not a.summary:
        print(f"{(s - t0) / 1e3:10.1f} us  {nm:52s} {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
    d = tot.setdefault(nm, [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
    prev_end = max(prev_end, e)
span = (prev_end - t0) / 1e3
# concurrency: time with at least two kernels in flight (two streams), by queue
evts = sorted([(int(r["Start_Timestamp"]), 1) for r in step] + [(int(r["End_Timestamp"]), -1) for r in step])
depth, last, overlap_ns = 0, evts[0][0], 0
for t, dlt in evts:
    if depth >= 2:
        overlap_ns += t - last
    depth += dlt
    last = t
queues = collections.Counter(r.get("Queue_Id", "?") for r in step)
print(f"--- step span {span:.1f} us, {len(step)} dispatches, sum of gaps {gaps / 1e3:.1f} us; >= 2 kernels in flight for {overlap_ns / 1e3:.1f} us; dispatches by queue {dict(queues)}")
for nm, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{nm:52s} n={n:4d} total {us:9.1f} us  avg {us / n:8.1f} us  {100 * us / span:5.1f} %")
