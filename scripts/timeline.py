"""Concurrency / idle analysis of a multi-stream rocprofv3 kernel trace.
usage: python scripts/timeline.py <kernel_trace.csv> [n_last_steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
# step boundaries: adam_kernel ends a step
ends = [e[1] for e in ev if "adam_kernel" in e[2]]
print("steps found:", len(ends))
lo, hi = ends[-3], ends[-1]          # last two complete steps (adam end -> adam end), pack kernel follows adam
sel = [e for e in ev if lo <= e[0] < hi]
span = (hi - lo) / 1e6
print(f"window {span:.3f} ms for 2 steps -> {span/2:.3f} ms/step, {len(sel)/2:.0f} kernels/step")
# sweep line
pts = []
for s, e, *_ in sel:
    pts.append((s, 1)); pts.append((min(e, hi), -1))
pts.sort()
busy = collections.Counter(); cur = 0; last = lo
for t, d in pts:
    busy[cur] += t - last; last = t; cur += d
busy[cur] += hi - last
tot = sum(busy.values())
for k in sorted(busy):
    print(f"  {k} kernels in flight: {busy[k]/1e6/2:.3f} ms/step ({100*busy[k]/tot:.1f} %)")
# per queue busy time
perq = collections.defaultdict(float)
for s, e, n, q, st in sel:
    perq[(q, st)] += (min(e, hi) - s) / 1e6 / 2
for k, v in sorted(perq.items(), key=lambda x: -x[1]):
    print(f"  queue/stream {k}: {v:.3f} ms/step busy")
# idle gaps: biggest intervals with nothing running
gaps = []; cur = 0; last = lo; prev_name = None
order = sorted([(s, 1, n) for s, e, n, *_ in sel] + [(min(e, hi), -1, n) for s, e, n, *_ in sel])
for t, d, n in order:
    if cur == 0 and t > last:
        gaps.append((t - last, last - lo, prev_name, n))
    cur += d; last = t
    if d == -1: prev_name = n
gaps.sort(reverse=True)
print("largest idle gaps (us, at ms, after kernel -> before kernel):")
for g, at, a, b in gaps[:15]:
    print(f"  {g/1e3:8.1f} us at {at/1e6:7.3f} ms  {str(a)[:50]} -> {str(b)[:50]}")
