"""One train step of a multi-stream rocprofv3 kernel trace, per stream, in time order (consecutive launches of one kernel merged).
usage: python scripts/trace_step.py <kernel_trace.csv> [step_from_end=2] [min_us=0]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    n = n.split("(")[0]; n = n.replace("unsigned short", "bf16")
    return n[:44]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
ends = [e[0] for e in ev if e[2].startswith("embed_gather_kernel")]          # first launch of a step's forward pass
lo, hi = ends[-back - 1], ends[-back]
sel = [e for e in ev if lo <= e[0] < hi]
print(f"step window {(hi-lo)/1e6:.3f} ms, {len(sel)} kernels")
streams = collections.OrderedDict()
for s, e, n, q in sel: streams.setdefault(q, []).append((s, e, n))
for q, lst in streams.items():
    busy = sum(e - s for s, e, _ in lst) / 1e6
    print(f"--- stream {q}: {len(lst)} kernels, busy {busy:.3f} ms, first {(lst[0][0]-lo)/1e6:.3f} last {(lst[-1][1]-lo)/1e6:.3f}")
    i = 0
    while i < len(lst):
        j = i
        while j + 1 < len(lst) and lst[j + 1][2] == lst[i][2]: j += 1
        s, e = lst[i][0], lst[j][1]
        kb = sum(x[1] - x[0] for x in lst[i:j + 1])
        gap = (lst[i][0] - lst[i - 1][1]) / 1e3 if i else 0.0
        print(f"  {(s-lo)/1e6:7.3f} -> {(e-lo)/1e6:7.3f}  {kb/1e3:8.1f} us  gap {gap:7.1f}  {lst[i][2]} x{j-i+1}")
        i = j + 1
