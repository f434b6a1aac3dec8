"""Per-layer time of the backbone's implicit-GEMM convolutions from a rocprofv3 kernel trace (single-stream run):
the conv launches of one step in program order, with the layer they belong to.  usage: conv_layers.py <kernel_trace.csv>"""
import csv, sys, importlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bb = importlib.import_module("cmpc-refseg_amd.backbone")
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
conv = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "conv_v3_kernel" in r["Kernel_Name"]]
layers = []
hw = {2: 80, 3: 40, 4: 40, 5: 40}
for stage, suf, b1, cin, mid, cout, stride, dil in bb.block_layout():
    n = f"res{stage}{suf}"
    m = 8 * hw[stage] * hw[stage]
    if b1:
        layers.append((n + "_b1", m, cin, cout, 1))
    layers.append((n + "_2a", m, cin, mid, 1)); layers.append((n + "_2b", m, mid, mid, 3)); layers.append((n + "_2c", m, mid, cout, 1))
per = len(layers)
steps = len(conv) // per
print(f"{len(conv)} conv launches = {steps} steps x {per} layers")
tot = 0
agg = {}
for i, (n, m, cin, cout, k) in enumerate(layers):
    ts = [conv[s * per + i] for s in range(2, steps)]
    t = sum(ts) / len(ts); tot += t
    fl = 2.0 * m * cin * cout * k * k
    by = 2.0 * (m * cin + m * cout + (m * cout if n.endswith("2c") else 0)) + 2.0 * cin * cout * k * k
    key = n[:4] + n[-3:]
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += t; a[2] += fl; a[3] += by
for k, (c, t, fl, by) in agg.items():
    print(f"{k:10s} x{c:3d}  {t:8.1f} us total  {t/c:6.1f} us each  {fl/t/1e6:7.1f} TFLOP/s  {by/t/1e6:6.2f} TB/s (algorithmic bytes)")
print(f"total {tot/1e3:.3f} ms per step")
