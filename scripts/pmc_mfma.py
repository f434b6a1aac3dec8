"""MFMA-busy fraction per kernel family from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES pass.
Normalisation on MI355X (8 XCDs, 256 CUs, 1024 SIMDs): rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs and
SQ_VALU_MFMA_BUSY_CYCLES summed over all SIMDs, so   busy = MFMA_BUSY / ((GUI_ACTIVE / 8) * 1024).
(The gfx94x derived-metric formula shipped with ROCm 7.2, MFMA_BUSY / (GUI_ACTIVE * CU_NUM * 4), is 8x lower for that reason.)
usage: pmc_mfma.py <dir> [out.txt]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
FAM = ("gemm_nt_v5", "gemm_nt_v4", "gemm_nt_v3", "gemm_tn_grouped_kernel", "conv_v3_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    fam = next((k for k in FAM if k in r["Kernel_Name"]), None)
    if not fam:
        continue
    acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[fam] += 1
lines = [__doc__.split("usage")[0].strip(), "", f"{'kernel family':28s} {'launches':>8s} {'MFMA busy':>10s}   (cycles per launch: GUI_ACTIVE/8)"]
for fam in FAM:
    v = acc[fam]
    if not v:
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    lines.append(f"{fam:28s} {n[fam]:8d} {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):9.1f} %   {cyc / n[fam]:10.0f}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
