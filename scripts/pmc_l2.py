"""L2 hit rate per kernel family from one rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass.  usage: pmc_l2.py <dir>"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
FAM = ("gemm_nt_v5", "gemm_nt_v4", "gemm_nt_v3", "gemm_tn_grouped_kernel", "conv_v3_kernel", "mutan_fwd", "mutan_bwd", "adam_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    fam = next((k for k in FAM if k in r["Kernel_Name"]), None)
    if fam:
        acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
for fam in FAM:
    v = acc[fam]
    if v:
        h, m = v["TCC_HIT_sum"], v["TCC_MISS_sum"]
        print(f"{fam:26s} L2 hit rate {100*h/(h+m):5.1f} %   (hits {h:.3g}, misses {m:.3g}; 128-B requests)")
