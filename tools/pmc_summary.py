"""Per-kernel means of rocprofv3 --pmc csv output (counter_collection.csv files under the given directories), per wave:
    python tools/pmc_summary.py gpurun_out/pmc_learn_*
SQ cycle counters tick once per 4 clocks; a wave's life = ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY (executing | ready but not issued:
issue arbitration and dependency stalls | s_waitcnt and barriers)."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
cols = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS",
        "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_MFMA", "SQ_INSTS_BRANCH", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS",
        "SQ_INST_CYCLES_VMEM", "SQC_ICACHE_REQ", "SQC_ICACHE_MISSES", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT"]
print("%-44s %6s " % ("kernel (per wave; cycles in 4-clock units)", "waves") + " ".join("%9s" % c.replace("SQ_", "").replace("INSTS_", "n").replace("ACTIVE_INST", "ACT").replace("WAIT_INST", "WI")[:9] for c in cols))
for k in sorted(acc):
    c = {n: v[0] / v[1] for n, v in acc[k].items()}
    if max(v[1] for v in acc[k].values()) < 20 or "SQ_WAVES" not in c:
        continue
    w = c["SQ_WAVES"]
    print("%-44s %6d " % (k[:44], w) + " ".join("%9.1f" % (c.get(n, float("nan")) / w) for n in cols))
