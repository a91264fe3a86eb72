#!/bin/bash
# usage: bash tools/cfg5_stall.sh <tag> [BSK_VARIANT] - wave-level issue / wait counters of the cfg5 kernels (two PMC passes)
tag=$1; export TMPDIR=/tmp; export CFG5_ITERS=${CFG5_ITERS:-3,5}
[ -n "$2" ] && export BSK_VARIANT=$2
out=$PWD/gpurun_out/stall_cfg5_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmcA -- python3 tools/cfg5_only.py > $out/pA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmcB -- python3 tools/cfg5_only.py > $out/pB.log 2>&1
python3 - $out <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "bsk::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void bsk::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
last = None
for (k, c), v in sorted(agg.items()):
    if k != last: print(k); last = k
    print(f"    {c:28s} {sum(v) / len(v) / 1e6:10.2f} M")
PY
