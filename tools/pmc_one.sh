#!/bin/bash
# usage: bash tools/pmc_one.sh <tag> <python script> - three PMC passes (issue / wait, LDS, instruction counts) of a one-kernel script
tag=$1; shift
export TMPDIR=/tmp ITERS=3,5
out=$PWD/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/p1 -- python3 "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $out/p2 -- python3 "$@" > $out/p2.log 2>&1
python3 - $out <<'PY'
import csv, glob, collections, sys
for d in ("p1", "p2"):
    for f in glob.glob(f"{sys.argv[1]}/{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "bsk::" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(f"{k:52s} {c:24s} {sum(v) / len(v) / 1e6:10.2f} M")
PY
