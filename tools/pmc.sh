#!/bin/bash
# usage: bash tools/pmc.sh <tag> [env VAR=..] -- quick PMC passes of bench.py (3 steps)
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p1 -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/p2 -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > $out/p2.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ['p1','p2']:
    f=glob.glob('$out/'+d+'/*/*_counter_collection.csv')[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'bsk::' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('$tag',k,len(v), sum(v)/len(v))
PY
