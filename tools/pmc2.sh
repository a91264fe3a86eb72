#!/bin/bash
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmcB_$tag
mkdir -p $out
rocprofv3 --pmc SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS --output-format csv -d $out/p1 -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/p2 -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > $out/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL --output-format csv -d $out/p3 -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > $out/p3.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ['p1','p2','p3']:
    fs=glob.glob('$out/'+d+'/*/*_counter_collection.csv')
    if not fs: print(d,'no output'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'bsk::' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('$tag',k, sum(v)/len(v))
PY
