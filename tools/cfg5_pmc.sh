#!/bin/bash
# usage: bash tools/cfg5_pmc.sh <tag> [BSK_VARIANT] - kernel-trace stats + PMC passes (FETCH_SIZE, WRITE_SIZE, LDS, VALU/MFMA) of cfg5
tag=$1; export TMPDIR=/tmp
[ -n "$2" ] && export BSK_VARIANT=$2
out=$PWD/gpurun_out/prof_cfg5_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/cfg5_only.py > $out/bench_trace.log 2>&1
export CFG5_ITERS=3,5      # the counter passes serialise the dispatches: few launches
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc1 -- python3 tools/cfg5_only.py > $out/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc2 -- python3 tools/cfg5_only.py > $out/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/pmc3 -- python3 tools/cfg5_only.py > $out/p3.log 2>&1
tail -2 $out/p3.log
