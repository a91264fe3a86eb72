#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args...]
# kernel-trace stats + two separate PMC passes, summaries under gpurun_out/prof_<tag>/
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu --no-extra "$@" > $out/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc1 -- python3 bench.py --no-cpu --no-extra --steps 3 --warmup 1 --spinup 20 "$@" > $out/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/pmc2 -- python3 bench.py --no-cpu --no-extra --steps 3 --warmup 1 --spinup 20 "$@" > $out/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- python3 bench.py --no-cpu --no-extra --steps 3 --warmup 1 --spinup 20 "$@" > $out/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc4 -- python3 bench.py --no-cpu --no-extra --steps 3 --warmup 1 --spinup 20 "$@" > $out/bench_pmc4.log 2>&1
find $out -name "*.csv" | head -50
