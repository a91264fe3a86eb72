#!/bin/bash
# round-2 first GPU pass: uniform-knot kernels - parity, timing against the general kernels, PMC
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
python -m pytest tests -x -q -m gpu -k "uniform_knot_path or kernel_variants or evaluate_derivative or jacobian_against or normal_against or full_size or random_shapes" > gpurun_out/r2a/tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r2a/tests.log
tail -5 gpurun_out/r2a/tests.log
python bench.py --no-cpu --no-extra --steps 200 > gpurun_out/r2a/bench_uni.json 2> gpurun_out/r2a/bench_uni.err
BSK_VARIANT=9 python bench.py --no-cpu --no-extra --steps 200 > gpurun_out/r2a/bench_gen.json 2> gpurun_out/r2a/bench_gen.err
python bench.py --no-cpu --no-extra --steps 200 --op jacobian > gpurun_out/r2a/bench_uni_jac.json 2>&1
BSK_VARIANT=9 python bench.py --no-cpu --no-extra --steps 200 --op jacobian > gpurun_out/r2a/bench_gen_jac.json 2>&1
python bench.py --no-cpu --no-extra --steps 200 > gpurun_out/r2a/bench_uni2.json 2>&1
cat gpurun_out/r2a/bench_*.json | cut -c1-400
bash tools/pmc.sh r2a_uni --no-extra > gpurun_out/r2a/pmc_uni.txt 2>&1
cat gpurun_out/r2a/pmc_uni.txt
