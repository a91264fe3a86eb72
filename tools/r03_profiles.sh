#!/bin/bash
# Round-3 profile set (run on the GPU box from the repo root; summaries land under gpurun_out/profiles_r03/<name>/ and
# are copied into profiles/ by hand afterwards - gpurun only brings gpurun_out/ back).
#   r03_eval_uni   default bench.py (cfg2 evaluate): kernel trace + PMC passes + bench line
#   r03_jac_uni    bench.py --op jacobian (spin-up on eval_uni: the jacobian's statistics are steady state)
#   r03_cfg5       tools/cfg5_only.py: kernel trace, FETCH_SIZE / WRITE_SIZE (separate passes), MFMA / VALU / LDS counters
#   r03_cfg4_tess  tools/cfg4_tess.py: kernel trace, FETCH_SIZE / WRITE_SIZE
export TMPDIR=/tmp
dst=$PWD/gpurun_out/profiles_r03
mkdir -p $dst
bash tools/prof.sh r03_eval --steps 200 > /dev/null 2>&1
python3 tools/save_profile.py gpurun_out/prof_r03_eval $dst/r03_eval_uni > /dev/null
bash tools/prof.sh r03_jac --op jacobian --steps 200 > /dev/null 2>&1
python3 tools/save_profile.py gpurun_out/prof_r03_jac $dst/r03_jac_uni > /dev/null
bash tools/cfg5_pmc.sh r03 > /dev/null 2>&1
python3 tools/save_profile.py gpurun_out/prof_cfg5_r03 $dst/r03_cfg5 > /dev/null
python3 tools/cfg5_stages.py > $dst/r03_cfg5/stage_times.txt 2>&1
out=$PWD/gpurun_out/prof_cfg4_r03
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/cfg4_tess.py > $out/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc1 -- python3 tools/cfg4_tess.py > $out/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc2 -- python3 tools/cfg4_tess.py > $out/p2.log 2>&1
python3 tools/save_profile.py $out $dst/r03_cfg4_tess > /dev/null
cp $out/bench_trace.log $dst/r03_cfg4_tess/run.log
ls -R $dst | head -40
