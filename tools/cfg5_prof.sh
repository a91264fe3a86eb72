#!/bin/bash
# usage: bash tools/cfg5_prof.sh <tag> [BSK_VARIANT]   - rocprofv3 kernel-trace stats of cfg5 (10 M points)
tag=$1; export TMPDIR=/tmp
[ -n "$2" ] && export BSK_VARIANT=$2
out=$PWD/gpurun_out/cfg5_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/cfg5_only.py $CFG5_N > $out/run.log 2>&1
f=$(find $out/trace -name "*_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>4s}  {r['Name'][:90]}")
PY
