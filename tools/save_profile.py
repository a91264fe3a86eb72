"""Condense a gpurun_out/prof_<tag> (tools/prof.sh) or pmc_<tag> (tools/pmc.sh) directory into
profiles/<name>/: kernel_stats.csv (rocprofv3 --kernel-trace --stats) and pmc_per_launch.json
(average per launch of every collected counter, bsk:: kernels only)."""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
out = {}
for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "bsk::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open(os.path.join(dst, "pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
for f in glob.glob(os.path.join(src, "bench_trace.log")):
    lines = [l for l in open(f) if l.startswith("{")]
    if lines:
        open(os.path.join(dst, "bench_line.json"), "w").write(lines[-1])
print(json.dumps(out, indent=1)[:1500])
