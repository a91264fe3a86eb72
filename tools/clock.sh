#!/bin/bash
# effective shader clock of the bench kernel: GRBM_GUI_ACTIVE / 8 / kernel time (MI355X_MICROARCH.md, DVFS)
export TMPDIR=/tmp
out=$PWD/gpurun_out/clock_$1; shift
mkdir -p $out
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p -- python3 bench.py --no-cpu --steps 5 --warmup 1 "$@" > $out/log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$out/p/*/*_counter_collection.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'bsk::' in r['Kernel_Name']]
g=[float(r['Counter_Value']) for r in rows if r['Counter_Name']=='GRBM_GUI_ACTIVE']
k=glob.glob('$out/p/*/*_kernel_trace.csv')
dur=[]
if k:
    for r in csv.DictReader(open(k[0])):
        if 'bsk::' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
print('GRBM_GUI_ACTIVE avg', sum(g)/len(g), 'n', len(g), 'kernel ns avg', sum(dur)/max(1,len(dur)))
if dur: print('clock GHz ~', sum(g)/len(g)/8/(sum(dur)/len(dur)))
PY
