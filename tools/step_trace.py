#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace of bench.py (the launches between the last pack_input kernels / RefNet stem launches), aggregated per kernel [and grid]:
launch count, total and average duration, the step's span and busy time (span - busy = gaps between kernels).
    usage: tools/step_trace.py <..._kernel_trace.csv> [g = split by grid] [rows]
    e.g.   rocprofv3 --kernel-trace --stats -d out -o b1 --output-format csv -- python3 bench.py --batch 1 --height 128 --width 128 ...
(profiles/r04_b1_step_kernels*.txt)"""
import csv,collections,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'pack_input' in r['Kernel_Name']]
if len(idx)>=3: a,b=idx[-3],idx[-1]
else:  # round 5 on: the stem conv reads the NCHW inputs itself -- a step starts at RefNet's stem launch (two stem launches per step)
    idx=[i for i,r in enumerate(rows) if 'stem_conv_kernel' in r['Kernel_Name']]
    a,b=idx[-4],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
agg=collections.defaultdict(lambda:[0,0.0])
tot=0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    k=r['Kernel_Name'].split('(')[0][:70]
    if len(sys.argv)>2: k+=f" g{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
    agg[k][0]+=1; agg[k][1]+=(e-s)/1e3; tot+=(e-s)/1e3
print('kernels',b-a,'span us',(int(rows[b]['Start_Timestamp'])-t0)/1e3,'busy',tot)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[3]) if len(sys.argv)>3 else 40]:
    print(f"{v[0]:4d} {v[1]:8.1f} us  avg {v[1]/v[0]:6.1f}  {k}")
