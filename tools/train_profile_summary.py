#!/usr/bin/env python3
"""Per-step kernel breakdown of a `rocprofv3 --kernel-trace` run of tools/bench_train.py: the last ten steps (delimited by the OT distance
kernel that opens each step), time per kernel name per step, launches per step and the step span on the device.

    python tools/train_profile_summary.py <rocprof output dir> [rows]"""
import csv,re,glob,collections,sys
f=glob.glob(sys.argv[1]+"/*/*_kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'ot_dist' in n]
def short(n):
    n=re.sub(r'\(fc::.*|\(float.*|\(.*','',n); n=n.replace('void ','').replace('fc::',''); return n[:58]
tot=collections.defaultdict(float); cnt=collections.Counter()
nst=10
a,b=idx[-1-nst],idx[-1]
for r in rows[a:b]:
    k=short(r['Kernel_Name']); tot[k]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3/nst; cnt[k]+=1
print("span/step", round((int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3/nst,1), "sum", round(sum(tot.values()),1), "launches", (b-a)/nst)
for k,v in sorted(tot.items(), key=lambda kv:-kv[1])[:int(sys.argv[2]) if len(sys.argv)>2 else 30]: print(f"{v:8.1f} us  {cnt[k]/nst:5.1f} x {v/(cnt[k]/nst):6.1f}  {k}")
