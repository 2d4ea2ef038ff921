"""trace_gaps.py DIR -- from a rocprofv3 --kernel-trace of tools/shard_trace.py: per step (finish to finish), the idle gap in front of each kernel and its duration."""
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows]
# take the last 30 steps: find ring_finish_kernel boundaries
names=[r['Kernel_Name'].split('(')[0].replace('void ','').replace('nbk::','') for r in rows]
idx=[i for i,nm in enumerate(names) if nm.startswith('ring_finish')]
idx=idx[-31:]
agg=collections.defaultdict(list)
for a,b in zip(idx[:-1],idx[1:]):
    t0=int(rows[a]['End_Timestamp'])
    prev_end=t0
    for i in range(a+1,b+1):
        s,e=int(rows[i]['Start_Timestamp']),int(rows[i]['End_Timestamp'])
        key=names[i][:28]+f"#{rows[i]['Stream_Id'] if 'Stream_Id' in rows[i] else ''}"
        agg[(i-a,key)].append(((s-prev_end)/1e3,(e-s)/1e3))
        prev_end=max(prev_end,e)
    agg[('step','total')].append(((int(rows[b]['End_Timestamp'])-t0)/1e3,0))
for k,v in sorted(agg.items(), key=lambda kv: (str(kv[0][0]).zfill(3))):
    gap=sum(x[0] for x in v)/len(v); dur=sum(x[1] for x in v)/len(v)
    print(k, 'gap before %.1f us, duration %.1f us (n=%d)'%(gap,dur,len(v)))
