"""One steady software-pipelined step as a kernel timeline (S = the sampling chain, G = stage G), from a rocprofv3 --kernel-trace
of bench.py: python profiles/micro/trace_step.py DIR_WITH_kernel_trace.csv  (DESIGN.md section 8 item 0)."""
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
S_names=('fps_','bq_index','gather_centres','fps_prefix','bq_query2_kernel<2','bq_query2_kernel<1','ball_query_kernel','index_gather')
fps=[i for i,r in enumerate(rows) if 'fps_indexed_kernel<8, 32' in r['Kernel_Name']]
starts=[int(rows[i]['Start_Timestamp']) for i in fps]
gaps=[round((b-a)/1e3) for a,b in zip(starts,starts[1:])]
print(gaps)
# a steady replayed step: one whose period is below 3.5 ms, from the middle
cands=[k for k,g in enumerate(gaps) if g<3500]
k=cands[len(cands)//2]
a=fps[k]; b=fps[k+1]
t0=int(rows[a]['Start_Timestamp'])
win=[r for r in rows if t0-150000 <= int(r['Start_Timestamp']) < int(rows[b]['Start_Timestamp'])-150000]
for r in win:
    n=r['Kernel_Name']
    tag='S' if any(x in n for x in S_names) else 'G'
    print('%s %9.1f %9.1f %8.1f  %s' % (tag, (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, n[:58]))
