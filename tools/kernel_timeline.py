import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_sweep_erm' in r['Kernel_Name']]
a=idx[-3]; b=idx[-2]
prev_end=None
for r in rows[a:b+1]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    name=r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0][:40]
    gap=(s-prev_end)/1e3 if prev_end else 0
    print(f"{name:42s} gap {gap:8.1f} us  dur {(e-s)/1e3:8.1f} us")
    prev_end=e
