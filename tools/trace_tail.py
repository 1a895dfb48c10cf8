import csv,glob,re,sys
d=sys.argv[1]; n=int(sys.argv[2]) if len(sys.argv)>2 else 16
f=sorted(glob.glob(d+'/**/*_kernel_trace.csv', recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if any(k in r['Kernel_Name'] for k in ('block_mapped_kernel','chunk_kernel'))]
for r in sel[-n:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{re.sub(r'<.*','',r['Kernel_Name'])[-22:]:22s} {(e-s)/1e3:8.1f} us")
