import csv,sys
rows=[r for r in csv.reader(open(sys.argv[1])) if r and r[0].isdigit()]
idx=[i for i,r in enumerate(rows) if 'prep_queries' in r[1]]
s=idx[-1]
t0=float(rows[s][6]); end=float(rows[-1][6])+float(rows[-1][7])
print("wall", end-t0)
tot={}
for r in rows[s:]:
    n=r[1]
    nm = 'scatter' if 'scatter' in n else 'compact' if 'compact' in n else 'scan_mfma' if 'scan_mfma' in n else 'rescore_score' if 'rescore_score' in n else 'rescore_select' if 'rescore_select' in n else 'scan_stream' if 'scan_stream' in n else 'select_final' if 'select_final' in n else n[:30]
    tot.setdefault(nm,[0,0]); tot[nm][0]+=float(r[7]); tot[nm][1]+=1
for k,v in sorted(tot.items(), key=lambda x:-x[1][0]): print(f"{k:32s} {v[0]:9.1f} us  x{v[1]}")
print([ (r[2], r[7]) for r in rows[s:] if 'scan_mfma' in r[1]])
