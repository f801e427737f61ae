import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(r["Kernel_Name"].split("(")[0].replace("volume_bwd_","").replace("void ","")[:28], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size_X") or r.get("Grid_Size")) for r in rows if "volume_b" in r["Kernel_Name"]]
out={}
i=0
while i<len(seq):
    if "plan_k" in seq[i][0]:
        j=i+1
        while j<len(seq) and "plan_k" not in seq[j][0] or (j<len(seq) and j-i<3 and seq[j][2]==seq[i][2] and "plan_k" in seq[j][0]): j+=1
        key=" ".join(f"{s[0]}[{s[2]}]" for s in seq[i:j])
        out.setdefault(key,[]).append([s[1] for s in seq[i:j]])
        i=j
    else:
        out.setdefault(seq[i][0]+f"[{seq[i][2]}]",[]).append([seq[i][1]]); i+=1
for k,v in out.items():
    n=len(v); med=[sorted(x[c] for x in v)[n//2] for c in range(len(v[0]))]
    print(f"{n:3d}x  "+"  ".join(f"{a.split('[')[0]}:{m:.0f}" for a,m in zip(k.split(' '),med))+f"   sum {sum(med):.0f} us   grids {[a.split('[')[1][:-1] for a in k.split(' ')]}")
