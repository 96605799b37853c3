import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:50]
    k=(n,r.get("Grid_Size_X","?"),r.get("Grid_Size_Y","?"))
    a=agg.setdefault(k,[0,0]); a[0]+=1; a[1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,a in agg.items():
    if a[0]>=20: print(f"{k[0]:52s} grid {k[1]:>8s}x{k[2]:<4s} calls {a[0]:5d} avg {a[1]/a[0]/1e3:8.1f} us")
