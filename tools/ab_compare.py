import re,sys
txt=open(sys.argv[1]).read()
parts=txt.split('LIB=')
res={}
for part in parts[1:]:
    e=part.split()[0]
    for l in part.splitlines():
        m=re.match(r'(gemm|conv) (.+?)\s+([\d.]+) GF \| v-1:\s+([\d.]+)us',l)
        if m: res.setdefault(m.group(1)+' '+m.group(2).strip(),{})[e]=float(m.group(4))
for k,v in res.items():
    if 'old' in v and 'new' in v:
        print(f"{k:40s} old {v['old']:7.1f}  new {v['new']:7.1f}  {100*(v['new']/v['old']-1):+5.1f}%")
