import re,sys
d=sys.argv[1]; names=sys.argv[2:]
def tab(t):
    r={}
    for l in open("%s/%s.txt"%(d,t)):
        m=re.match(r"\s*(-?\d+) (\S.*?)\s+n=\s*\d+\s+([\d.]+) us/fwd",l)
        if m: r[(int(m.group(1)),m.group(2)[:26])]=float(m.group(3))
    return r
tabs={n:tab(n) for n in names}
base=tabs[names[0]]
for k in sorted(base):
    vals=[tabs[n].get(k,-1) for n in names]
    if max(vals)-min(vals)>float(__import__("os").environ.get("THR","2.0")): print(k, " ".join("%7.1f"%v for v in vals))
print("total", " ".join("%7.1f"%sum(tabs[n].values()) for n in names))
