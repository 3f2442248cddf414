#!/usr/bin/env python3
"""Round 4: grid search of the band-count cost model (boxblur_rt.hip ichain_band_rows) against tools/rt_band_sweep.py output (gpurun_out/r4_rt_band_sweep.txt)."""
import re, itertools, math
rows=[]
for l in open('profiles/r04_rt_band_sweep.txt'):
    m=re.match(r'(uint\d+) (\d+)x(\d+) x(\d+) \((\d+), (\d+), (\d+), (\d+)\) colgroups (\d+): (.*) us',l)
    if not m: continue
    dt,w,h,fr,_,_,R,P,cg,rest=m.groups(); w,h,fr,R,P=int(w),int(h),int(fr),int(R),int(P)
    meas={}
    for part in rest.split('|'):
        part=part.strip()
        if part.startswith('per-pass'): meas['pp']=float(part.split()[1])
        elif part.startswith('whole'): meas[1]=float(part.split()[1])
        else:
            k,v=part.split(':'); meas[int(k)]=float(v)
    planes=[(w,h),(w//2,h//2),(w//2,h//2)]*fr
    rows.append((dt,planes,R,P,meas))
def model(planes,R,P,nb,a,w0,b,c,endc):
    warm=P*(2*R+1); drain=P*(R+1)
    lds=(P*(2*R+3)+40)*128
    cap=256*min(16,math.floor(160*1024/lds))
    maxh=max(h for w,h in planes)
    br=-(-maxh//nb)
    total=0;waves=0;longest=0
    for (w,h) in planes:
        ncg=-(-w//64)
        if nb==1:
            t=h+drain+endc*warm
            total+=ncg*t;waves+=ncg;longest=max(longest,t);continue
        nbp=-(-h//br)
        for bi in range(nbp):
            t=min(br,h-bi*br)+warm+drain
            total+=ncg*t;longest=max(longest,t)
        waves+=ncg*nbp
    conc=min(waves,cap); wps=conc/1024
    slow=1+a*max(0,wps-w0)
    T=max(longest,total/conc)*slow
    if waves>conc: T+=b*longest*slow
    if nb>1: T+=c*drain
    return T
best=None
for a,w0,b,c,endc in itertools.product([0.2,0.3,0.36,0.45,0.55],[0.75,1.0,1.25,1.5],[0,0.15,0.3,0.5],[1.0,2.0,3.0],[1.0,2.0,3.0]):
    loss=0
    for dt,planes,R,P,meas in rows:
        cands=[k for k in meas if k!='pp']
        pred={k:model(planes,R,P,k,a,w0,b,c,endc) for k in cands}
        pick=min(pred,key=pred.get)
        loss+=meas[pick]/min(meas[k] for k in cands)-1
    if best is None or loss<best[0]: best=(loss,a,w0,b,c,endc)
print(best)
loss,a,w0,b,c,endc=best
for dt,planes,R,P,meas in rows:
    cands=[k for k in meas if k!='pp']
    pred={k:model(planes,R,P,k,a,w0,b,c,endc) for k in cands}
    pick=min(pred,key=pred.get); bk=min(cands,key=lambda k:meas[k])
    print(dt,len(planes)//3,R,P,'pick',pick,meas[pick],'best',bk,meas[bk],'pp',meas['pp'])
