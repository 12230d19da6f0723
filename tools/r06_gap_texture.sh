#!/bin/bash
# round 6: main-queue gaps of the texture stage (bench.py --stage texture), traced
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_gap_tex; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python $R/bench.py --stage texture --no-cpu-baseline --no-roofline --steps 4 --warmup 3 > $O/t.log 2>&1
cd $R
python - <<'PY'
import csv, glob, re, collections, os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06_gap_tex'
rows=[]
for f in glob.glob(O+'/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
ad=[i for i,r in enumerate(rows) if 'l1_kernel' in r[3]]
a,b=ad[-2],ad[-1]
win=rows[a:b]; t0=win[0][0]; t1=rows[b][0]
def short(k): return k.split("(")[0].replace("void ","").replace("swn::","").replace("(anonymous namespace)::","")[:46]
perq=collections.defaultdict(float)
for s,e,q,k in win: perq[q]+=(e-s)*1e-6
mainq=max(perq,key=perq.get)
print("texture step window %.3f ms, %d launches, per-queue kernel ms %s"%((t1-t0)*1e-6,len(win),dict(perq)))
ev=[]
for s,e,q,k in win: ev.append((s,1,q)); ev.append((min(e,t1),-1,q))
ev.sort(); act=collections.Counter(); last=t0; hist=collections.Counter()
for t,d,q in ev:
    n=sum(1 for v in act.values() if v>0); hist[n]+=t-last; last=t; act[q]+=d
print({n:round(v*1e-6,3) for n,v in hist.items()})
prev=None; pk=None
for s,e,q,k in win:
    if q!=mainq: continue
    if prev is not None and s-prev>20000:
        other=collections.defaultdict(float)
        for s2,e2,q2,k2 in win:
            if q2!=mainq and e2>prev and s2<s: other[short(k2)]+=(min(e2,s)-max(s2,prev))*1e-6
        print("gap %.3f ms at +%.3f after [%s] before [%s]; other: %s"%((s-prev)*1e-6,(prev-t0)*1e-6,short(pk),short(k),", ".join("%s %.3f"%kv for kv in sorted(other.items(),key=lambda x:-x[1])[:5])))
    if prev is None or e>prev: prev,pk=e,k
PY
rm -rf $O/t
