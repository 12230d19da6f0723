# calibrates rocprofv3's cycle counters against the in-kernel shader clock (s_memtime / s_memrealtime) on the SAME kernel:
# tile_lab variant 0 (the shipped 128x128 two-plane loop), 36-plane resblock shape and the fills-only variant 3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05clock
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in 0 3; do
LAB_SHAPE=1 LAB_ONLY=$V timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc$V -o lab -- $R/tools/tile_lab 10 > $O/run$V.txt 2>&1
cat $O/run$V.txt | grep "\[ "
done
cd $R
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r05clock'
for V in (0,3):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(f'{O}/pmc{V}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:40]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    dur=collections.defaultdict(list)
    for f in glob.glob(f'{O}/pmc{V}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r['Kernel_Name'][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-3)
    for k in agg:
        if 'gemm_w' not in k: continue
        d=n[(k,'GRBM_GUI_ACTIVE')]; us=sum(dur[k])/max(1,len(dur[k]))
        print('variant',V,k,'dispatches',d,'avg us %.1f'%us, {c: '%.4g'%(v/d) for c,v in agg[k].items()})
        print('   GRBM/8/us = %.3f GHz   SQ_BUSY/32/us = %.3f GHz'%(agg[k]['GRBM_GUI_ACTIVE']/d/8/us*1e-3, agg[k]['SQ_BUSY_CYCLES']/d/32/us*1e-3))
PY
