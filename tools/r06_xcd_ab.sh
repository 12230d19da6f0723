#!/bin/bash
# XCD-aware block orders of round 6 (norm_act.hip in_fused_group, wino.hip wino_xcd_block) against the plain orders: bit-equality (weight-arena
# hashes of two training steps), same-box step time, per-kernel averages in order on one stream.  usage: tools/r06_xcd_ab.sh SWITCH kernel-substring...
R=$GRAFT_REPO_ROOT; SW=${1:-SWN_WINO_XCD}; shift; O=$R/gpurun_out/r06_xcd_$SW.txt; cd $R
(tools/_bin/native_ab 32 256 2 0 hash | grep "^hash after"; env $SW=0 tools/_bin/native_ab 32 256 2 0 hash | grep "^hash after") > $O 2>&1
tools/_bin/native_ab 32 256 20 3 ab "$SW=0" 2>&1 | grep "ab mean" >> $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/xp$v
  env $SW=$v SWN_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xp$v -o t -- $R/tools/_bin/native_ab 32 256 6 0 bench > /dev/null 2>&1
  echo "== $SW=$v (in order on one stream: kernel, calls, average ns)" >> $O
  python - "$v" "$@" <<'PY' >> $O
import csv,glob,sys
v=sys.argv[1]; pats=sys.argv[2:]
for f in glob.glob('/tmp/xp%s/**/*kernel_stats.csv'%v, recursive=True):
    for r in csv.DictReader(open(f)):
        if any(p in r['Name'] for p in pats):
            print(r['Name'].split('(')[0][-48:], r['Calls'], round(float(r['AverageNs'])))
PY
done
cat $O
