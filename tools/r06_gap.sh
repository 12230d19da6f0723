#!/bin/bash
# round 6: what the main queue waits for at the top of the forward pass (the 2.2 ms gap of profiles/timeline_eager_r06.txt)
#   variants: default | everything at the top of the pass (SWN_PREFETCH=3, the order before this round) | the context on its own stream | the C++ host
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_gap; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --steps 4 --warmup 3"
one() {   # tag, env, command...
  tag=$1; shift; e=$1; shift
  env $e rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- "$@" > $O/$tag.log 2>&1
  (cd $R && python tools/timeline.py $O/$tag --dump-gap > $O/timeline_$tag.txt 2>&1)
  rm -rf $O/$tag
  echo "== $tag"; grep -a -o '"ms_per_step": [0-9.]*' $O/$tag.log | head -1; grep -a "^bench" $O/$tag.log | head -1
  grep -n "^gap\|step window\|busy:" $O/timeline_$tag.txt | cut -c1-260
}
one default X=1 $B
one at_top SWN_PREFETCH=3 $B
one ownstream SWAPNET_OWN_STREAM=1 $B
one native X=1 $R/tools/_bin/native_ab 32 256 6 0 bench
# un-profiled step times of the same variants
cd $R
for v in X=1 SWN_PREFETCH=3 SWAPNET_OWN_STREAM=1 SWN_PREFETCH=0; do env $v python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$v', d['ms_per_step'], d['value'])"; done
