#!/bin/bash
# usage: profiles/r05_render_timeline.sh <out.txt>  - kernel timeline of one rendered queued frame (rocprofv3 serialises the queues: it
# shows what a rendered frame is made of, not what overlaps).  The KFH_RENDER_OVERLAP switch it compares existed only during the experiment.
export TMPDIR=/tmp
{
for ov in 0 1; do
  d=/tmp/rt_$$_$ov
  echo "== KFH_RENDER_OVERLAP=$ov"
  KFH_RENDER_OVERLAP=$ov rocprofv3 --kernel-trace --output-format csv -d $d -- python3 profiles/r05_render_timeline.py 2> $d.err | grep frames/s
  python3 profiles/timeline.py $d 60
  rm -rf $d $d.err
done
} > $1 2>&1
