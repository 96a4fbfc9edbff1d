#!/bin/bash
# sweep the wave-scheduling thresholds of render_kernel_stream (RT06_TUNE=keep,shade,leaf) on config 2 at reduced spp
for t in "$@"; do
  RT06_TUNE=$t python tools/render.py --scene book1_final --width 1200 --height 800 --spp 250 --out /tmp/x.png 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$t', d['render_ms'])"
done
