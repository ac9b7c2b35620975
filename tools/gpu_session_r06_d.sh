#!/bin/bash
# Round 6, session D: ATen-side launches of a plain iteration (17), a path-length iteration (20) and a full one (16), by op and input shapes.
O=gpurun_out/r06_d; mkdir -p $O
for it in 17 20 16; do
  timeout 600 python tools/aten_profile.py --iter $it --top 120 > $O/aten_iter$it.log 2>&1
done
head -70 $O/aten_iter17.log
