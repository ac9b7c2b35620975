#!/bin/bash
# Round 6, session R: 32k x 128n stride-2 weight-gradient tiles: whole-step A/B (three alternations), full GPU suite, race screen
O=gpurun_out/r06_r; mkdir -p $O
bash tools/step_ab.sh r06_r new non128 > $O/step_ab.log 2>&1; cat $O/step_ab.log | tail -8
timeout 1800 python -m pytest tests -q -x -m gpu > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log; tail -3 $O/gputests.log
timeout 900 python tools/race_screen.py 100 2>&1 | grep -v amdgpu > $O/race_screen.log; tail -1 $O/race_screen.log
