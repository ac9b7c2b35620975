#!/bin/bash
# Round 6, session B: whole-step A/B of the stride-2 ws kernel (same box, three alternations), then the full GPU suite.
bash tools/step_ab.sh r06_b new nos2ws > gpurun_out/step_ab_r06_b.log 2>&1
cat gpurun_out/step_ab_r06_b.log
timeout 1200 python -m pytest tests -q -x -m gpu > gpurun_out/gputests_r06_b.log 2>&1; echo "rc=$?" >> gpurun_out/gputests_r06_b.log
tail -5 gpurun_out/gputests_r06_b.log
