#!/bin/bash
O=gpurun_out/r06_f; mkdir -p $O
timeout 600 python tools/hbm_profile.py > $O/hbm_profile.log 2>&1
head -90 $O/hbm_profile.log
