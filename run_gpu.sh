cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv2d or network" 2>&1 | tail -3
for v in db nodb db nodb; do
  if [ $v = db ]; then unset GC_NO_DB; else export GC_NO_DB=1; fi
  echo == $v
  python tools/kbench.py --mode bf16x3 --reps 20 --only "conv3x3 s1" 2>&1 | grep -v "wgrad" | grep "@32 \|@64 \|@128\|@256"
done
