# same-box A/B of library builds on the low-resolution tail: tools/tail_ab.sh <lib> <lib> ...   ("new" = the in-tree library, else csrc/alt/libalt_<name>.so)
out=gpurun_out/tail_ab.log; rm -f $out
for rep in 1 2; do
for b in 2 4 8; do
  for lib in "$@"; do
    if [ $lib = new ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$PWD/gan-control_amd/csrc/alt/libalt_$lib.so; fi
    echo "== $lib B=$b" >> $out
    python tools/kbench.py --mode bf16x3 --batch $b --reps 30 --only "512->512 @" 2>&1 | grep -v "wgrad\|amdgpu.ids\|@64\|@65\|@63" >> $out
  done
done
done
