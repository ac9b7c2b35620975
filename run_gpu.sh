cd $GRAFT_REPO_ROOT
timeout 600 python tools/train_sanity.py --size 64 --batch 8 --iters 150 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python tools/train_sanity.py --size 128 --batch 4 --iters 40 --ada 2>&1 | grep -v amdgpu.ids | tail -3
