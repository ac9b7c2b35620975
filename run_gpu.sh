cd $GRAFT_REPO_ROOT
timeout 600 python tools/train_sanity.py --size 64 --batch 8 --iters 150 2>&1 | grep -v amdgpu.ids | tail -9
timeout 600 python tools/train_sanity.py --size 64 --batch 8 --iters 60 --ada --precision f32 2>&1 | grep -v amdgpu.ids | tail -4
