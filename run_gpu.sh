set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --size 256 --batch-per-gpu 4 --steps 8 --warmup 4 --no-cpu-baseline 2>&1 | tail -3
timeout 1500 python bench.py --steps 16 --warmup 16 2>&1 | tail -3
