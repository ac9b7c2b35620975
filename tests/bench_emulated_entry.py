"""Test entry for bench.py's rank plumbing on CPUs (tests/test_bench_launcher.py runs it; never part of the product).

Installs the emulated C ABI of tests/conftest.py, switches bench.py to its CPU / gloo test mode and hands over to bench.main():
``python tests/bench_emulated_entry.py --gpus 2 ...`` therefore exercises the self-launcher (parent starts
``python -m torch.distributed.run --nproc-per-node 2 <this file> ...`` as a child and relays its JSON line), the rank / world-size
handling, the max-over-ranks timing and the JSON contract without a GPU.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, os.path.join(REPO, 'gan-control_amd'), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402


def main():
    if 'WORLD_SIZE' in os.environ:          # a rank of the job (the launching parent needs none of this)
        from conftest import EmulatedBackend
        from gan_control_amd.models.op import _backend
        _backend._install_for_tests(EmulatedBackend())
    bench._TEST_CPU['enabled'] = True
    bench.main(entry=os.path.abspath(__file__))


if __name__ == '__main__':
    main()
