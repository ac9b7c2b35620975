#!/usr/bin/env python3
"""Train the G+D loop on a procedural multi-modal image set and report whether it trains (tests/train_sanity.py has the method and its
reasons): three seeds each in exact fp32 and split-bf16 on the GPU plus one run of the CPU oracle's own loop, Frechet distance in a fixed
random-feature space against held-out images, windowed loss curves -> profiles/train_sanity_r05.json.

    python tools/train_sanity.py [--iters 2500] [--size 32] [--batch 16] [--seeds 0 1 2] [--oracle-seconds 240] [--out profiles/train_sanity_r05.json]
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'gan-control_amd'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=2500)
    ap.add_argument('--size', type=int, default=32)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--seeds', type=int, nargs='+', default=[0, 1, 2])
    ap.add_argument('--modes', nargs='+', default=['f32', 'bf16x3'])
    ap.add_argument('--n-eval', type=int, default=2000)
    ap.add_argument('--oracle-seconds', type=float, default=240)
    ap.add_argument('--out', default=os.path.join(REPO, 'profiles', 'train_sanity_r05.json'))
    a = ap.parse_args()
    import train_sanity
    rep = train_sanity.main(size=a.size, batch=a.batch, iters=a.iters, seeds=tuple(a.seeds), modes=tuple(a.modes), n_eval=a.n_eval,
                            oracle_seconds=a.oracle_seconds, out=a.out)
    print(json.dumps({k: rep[k] for k in ('real_vs_real', 'final_distance', 'bf16x3_vs_f32', 'oracle_vs_hip_first_iterations', 'criteria') if k in rep}, indent=1))


if __name__ == '__main__':
    main()
