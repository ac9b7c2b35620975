"""-m gpu: a SHORT version of tools/train_sanity.py -- does the G+D loop move generated samples towards the data, in both arithmetics?

The full run (2 500 iterations x 3 seeds x 2 arithmetics + the CPU oracle's own loop) is profiles/train_sanity_r05.json; this test keeps one seed
per arithmetic and SHORT_ITERS iterations so that it stays ~1 minute of the GPU suite.  Bounds are set from the full run's curves at that
iteration count with a wide margin (trajectories are seed-dependent; the test is about "trains / collapses / diverges", not about a number).
"""
import pytest
import torch

SHORT_ITERS = 600


@pytest.mark.gpu
def test_training_moves_samples_towards_the_data_in_both_arithmetics():
    import train_sanity as ts
    from gan_control_amd.models.op import _backend
    dev = 'cuda:0'
    hip = _backend.get()
    prev = hip.conv_mode
    try:
        space = ts.FeatureSpace(dev)
        train_set = ts.procedural_images(2048, 32, seed=7, device=dev)
        held_out = ts.procedural_images(1000, 32, seed=8, device=dev)
        real = {'inception': space.stats(held_out), 'pixel': ts.pixel_stats(held_out)}
        floor = ts.pixel_distance(train_set[:1000], real['pixel'])
        runs = {}
        for mode in ('f32', 'bf16x3'):
            runs[mode] = ts.run_hip(space, real, train_set, 32, 16, SHORT_ITERS, 0, mode, 1000)
            print(mode, runs[mode])
    finally:
        hip.conv_mode = prev
        torch.cuda.empty_cache()
    for mode, r in runs.items():
        d0, d1 = r['distance'][KEY]['0'], r['distance'][KEY][str(SHORT_ITERS)]
        assert r['finite'], mode
        assert d1 < d0 / FALL_AT_SHORT_ITERS, (mode, d0, d1, floor)
        # the game is alive: the discriminator's logistic loss neither collapses to 0 nor blows up (2 ln 2 = 1.39 at equilibrium)
        assert all(0.1 < w < 2.5 for w in r['d_logistic_windows'][1:]), (mode, r['d_logistic_windows'])
    a, b = (runs[m]['distance'][KEY][str(SHORT_ITERS)] for m in ('f32', 'bf16x3'))
    assert abs(a - b) <= MODE_GAP * max(a, b), (a, b)        # same data order, seeds and initial weights: the arithmetics end in the same region


# profiles/train_sanity_r05.json, EMA generator, pixel-space distance: 16.5-23.6 at initialisation, 5.2-9.5 after 500 iterations in all six runs
# (2.5-4.3 x), f32 and bf16x3 within 7 % of each other at equal seed
KEY = 'ema/pixel'
FALL_AT_SHORT_ITERS = 1.8
MODE_GAP = 0.25
