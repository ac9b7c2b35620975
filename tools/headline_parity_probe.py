#!/usr/bin/env python3
"""Dev tool (GPU): every error tests/step_checks.py::check_isolated looks at, for the bench workload's fixture (1024 x 1024, 4 images,
tests/golden/step_1024_b4.npz) with the trainer built exactly as bench.py builds it, per arithmetic mode -- the source of the tolerances of
tests/test_ops_gpu.py::test_headline_iteration_against_the_reference (asserted at ~2 x the values measured here)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'gan-control_amd'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import step_checks  # noqa: E402
from gan_control_amd.models.op import _backend  # noqa: E402
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config  # noqa: E402

write = '--write' in sys.argv
names = [a for a in sys.argv[1:] if a != '--write'] or ['step_1024_b4']
out = {}
for name in names:
    for mode in os.environ.get('PROBE_MODES', 'f32,bf16x3').split(','):
        _backend.get().conv_mode = mode
        m = step_checks._Measure()
        step_checks.check_isolated('cuda', name=name, measure=m, trainer=lambda size, batch: GeneratorTrainer(default_config(size, batch), device='cuda', seed=0))
        out[f'{name}/{mode}'] = {k: float('%.3e' % v) for k, v in m.items() if not k.startswith('_')}
        torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
if write:      # merge into the committed ratchet table (tests/step_checks.py::check_ratchet)
    path = os.path.join(REPO, 'tests', 'golden', 'parity_measured.json')
    table = json.load(open(path)) if os.path.exists(path) else {}
    table.update(out)
    json.dump(table, open(path, 'w'), indent=1, sort_keys=True)
    print('wrote', path)
