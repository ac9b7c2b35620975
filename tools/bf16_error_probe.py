#!/usr/bin/env python3
"""Measure what `--precision bf16` (one bf16 MFMA per product) costs against the reference's fixtures at 512 x 512 (dev tool, GPU only).

Prints -- and writes to gpurun_out/bf16_errors.json -- the network-forward errors (tests/op_checks.py::network_errors) and every error
the step check looks at (tests/step_checks.py::check_step in measure mode) for the modes given on the command line.  The bounds in
tests/test_baseline_configs.py (BF16_NETWORK_512, BF16_STEP_512) are 2 x the bf16 column of this tool's output.
usage: bf16_error_probe.py [bf16 bf16x3 f32]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'gan-control_amd'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from gan_control_amd.models.op import _backend  # noqa: E402
import op_checks as oc  # noqa: E402
import step_checks  # noqa: E402

out = {}
for mode in (sys.argv[1:] or ['bf16']):
    _backend.get().conv_mode = mode
    net = oc.network_errors(512, 'cuda')
    m = step_checks._Measure()
    step_checks.check_step('cuda', name='step_512', measure=m)
    m.pop('_bounds', None)
    out[mode] = {'network_512': net, 'step_512': dict(m)}
    print(mode, json.dumps(out[mode], indent=1))
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(REPO, 'gpurun_out', 'bf16_errors.json'), 'w'), indent=1)
