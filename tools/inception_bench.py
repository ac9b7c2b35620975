#!/usr/bin/env python3
"""Images per second of the FID feature network at the reference's evaluation batch (20 images of 1024 x 1024, tracker.py:322-329), and
the pool3 features of the MFMA route against the direct-convolution route (dev tool, GPU only).  GANCONTROL_INCEPTION_MFMA=0|1."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch  # noqa: E402
from gan_control_amd.fid_utils import inception as pinc  # noqa: E402
from oracle import inception as oinc  # noqa: E402

net = pinc.InceptionV3(output_blocks=[3], normalize_input=False)
net.load_state_dict(oinc.procedural_inception_fill_(net.state_dict()))
net = net.cuda().eval()
x = torch.rand(20, 3, 1024, 1024, device='cuda') * 2 - 1
res = {}
for mfma in (False, True):
    pinc._MFMA = mfma
    with torch.no_grad():
        f = net(x)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            f = net(x)[0]
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    res[mfma] = f.reshape(20, -1).double().cpu()
    print('%s: %.1f images/s (%.1f ms per batch of 20; 50 000 images in %.0f s)' % ('MFMA for 1x1 / 3x3' if mfma else 'direct kernel only ', 20 / dt, dt * 1e3, 50000 / (20 / dt)))
print('pool3 features, MFMA route vs direct route: max rel err %.2e' % float((res[True] - res[False]).abs().max() / res[False].abs().max()))
