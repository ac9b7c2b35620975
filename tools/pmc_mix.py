#!/usr/bin/env python3
"""Launch the shape mix of the bench's dominant convolution kernel (round 2: conv_bf16x3_ws_kernel<3>; round 1: conv_bf16x3_kernel<1,4,2,2>,
3x3 stride 1), two launches per shape in a fixed order -- the target of the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.  `--parse <fetch dir> <write dir> <out.json>`
pairs the kernel's dispatches with MIX in order, keeps the second launch of each shape and writes the launch-weighted traffic.

MIX = (batch, channels, resolution, launches per 16 training iterations), from tools/shape_profile.py on the FFHQ-1024 step.
"""
import csv, glob, json, os, sys
MIX = [(4, 128, 256, 84), (4, 256, 128, 84), (4, 512, 64, 84), (8, 128, 256, 32), (8, 256, 128, 32), (8, 512, 64, 32), (2, 128, 256, 16), (2, 256, 128, 16)]
# (the 64-channel layers of round 1's mix run on the wide-tile instantiation <3,2,2> since round 2: a kernel of its own)
KERNEL = os.environ.get('PMC_KERNEL', 'conv_bf16x3_ws_kernel<3, 2, 1,')                 # substring of the rocprofv3 kernel name (round 4: a fourth template argument, the epilogue kind)
KERNEL_LABEL = os.environ.get('PMC_KERNEL_LABEL', 'conv_bf16x3_ws_kernel<3,2,1>|up1,down1,k3')     # the name bench.py reports


def run():
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
    import torch
    from gan_control_amd.models.op import _backend
    from gan_control_amd.models.op._backend import ConvGeom
    be = _backend.get()
    be.conv_mode = 'bf16x3'
    for b, c, res, _ in MIX:
        g = ConvGeom(3, 3, 1, 1, 1, 1, res, res)
        x = torch.randn(b, c, res, res, device='cuda'); w = torch.randn(3, 3, c, c, device='cuda')
        for _ in range(2):
            be.conv2d(x, w, None, None, g)
        torch.cuda.synchronize()
        del x, w
    print('done')


def per_dispatch(directory, counter):
    vals = {}
    for f in glob.glob(directory + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and KERNEL in r['Kernel_Name']:
                vals[int(r['Dispatch_Id'])] = vals.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
    return [vals[k] for k in sorted(vals)]


def parse(fdir, wdir, out):
    f, w = per_dispatch(fdir, 'FETCH_SIZE'), per_dispatch(wdir, 'WRITE_SIZE')
    assert len(f) == len(w) == 2 * len(MIX), (len(f), len(w))
    rows, tot, n = [], 0.0, 0
    for i, (b, c, res, launches) in enumerate(MIX):
        read_b = 2 * f[2 * i + 1] * 1024           # 16-byte-per-lane streaming loads: FETCH_SIZE tallies 64 of every 128 bytes (MI355X_MICROARCH.md, HBM)
        write_b = w[2 * i + 1] * 1024
        algo = 4.0 * b * res * res * (c + c) + 4.0 * 9 * c * c
        rows.append({'batch': b, 'channels': c, 'resolution': res, 'launches_per_16_iterations': launches, 'read_bytes': read_b, 'write_bytes': write_b,
                     'algorithmic_bytes': algo, 'traffic_over_algorithmic': (read_b + write_b) / algo})
        tot += launches * (read_b + write_b); n += launches
        print(f'B{b} {c:4d}ch @{res:4d}: read {read_b / 2**20:8.1f} MiB  write {write_b / 2**20:8.1f} MiB  algorithmic {algo / 2**20:8.1f} MiB  x{(read_b + write_b) / algo:5.2f}')
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gan-control_amd'))
    from gan_control_amd import _lib
    algo_w = sum(r['launches_per_16_iterations'] * r['algorithmic_bytes'] for r in rows) / n
    # source_hash: the kernel sources the measured library was built from -- bench.py quotes this file only for a library built from the same
    json.dump({'kernel': KERNEL_LABEL, 'traffic_bytes_per_launch': tot / n, 'algorithmic_bytes_per_launch': algo_w, 'source_hash': _lib.source_hash(),
               'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over tools/pmc_mix.py; FETCH_SIZE doubled (16-byte-per-lane loads); '
                         'fabric-side bytes (Infinity-Cache hits included), launch-weighted over the shape mix of the FFHQ-1024 step', 'shapes': rows}, open(out, 'w'), indent=1)
    print('launch-weighted traffic per launch: %.1f MiB' % (tot / n / 2**20))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--parse':
        parse(*sys.argv[2:5])
    else:
        run()
