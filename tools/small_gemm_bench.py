#!/usr/bin/env python3
"""The tiny products of the style path (gc_small_gemm_f32 against torch.addmm), each shape 20 times in a fixed order; run under
`rocprofv3 --kernel-trace --output-format csv` and pass the trace to `--parse` for microseconds per shape (the Python call costs more
than the kernels, so host-side timing says nothing).  GANCONTROL_SMALL_GEMM=0 times the library GEMMs."""
import csv, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
SHAPES = [  # (M, K, N, a transposed, b layout)
    (4, 512, 512, False, 'nk'), (4, 512, 512, False, 'kn'), (512, 4, 512, True, 'kn'),
    (4, 512, 256, False, 'nk'), (4, 256, 512, False, 'kn'), (256, 4, 512, True, 'kn'),
    (4, 512, 64, False, 'nk'), (4, 64, 512, False, 'kn'), (4, 512, 32, False, 'nk'), (4, 32, 32, False, 'nk'),
    (8, 8192, 512, False, 'nk'), (8, 512, 8192, False, 'kn'), (512, 8, 8192, True, 'kn'), (8, 512, 1, False, 'nk'), (2, 512, 512, False, 'nk'),
]
REPS = 20


def run():
    import torch
    from gan_control_amd.models.op.linear import _addmm
    for m, k, n, a_t, bl in SHAPES:
        a = torch.randn(k, m, device='cuda').t() if a_t else torch.randn(m, k, device='cuda')
        b = torch.randn(n, k, device='cuda').t() if bl == 'nk' else torch.randn(k, n, device='cuda')
        torch.cuda.synchronize()
        for _ in range(REPS):
            _addmm(None, a, b, 0.0, 0.5)
        torch.cuda.synchronize()


def parse(path):
    rows = [r for r in csv.DictReader(open(path)) if ('outer_kernel' in r['Kernel_Name'] or 'Cijk' in r['Kernel_Name'])]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    assert len(rows) == REPS * len(SHAPES), len(rows)
    for i, s in enumerate(SHAPES):
        d = sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[i * REPS:(i + 1) * REPS])
        print('%-34s %7.1f us  %s' % (s, d[len(d) // 2] / 1e3, rows[i * REPS]['Kernel_Name'][:60]))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--parse':
        parse(sys.argv[2])
    else:
        run()
