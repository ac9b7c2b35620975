#!/usr/bin/env python3
"""MFMA-utilisation counters of the matrix kernels that SHIP, each at the FFHQ-1024 step's largest shape (north_star: "rocprof showing ...
MFMA utilisation on the modulated conv against gfx950 peak").

    python3 tools/pmc_mfma.py                      # the target: three launches of each kernel (run under `rocprofv3 --pmc <set> --kernel-trace -- python3 tools/pmc_mfma.py`)
    python3 tools/pmc_mfma.py --parse <dir> <out.md> <out.json>     # <dir>/pass*.csv (tools/pmc_mfma.sh) -> table + json stamped with the library's source hash

Per kernel the LAST of its dispatches is read (the first ones warm the caches), summed over the 8 XCDs.  Units (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is
per-XCD cycles summed over 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over all 1024 SIMDs (= 32 x the 32x32x16 bf16 MFMAs issued); SQ_WAVE_CYCLES /
SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves.
"""
import csv, glob, json, os, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (label, rocprofv3 kernel-name substring, what runs it: kind, B, K, N, res, k, up, down)
KERNELS = [
    ('ws stride-1, 64-ch blocks  B8 256->256 @128^2', 'conv_bf16x3_ws_kernel<3, 2, 1,', ('conv', 8, 256, 256, 128, 3, 1, 1)),
    ('ws stride-1, 64-ch wide    B8  64-> 64 @512^2', 'conv_bf16x3_ws_kernel<3, 2, 2,', ('conv', 8, 64, 64, 512, 3, 1, 1)),
    ('ws stride-1, 32-ch blocks  B8  32-> 32 @1024^2', 'conv_bf16x3_ws_kernel<3, 1, 4,', ('conv', 8, 32, 32, 1024, 3, 1, 1)),
    ('ws stride-2 E/O, 128-ch    B8 128->256 @257^2', 'conv_s2ws_bf16x3_kernel<2,', ('conv', 8, 128, 256, 257, 3, 1, 2)),
    ('ws stride-2 E/O,  64-ch    B8  32-> 64 @1025^2', 'conv_s2ws_bf16x3_kernel<1,', ('conv', 8, 32, 64, 1025, 3, 1, 2)),
    ('transposed fused (main)    B8 512->256 @64^2', 'convt_fused_bf16x3_kernel<2, 2, 2, 32,', ('conv', 8, 512, 256, 64, 3, 2, 1)),
    ('weight gradient ws2        B8 128->128 @256^2', 'wgrad_bf16x3_ws2_kernel', ('wgrad', 8, 128, 128, 256, 3, 1, 1)),
    ('weight gradient stride 2   B8 128->256 @257^2', 'wgrad_bf16x3_s2_kernel', ('wgrad', 8, 128, 256, 257, 3, 1, 2)),
]
SETS = ['GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES',
        'GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS',
        'GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR']


def run():
    sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
    import torch
    from gan_control_amd.models.op import _backend
    from gan_control_amd.models.op._backend import ConvGeom
    be = _backend.get()
    be.conv_mode = 'bf16x3'
    for _, _, (kind, b, K, N, res, k, up, down) in KERNELS:
        if up > 1:
            pad, oh = 2, 2 * res + 1
        else:
            pad = k // 2 if down == 1 else 0
            oh = (res + 2 * pad - k) // down + 1
        g = ConvGeom(k, k, up, down, pad, pad, oh, oh)
        x = torch.randn(b, K, res, res, device='cuda')
        if kind == 'conv':
            w = torch.randn(k, k, K, N, device='cuda')
            for _ in range(3):
                be.conv2d(x, w, None, None, g)
        else:
            dy = torch.randn(b, N, oh, oh, device='cuda')
            for _ in range(3):
                be.conv2d_wgrad(x, dy, None, None, g)
        torch.cuda.synchronize()
    print('done')


def parse(directory, out_md, out_json):
    # counter -> kernel substring -> {dispatch id: summed value}
    vals, passes = {}, {}
    for f in sorted(glob.glob(os.path.join(directory, 'pass*.csv'))):
        for r in csv.DictReader(open(f)):
            for _, sub, _ in KERNELS:
                if sub in r['Kernel_Name']:
                    d = vals.setdefault(r['Counter_Name'], {}).setdefault(sub, {})
                    d[int(r['Dispatch_Id'])] = d.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
                    passes.setdefault(r['Counter_Name'], set()).add(f)
    def last(counter, sub):
        # a counter collected in several passes (GRBM_GUI_ACTIVE rides in every set) is the MEAN over them: the dispatch ids coincide across passes
        d = vals.get(counter, {}).get(sub)
        return d[max(d)] / len(passes[counter]) if d else None
    sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
    from gan_control_amd import _lib
    rows, lines = [], []
    lines.append('| kernel (shape) | launch cycles per XCD | matrix pipes busy | MFMAs | vector instr. per MFMA | wave time parked (s_waitcnt / barrier) | issue-stalled | of that on LDS | LDS array active | bank conflicts of that |')
    lines.append('|---|---|---|---|---|---|---|---|---|---|')
    for label, sub, spec in KERNELS:
        g = last('GRBM_GUI_ACTIVE', sub)
        if g is None:
            continue
        cyc = g / 8.0
        mb, nm, nv = last('SQ_VALU_MFMA_BUSY_CYCLES', sub), last('SQ_INSTS_MFMA', sub), last('SQ_INSTS_VALU', sub)
        wc, wa, wi, wl = last('SQ_WAVE_CYCLES', sub), last('SQ_WAIT_ANY', sub), last('SQ_WAIT_INST_ANY', sub), last('SQ_WAIT_INST_LDS', sub)
        la, lc = last('SQ_LDS_IDX_ACTIVE', sub), last('SQ_LDS_BANK_CONFLICT', sub)
        frac = lambda a, b: None if a is None or not b else a / b
        busy = frac(mb, 1024.0 * cyc)
        row = {'kernel': label.strip(), 'name_substring': sub, 'shape': spec, 'cycles_per_xcd': cyc, 'mfma_busy': busy, 'insts_mfma': nm,
               'valu_per_mfma': frac((nv or 0) - (nm or 0), nm), 'wait_any_frac': frac(wa, wc), 'wait_inst_any_frac': frac(wi, wc), 'wait_inst_lds_of_stalled': frac(wl, wi),
               'lds_active_frac': frac(la, 256.0 * cyc), 'lds_conflict_frac': frac(lc, la)}
        rows.append(row)
        pct = lambda v: 'n/a' if v is None else '%.0f %%' % (100 * v)
        lines.append('| %s | %.3g | **%s** | %s | %s | %s | %s | %s | %s | %s |' % (
            label.strip(), cyc, pct(busy), 'n/a' if nm is None else '%.3g' % nm, 'n/a' if row['valu_per_mfma'] is None else '%.2f' % row['valu_per_mfma'],
            pct(row['wait_any_frac']), pct(row['wait_inst_any_frac']), pct(row['wait_inst_lds_of_stalled']), pct(row['lds_active_frac']), pct(row['lds_conflict_frac'])))
    h = _lib.source_hash()
    open(out_md, 'w').write(
        '# MFMA-utilisation counters of the shipped matrix kernels (round 6; MI355X, rocprofv3 --pmc, one counter set per pass with --kernel-trace only)\n\n'
        'Target: `python3 tools/pmc_mfma.py` (three launches per kernel, the last one read; split-bf16 mode); passes: `tools/pmc_mfma.sh`; library source hash `%s`.\n'
        '"matrix pipes busy" = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the share of the launch in which a SIMD\'s matrix pipe executes an MFMA '
        '(three bf16 MFMAs per product in this arithmetic: 100 %% busy = 833 TF/s algorithmic at the 2.4 GHz peak clock, proportionally less at the clock the launch actually ran at: cycles per XCD / kernel time).  Wave-time shares are of SQ_WAVE_CYCLES '
        '(all waves: in the wave-specialised kernels that includes the staging waves, which wait by design).\n\n' % h + '\n'.join(lines) + '\n')
    json.dump({'source_hash': h, 'source': 'rocprofv3 --pmc (tools/pmc_mfma.sh), last dispatch per kernel, summed over XCDs', 'kernels': rows}, open(out_json, 'w'), indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--parse':
        parse(*sys.argv[2:5])
    elif len(sys.argv) > 1 and sys.argv[1] == '--sets':
        print('\n'.join(SETS))
    else:
        run()
