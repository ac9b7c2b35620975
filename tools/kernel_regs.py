#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel of a gfx950 code object, from its metadata notes (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include --cuda-device-only -c gan-control_amd/csrc/conv_bf16x3.hip -o /tmp/x.bundle
    python tools/kernel_regs.py /tmp/x.bundle [--all]        # default: only kernels that spill or use scratch
"""
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin/'


def main():
    path, show_all = sys.argv[1], '--all' in sys.argv
    dev = tempfile.mktemp(suffix='.co')
    r = subprocess.run([LLVM + 'clang-offload-bundler', '--unbundle', '--type=o', '--input=' + path, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + dev],
                       capture_output=True, text=True)
    if r.returncode != 0:
        dev = path                      # already a bare code object
    txt = subprocess.run([LLVM + 'llvm-readelf', '--notes', dev], capture_output=True, text=True, check=True).stdout
    for k in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
        g = lambda key: (re.search(r'\.' + key + r':\s+(\S+)', k) or [None, None])[1]
        name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
        spill = int(g('vgpr_spill_count') or 0) + int(g('sgpr_spill_count') or 0) + int(g('private_segment_fixed_size') or 0)
        if show_all or spill:
            print('%-64s vgpr %3s  vgpr_spill %3s  sgpr %3s  sgpr_spill %3s  scratch %4s B  lds %6s B' % (
                name[:64], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))


if __name__ == '__main__':
    main()
