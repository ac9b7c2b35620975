"""The evidence pipeline (no GPU): the committed counter summaries must follow from the committed raw passes, and bench.py must sort kernel names into the
families the roofline tables are built from."""
import importlib.util
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('gc_bench_module', os.path.join(REPO, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ['bench.py']
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_kernel_names_reach_their_families():
    fam = _bench().family_of
    cases = {
        'void (anonymous namespace)::conv_bf16x3_ws_kernel<3, 2, 1, 0, false>(gcconv::Bf16Args)': 'conv_s1',
        'void (anonymous namespace)::conv_bf16x3_kernel<1, 4, 2, 1, 1, 2, 3>(gcconv::Bf16Args)': 'conv_s2',
        'void (anonymous namespace)::conv_s2ws_bf16x3_kernel<2, 0, false>(gcconv::Bf16Args)': 'conv_s2',      # round 6: its template arguments carry no geometry
        'conv_s2ws_bf16x3_kernel<1>|up1,down2,k3': 'conv_s2',
        'void (anonymous namespace)::convt_fused_bf16x3_kernel<2, 2, 2, 32, 0, false>(gcconv::Bf16Args)': 'convt',
        '(anonymous namespace)::convt_edge_bf16x3_kernel(gcconv::Bf16Args)': 'convt',
        '(anonymous namespace)::wgrad_bf16x3_ws2_kernel((anonymous namespace)::WgArgs, int)': 'wgrad',
        'void (anonymous namespace)::fir44_tile_kernel<true, 2, false>(float const*, float const*, float*, int)': 'fir',
        '(anonymous namespace)::plane_dot_kernel(float const*, float const*, float*, long, int, long)': 'bias_act',
        'void (anonymous namespace)::pw_narrow_kernel<true>((anonymous namespace)::PwArgs)': 'pointwise',
        'void at::native::vectorized_elementwise_kernel<4, at::native::CUDAFunctor_add<float>, std::array<char*, 3ul> >': 'aten',
        '(anonymous namespace)::splitk_finish_kernel(gcconv::ConvArgs, int, long long)': 'conv_s1',
    }
    for name, want in cases.items():
        assert fam(name) == want, (name, fam(name), want)


def test_mfma_counter_summary_follows_from_the_committed_passes(tmp_path):
    """profiles/pmc_r06_mfma.json (what bench.py quotes as roofline.mfma_busy) == tools/pmc_mfma.py --parse over profiles/pmc_r06_mfma/pass*.csv."""
    raw = os.path.join(REPO, 'profiles', 'pmc_r06_mfma')
    committed = json.load(open(os.path.join(REPO, 'profiles', 'pmc_r06_mfma.json')))
    out_md, out_json = str(tmp_path / 'a.md'), str(tmp_path / 'a.json')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'pmc_mfma.py'), '--parse', raw, out_md, out_json], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    again = json.load(open(out_json))
    a = {k['name_substring']: k for k in committed['kernels']}
    b = {k['name_substring']: k for k in again['kernels']}
    assert a.keys() == b.keys() and len(a) >= 8
    for name in a:
        for key in ('mfma_busy', 'cycles_per_xcd', 'valu_per_mfma', 'wait_any_frac'):
            assert abs(a[name][key] - b[name][key]) <= 1e-9 * max(1.0, abs(a[name][key])), (name, key, a[name][key], b[name][key])
    dom = a['conv_bf16x3_ws_kernel<3, 2, 1,']
    assert 0.5 < dom['mfma_busy'] < 1.0          # the dominant kernel: matrix pipes busy most of the launch (0.75 measured)
