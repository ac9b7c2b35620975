"""bench.py --gpus N launches its own one-process-per-GPU job (no GPU needed for the plumbing: gloo + the emulated C ABI)."""
import json
import os
import subprocess
import sys

from conftest import REPO

ENTRY = os.path.join(REPO, 'tests', 'bench_emulated_entry.py')
ARGS = ['--steps', '2', '--warmup', '1', '--size', '16', '--batch-per-gpu', '4', '--no-cpu-baseline']


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '2'
    return env


def test_gpus_2_self_launches_two_ranks():
    """The parent never joins the job: it starts torch.distributed.run as a child, relays ONE JSON line and the exit code."""
    out = subprocess.run([sys.executable, ENTRY, '--gpus', '2'] + ARGS, env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['rccl']['ranks'] == 2 and line['rccl']['backend'] == 'gloo'
    assert line['config']['global_batch'] == 8 and line['config']['parallelism'] == 'dp2' and line['scaling'] == 'weak'
    assert line['steps'] == 2 and line['warmup'] == 1 and line['value'] > 0
    assert abs(line['value'] - 2 * 8 / (line['ms_per_step'] * 2e-3)) < 1e-6 * line['value']          # whole-job images / max-over-ranks time
    assert line['phases_fired'] == {'d': 2, 'g': 2, 'r1': 0, 'pl': 0, 'cadence': {'r1_every': 16, 'pl_every': 4}}      # iterations 1, 2


def test_gpus_8_weak_scaling_line():
    """BASELINE config 3's partitioning through the driver's own command shape: 8 ranks x 4 images = global batch 32, one JSON line from rank 0,
    value = the whole job's images over the slowest rank's time."""
    env = _env()
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run([sys.executable, ENTRY, '--gpus', '8', '--steps', '1', '--warmup', '0', '--size', '16', '--batch-per-gpu', '4', '--no-cpu-baseline'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['rccl']['ranks'] == 8
    assert line['config']['global_batch'] == 32 and line['config']['parallelism'] == 'dp8' and line['scaling'] == 'weak'
    assert abs(line['value'] - 32 / (line['ms_per_step'] * 1e-3)) < 1e-6 * line['value']
    assert line['comm']['bytes_per_step'] > 0


def test_child_failure_reaches_the_caller():
    """A failing rank makes the launcher exit non-zero (no JSON line, no silent success)."""
    out = subprocess.run([sys.executable, ENTRY, '--gpus', '2', '--size', '24'] + ARGS[:4] + ['--batch-per-gpu', '4', '--no-cpu-baseline'],
                         env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]


def test_world_size_mismatch_is_refused():
    env = _env()
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, ENTRY, '--gpus', '2'] + ARGS, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and 'WORLD_SIZE=1' in (out.stderr + out.stdout)


def test_bench_refuses_to_run_without_a_gpu_or_the_test_switch():
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '1', '--warmup', '0', '--no-cpu-baseline'],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and 'no CPU fallback' in (out.stderr + out.stdout)


def test_more_ranks_than_gpus_is_refused_quickly():
    """`bench.py --gpus 2` on a machine with fewer GPUs exits non-zero with a readable message instead of starting ranks
    that die at set_device while the others wait at the rendezvous (here: no GPU at all; the -m gpu twin runs on the one-GPU box)."""
    import time
    import torch
    want = torch.cuda.device_count() + 1
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(max(2, want)), '--steps', '1', '--warmup', '0', '--no-cpu-baseline'],
                         env=_env(), capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and time.time() - t0 < 60
    assert 'GPU(s) visible' in out.stderr and not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]


def test_rank_environment_knobs_and_forced_single_rank_ddp():
    """--bucket-mb / --last-bucket-mb / --rccl-algo reach the ranks as environment (and show in the line); --force-ddp runs ONE rank through
    the launcher with the gradient reducer on: the line then carries `comm` like an N > 1 line."""
    out = subprocess.run([sys.executable, ENTRY, '--gpus', '2', '--bucket-mb', '0.25', '--last-bucket-mb', '0.01', '--rccl-algo', 'Ring'] + ARGS,
                         env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    assert line['rccl']['knobs'] == {'NCCL_ALGO': 'Ring', 'NCCL_PROTO': None, 'GANCONTROL_BUCKET_MB': '0.25', 'GANCONTROL_LAST_BUCKET_MB': '0.01'}
    assert line['rccl']['buckets']['bucket_bytes'] == 262144 and line['rccl']['buckets']['last_bucket_bytes'] == 10485
    assert line['rccl']['buckets']['generator'] > 8
    out = subprocess.run([sys.executable, ENTRY, '--force-ddp'] + ARGS, env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    assert line['n_gpus'] == 1 and line['rccl']['ranks'] == 1 and line['comm']['bytes_per_step'] > 0


import pytest  # noqa: E402


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_exits_non_zero_within_a_minute():
    """The -m gpu twin of test_more_ranks_than_gpus_is_refused_quickly: on the one-GPU box `bench.py --gpus 2` must not hang."""
    import time
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('this box has the GPUs the command asks for')
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline'],
                         env=_env(), capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and time.time() - t0 < 60
    assert '1 GPU(s) visible' in out.stderr
