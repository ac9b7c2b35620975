"""Helper of test_ops_gpu.py::test_ddp_buckets_launch_from_hooks: run under torch.distributed.run (one rank, RCCL, the
collective path forced on with GANCONTROL_FORCE_DDP=1).  Runs 17 iterations at 64x64 so that every phase (D step, R1,
G step, path length) occurs at least twice, and prints the reducers' launch reports of the last occurrence as JSON."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'gan-control_amd')):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', torch.cuda.current_device()))
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    _backend.get().conv_mode = 'bf16x3'
    tr = GeneratorTrainer(default_config(64, 4 * world), device='cuda', seed=0)
    real = tr.synthetic_batch()
    first = {}
    for i in range(17):
        tr.train_iteration(i, real)
        if i == 0:
            first = {'d': dict(tr.d_reducer.report), 'g': dict(tr.g_reducer.report)}
    torch.cuda.synchronize()
    views = all(p.grad is None or p.grad.data_ptr() == tr.g_reducer._bucket_of[p].view(p).data_ptr() for p in tr.generator.parameters())
    out = {'first': first, 'last': {'d': tr.d_reducer.report, 'g': tr.g_reducer.report}, 'buckets': {'d': len(tr.d_reducer.buckets), 'g': len(tr.g_reducer.buckets)},
           'grads_are_bucket_views': views, 'losses': tr.reduced_stats()}
    if rank == 0:
        print('PROBE ' + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
