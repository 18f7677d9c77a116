"""Rank process of the data-parallel GPU tests (started by
tests/test_gpu_parallel.py, never collected by pytest).

  mode dp    : `world` ranks share the visible GPU(s) (WN_SHARE_GPU=1, gloo:
               RCCL refuses two ranks on one device); every rank runs the REAL
               model on its clip shard: net.loss -> optimizer.minimize (flat
               bucket all-reduce, 1/N folded into the update kernel; with
               spec['overlap'] as two calls, the tail from inside the backward
               pass) for `steps` steps and rank 0 saves the parameters.
  mode nccl1 : world_size-1 process group over backend "nccl" (= RCCL):
               broadcast of net.params and all-reduce of net.grads go through
               the RCCL code path on the device bucket.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from util import O, MID, cfg_with, build_pair  # noqa: E402


def main():
    spec = json.loads(sys.argv[1])
    from wavenet import parallel, optimizer_factory
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local % torch.cuda.device_count())
    B, T, steps = spec['B'], spec['T'], spec['steps']
    cfg = cfg_with(MID, batch_size=B // world, **spec.get('cfg', {}))
    net, var = build_pair(cfg)
    rng = np.random.default_rng(17)
    audio = rng.uniform(-1, 1, (steps, B, T)).astype(np.float32)
    ids = rng.integers(0, cfg.get('global_condition_cardinality') or 1,
                       (steps, B))
    gc = cfg.get('global_condition_cardinality') is not None
    if spec['mode'] == 'nccl1':
        import torch.distributed as dist
        dist.init_process_group(backend='nccl', rank=0, world_size=1)
        net.loss(audio[0], ids[0] if gc else None)
        before = net.grads.clone()
        dist.all_reduce(net.grads)
        dist.broadcast(net.params, src=0)
        torch.cuda.synchronize()
        ok = bool(torch.equal(before, net.grads))
        json.dump({'ok': ok, 'backend': dist.get_backend(),
                   'absmax': float(before.abs().max())},
                  open(spec['out'], 'w'))
        dist.destroy_process_group()
        return
    if spec['mode'] == 'nccl1_overlap':
        # the DEFAULT stack (persistent stack launches) at world size 1 over
        # RCCL with the collectives really issued (rehearse_world_one)
        import torch.distributed as dist
        from util import DEFAULT
        from wavenet import WaveNetModel
        from util import model_kwargs, synth_audio
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(backend='nccl', rank=0, world_size=1)
        parallel.rehearse_world_one = True
        assert parallel.is_distributed()
        net = WaveNetModel(seed=0, **model_kwargs(cfg_with(DEFAULT, batch_size=B)))
        net.dp_overlap_allreduce = bool(spec['overlap'])
        opt = optimizer_factory['adam'](learning_rate=1e-3, momentum=0.9)
        a = synth_audio(B, T)
        losses = []
        for s in range(steps):
            loss = net.loss(a)
            assert (net._tail_work is not None) == bool(spec['overlap'])
            opt.minimize(loss)
            assert net._tail_work is None
            losses.append(float(loss))
        torch.cuda.synchronize()
        net.check_device_errors()
        np.savez(spec['out'], params=net.params.cpu().numpy(), losses=np.asarray(losses),
                 comm_stream_used=bool(parallel._comm_streams))
        dist.destroy_process_group()
        return
    if rank == 1:                       # broadcast must repair this
        with torch.no_grad():
            net.params.add_(1.0)
    parallel.broadcast_parameters(net)
    opt = optimizer_factory[spec['opt']](learning_rate=spec['lr'],
                                         momentum=0.9)
    lo, hi = parallel.shard_range(B, rank, world)
    # (spec['overlap']: the tail of the gradient bucket all-reduced from inside
    # the backward pass, as bench.py / train.py run under torch.distributed)
    net.dp_overlap_allreduce = bool(spec.get('overlap', False))
    losses = []
    for s in range(steps):
        loss = net.loss(audio[s, lo:hi], ids[s, lo:hi] if gc else None)
        opt.minimize(loss)
        losses.append(float(parallel.allreduce_mean_scalar(loss)))
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(spec['out'], params=net.params.cpu().numpy(),
                 losses=np.asarray(losses))
    if parallel.is_distributed():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
