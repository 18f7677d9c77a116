"""Data-parallel path on CPU: two `gloo` ranks.  The flat-bucket all-reduce +
1/N scaling of wavenet.parallel must reproduce the single-process gradient of
the full batch ("N ranks x B/N == 1 rank x B").  Per-rank gradients
come from the CPU oracle here (tests may use it as a stand-in producer; on the
GPU the same helpers run on the HIP bucket over RCCL)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import O, TINY, cfg_with, ROOT, PKG


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Bucket(object):
    def __init__(self, flat):
        self.grads = flat
        self.params = flat.clone()


def _worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from wavenet import parallel
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and parallel.is_distributed()
    B, T = 2 * world, 21
    cfg = cfg_with(TINY, batch_size=B // world)
    var = O.create_variables(cfg, seed=0, dtype=np.float64, bias_scale=0.1)
    audio = np.random.default_rng(5).uniform(-1, 1, (B, T)).astype(np.float32)
    lo, hi = parallel.shard_range(B, rank, world)
    assert (lo, hi) == (2 * rank, 2 * rank + 2)
    loss, g = O.loss_and_grads(cfg, var, audio[lo:hi], dtype=np.float64)
    bucket = _Bucket(torch.from_numpy(O.pack(g)))
    # rank 1 starts from garbage weights: broadcast must fix that
    ref_params = torch.from_numpy(O.pack(var))
    bucket.params = ref_params.clone() + (1.0 if rank == 1 else 0.0)
    parallel.broadcast_parameters(bucket, src=0)
    assert torch.equal(bucket.params, ref_params)
    scale = parallel.allreduce_gradients(bucket)
    avg = bucket.grads * scale
    mloss = parallel.allreduce_mean_scalar(torch.tensor(loss))
    # per-step agreement (train.py): rank 1 holds a shorter tail piece, and
    # in the second call its reader "failed"
    t_common, ok = parallel.agree_step(37 if rank == 1 else 100 + rank, True)
    assert (t_common, ok) == (37, True)
    t_common, ok = parallel.agree_step(64, rank == 0)
    assert ok is False
    t_common, ok = parallel.agree_step(1 if rank == 0 else 500, True)
    assert t_common == 1            # every rank skips this step together
    # collective abort decision (train.py): an error only rank 1 saw
    assert parallel.any_rank(rank == 1) is True
    assert parallel.any_rank(False) is False
    np.save(os.path.join(out_dir, 'g%d.npy' % rank), avg.numpy())
    np.save(os.path.join(out_dir, 'l%d.npy' % rank), mloss.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_dp_ranks_equal_full_batch(tmp_path, world):
    """world = 8 is the rank count of BASELINE.json configs[2] / [3]: shard
    ranges, the 1/N of the flat bucket, broadcast, the per-step agreement and
    the abort decision at the size the driver's first 8-GPU run has."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world,
             join=True)
    B, T = 2 * world, 21
    cfg = cfg_with(TINY, batch_size=B)
    var = O.create_variables(cfg, seed=0, dtype=np.float64, bias_scale=0.1)
    audio = np.random.default_rng(5).uniform(-1, 1, (B, T)).astype(np.float32)
    loss, g = O.loss_and_grads(cfg, var, audio, dtype=np.float64)
    full = O.pack(g)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), 'g%d.npy' % r))
        assert np.abs(got - full).max() < 1e-12
        assert abs(float(np.load(os.path.join(str(tmp_path),
                                              'l%d.npy' % r))) - loss) < 1e-12


def _two_call_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from wavenet import parallel
    parallel.init_from_env(backend='gloo')
    # the default stack's bucket layout on a bookkeeping-only model (device
    # 'cpu': names / shapes / segments, no compute), float32 like the device's
    import json
    from wavenet import WaveNetModel
    from util import model_kwargs
    p_ = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
    cfg = {k: p_[k] for k in p_ if k != 'sample_rate'}
    cfg.update(batch_size=1, global_condition_channels=32,
               global_condition_cardinality=377)
    net = WaveNetModel(device='cpu', **model_kwargs(cfg))
    lo = parallel.tail_start(net)
    n = net.grads.numel()
    assert n == 1630432 and n - lo == 1238784
    assert 0.75 < (n - lo) / float(n) < 0.77   # 76 % of the bucket (82 % without conditioning)
    g = torch.from_numpy(np.random.default_rng(100 + rank).standard_normal(n)
                         .astype(np.float32))
    one = _Bucket(g.clone())
    s1 = parallel.allreduce_gradients(one)                 # one call
    net.grads.copy_(g)
    parallel.begin_tail_allreduce(net)                     # inside the backward pass
    # a backward pass no update followed (every rank alike): the dangling
    # collective is joined with a warning, the ranks' sequences stay equal
    with pytest.warns(UserWarning, match='never joined'):
        parallel.begin_tail_allreduce(net)
    assert parallel.abandon_tail_allreduce(net) and net._tail_work is None
    assert not parallel.abandon_tail_allreduce(net)
    net.grads.copy_(g)
    parallel.begin_tail_allreduce(net)
    s2 = parallel.allreduce_gradients(net)                 # head + join
    assert s1 == s2 == 1.0 / world and net._tail_work is None
    np.save(os.path.join(out_dir, 'two%d.npy' % rank),
            np.array([float(torch.equal(one.grads, net.grads)),
                      float((one.grads - net.grads).abs().max()),
                      float(one.grads.abs().max())]))
    dist.destroy_process_group()


def test_two_call_allreduce_equals_one_bucket(tmp_path):
    """The tail of the bucket (skip convs + post-processing) all-reduced from
    inside the backward pass, the head at the update (round 5,
    `dp_overlap_allreduce`) against ONE all-reduce of the whole bucket: 4 gloo
    ranks, the default stack's float32 layout with global conditioning."""
    world = 4
    mp.spawn(_two_call_worker, args=(world, _free_port(), str(tmp_path)),
             nprocs=world, join=True)
    for r in range(world):
        same, diff, scale = np.load(os.path.join(str(tmp_path), 'two%d.npy' % r))
        # gloo's ring adds a chunk's four terms in an order that depends on the
        # chunk's position in the buffer: not necessarily bitwise, always within
        # float32 rounding of a four-term sum
        print('rank %d: bitwise %s, max diff %.3e of %.3e' % (r, bool(same), diff, scale))
        assert same == 1.0 or diff <= 4e-7 * scale, (r, same, diff, scale)


def test_bench_gc_ids_of_eight_ranks():
    """configs[3] at 8 GPUs: rank r owns global clips [8 r, 8 r + 8) with
    speaker id (37 b) mod 377 -- 64 distinct ids, every one a valid row."""
    sys.path.insert(0, ROOT)
    import bench
    ids = [bench.rank_gc_ids(r, 8) for r in range(8)]
    flat = [i for row in ids for i in row]
    assert flat == [(37 * b) % 377 for b in range(64)]
    assert len(set(flat)) == 64 and min(flat) >= 0 and max(flat) < 377
    env = bench.collective_env()
    assert set(env) >= {'rccl_version', 'NCCL_ALGO', 'NCCL_PROTO', 'env'}


def test_world_size_one_is_identity():
    from wavenet import parallel
    t = torch.arange(5, dtype=torch.float32)
    assert parallel.allreduce_flat_(t) == 1.0
    assert torch.equal(t, torch.arange(5, dtype=torch.float32))
    assert not parallel.is_distributed()
    assert parallel.agree_step(123, True) == (123, True)


def test_bench_rejects_rank_count_mismatch():
    """`--gpus 8` with WORLD_SIZE=1 must fail loudly, not benchmark 1 GPU."""
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'),
                        '--gpus', '8', '--steps', '1', '--warmup', '0'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300)
    assert p.returncode != 0
    assert b'WORLD_SIZE 1 != --gpus 8' in p.stderr


def test_bench_self_launch_reports_rank_failure():
    """`python bench.py --gpus 2` run plainly (no WORLD_SIZE) starts the rank
    processes itself; without GPUs every rank must fail loudly and the
    launcher must hand that failure on (non-zero exit), not hang or print a
    one-GPU number."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'WN_SHARE_GPU'):
        env.pop(k, None)
    import torch as _t
    if _t.cuda.is_available():
        pytest.skip('needs a box without GPUs')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'),
                        '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    assert p.returncode != 0
    assert b'has no GPU' in p.stderr
    assert b'"metric"' not in p.stdout
