"""Data-parallel training over the GPUs of one node (new capability; the
reference is single-process: train.py:261, no collective anywhere).

One process per GPU (torchrun), `torch.distributed` with backend "nccl"
(= RCCL over xGMI on ROCm).  Clips are independent and the loss is a mean over
B*T rows (wavenet/model.py:666), so with equal per-rank B*T

    g_global = (1/N) * sum_r g_r          (exact, incl. the zero-label row)

The whole gradient lives in ONE flat fp32 bucket (6.06 MB for the default
stack); the 1/N is folded into the optimizer kernel (`grad_scale`).  At 6 MB
the ring moves 2(N-1)/N * 6 MB per GPU (~70 us of wire time at N=8 over
per-link-bound xGMI, plus the collective's launch and the wait for the slowest
rank) against a >= 7 ms step.

Two calls per step (round 5, `model.dp_overlap_allreduce`, what bench.py /
train.py switch on under torch.distributed): the bucket's TAIL -- skip convs
and post-processing, 82 % of the bytes (76 % with global conditioning), laid
out last -- is complete when the
three weight-gradient GEMMs are, i.e. BEFORE the 1.5 ms backward stack launch
starts.  `begin_tail_allreduce` (called from inside the backward pass at that
point) issues its all-reduce on a communication stream behind an event, so it
runs beside the backward stack; `allreduce_gradients` (optimizer.minimize)
then reduces only the head (embedding, causal layer, residual blocks)
and joins.  Off (or L2 regularisation on, which adds lambda * params to the
whole bucket after the backward pass): one all-reduce of the whole bucket
right before the update.  Both ways sum the same N values per element and
every rank ends with the same bits as every other rank; against the one-call
result they agree to one float32 rounding of an N-term sum, not bitwise (a
ring adds a chunk's terms in an order that depends on the chunk's position in
the buffer: 4 gloo ranks, max difference 1 ulp -- tests/test_parallel_gloo.py).

The helpers are device-agnostic on purpose: they only touch flat tensors, so
the sharding / averaging logic is covered on CPU by world_size-2 `gloo` tests.
"""
import os

import torch
import torch.distributed as dist


# Rehearsal switch (tests only): treat an initialised world-size-1 group as
# distributed, so that the collective calls -- incl. the tail all-reduce on the
# communication stream beside the backward stack -- run through RCCL on a box
# with one GPU.  A sum over one rank is the identity: results must not change.
rehearse_world_one = False


_ctl_group = [None]      # gloo group for the control plane when the backend is RCCL


def _ctl(device):
    """(group, device) for a control-plane collective"""
    if _ctl_group[0] is not None:
        return _ctl_group[0], 'cpu'
    return None, device


def is_distributed():
    return dist.is_available() and dist.is_initialized() and \
        (dist.get_world_size() > 1 or rehearse_world_one)


def init_from_env(backend=None, host_control_plane=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (the
    launcher's env).  Returns (rank, world, local_rank).  No-op when
    WORLD_SIZE is absent or 1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # WN_DIST_BACKEND=gloo lets the N>1 path be rehearsed on a box
            # with fewer GPUs than ranks (gloo stages CUDA tensors via host)
            backend = os.environ.get('WN_DIST_BACKEND') or (
                'nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        # a rank that dies must not leave the others in a collective for
        # ever: bounded wait (WN_DIST_TIMEOUT seconds, default 300)
        import datetime
        tmo = datetime.timedelta(
            seconds=float(os.environ.get('WN_DIST_TIMEOUT', '300')))
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=tmo)
        if backend == 'nccl' and host_control_plane:
            # control-plane collectives (agree_step, any_rank: a few integers
            # per step) over a gloo group on the host: on the RCCL group they
            # need a device tensor and a .cpu() behind it, which waits for
            # everything queued on the stream -- the training loop would stop
            # overlapping its host work with the device every step.  Asked for
            # by train.py only: bench.py has no per-step control collective and
            # keeps its first contact with an N-GPU node to ONE process group.
            try:
                _ctl_group[0] = dist.new_group(backend='gloo', timeout=tmo)
            except Exception:           # noqa: BLE001 (no gloo transport: the RCCL group serves)
                _ctl_group[0] = None
    return rank, world, local


def shard_range(global_batch, rank, world):
    """Clips [lo, hi) of the global minibatch owned by `rank` (equal shards;
    the exact-mean identity above needs equal per-rank B*T)."""
    if global_batch % world:
        raise ValueError('global batch %d is not divisible by world size %d'
                         % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


# bench.py's instrumented pass: a list here makes every gradient all-reduce
# append a (start, end) pair of timing events recorded on the compute stream
# (the collective runs on the backend's own stream; the compute stream waits
# for it, so the pair brackets launch + wire time + the wait for the slowest
# rank to arrive)
timing_events = None


def allreduce_flat_(flat):
    """In-place all-reduce(sum) of one flat bucket; returns the scale (1/N)
    the caller must apply (folded into the optimizer kernel)."""
    if not is_distributed():
        return 1.0
    ev = timing_events
    if ev is not None and flat.is_cuda:
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        e.record()
        ev.append((s, e))
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


_comm_streams = {}


def _comm_stream(device):
    """The stream the early (tail) all-reduce is issued from: the collective
    library orders its own stream after the CURRENT stream at the call, so the
    call is made with this one current, after it has waited for the event that
    marks the tail complete -- not for the backward stack behind it."""
    key = (device.type, device.index)
    if key not in _comm_streams:
        _comm_streams[key] = torch.cuda.Stream(device=device)
    return _comm_streams[key]


def tail_start(model):
    """First element of the bucket's tail: skip convs + post-processing
    (wavenet/model.py `segments`: ... [layers][skip_w][skip_b][post1_w]
    [post2_w][post1_b][post2_b]), everything the residual-stack backward does
    not write."""
    return int(model.segments['skip_w'][0])


def begin_tail_allreduce(model):
    """Issue the all-reduce(sum) of the gradient bucket's tail NOW -- the
    caller (the backward pass) guarantees that the tail is complete on the
    current stream and that `allreduce_gradients` follows in this step.
    No-op when not distributed."""
    if not is_distributed():
        return
    if getattr(model, '_tail_work', None) is not None:
        # loss(backward=True) without optimizer.minimize (an evaluation, a
        # gradient check): the previous tail was issued on EVERY rank (the
        # ranks run the same program), so joining it here keeps the ranks'
        # collective sequences equal; its sums went into gradients this pass
        # has already overwritten.
        import warnings
        warnings.warn('the tail all-reduce of the previous backward pass was '
                      'never joined (no optimizer.minimize followed): joined '
                      'and dropped now')
        abandon_tail_allreduce(model)
    lo = tail_start(model)
    tail = model.grads[lo:]
    if tail.is_cuda:
        comm = _comm_stream(tail.device)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(comm):
            comm.wait_event(ready)
            work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
    else:
        work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
    model._tail_work = (work, lo)


def abandon_tail_allreduce(model):
    """Join and forget a tail all-reduce that `allreduce_gradients` will not
    join (an exception later in the same backward pass, a backward pass that
    no update follows).  The collective itself was issued and completes on
    every rank -- only this rank's bookkeeping is reset -- so the next step
    starts from a clean sequence.  Returns True when there was one."""
    pending = getattr(model, '_tail_work', None)
    if pending is None:
        return False
    model._tail_work = None
    try:
        pending[0].wait()
    except Exception as e:           # a failed collective: report, do not mask
        import warnings
        warnings.warn('joining the abandoned tail all-reduce failed: %r' % (e,))
    return True


def allreduce_gradients(model):
    """All-reduce the model's flat gradient bucket -- or, when the backward
    pass has already started the tail's (`begin_tail_allreduce`), the head
    only, then join the tail; returns grad_scale."""
    pending = getattr(model, '_tail_work', None)
    if pending is None:
        return allreduce_flat_(model.grads)
    work, lo = pending
    model._tail_work = None
    scale = allreduce_flat_(model.grads[:lo])
    work.wait()          # (RCCL: the current stream waits; gloo: the host does)
    return scale


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s weights (one flat broadcast)."""
    if is_distributed():
        dist.broadcast(model.params, src=src)


def allreduce_mean_scalar(value_tensor):
    """Mean of a scalar (e.g. the loss) over ranks, for logging."""
    if not is_distributed():
        return value_tensor
    t = value_tensor.detach().clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t / dist.get_world_size()


def agree_step(n_samples, ok=True, device='cpu'):
    """Make the per-step decisions the SAME on every rank before any compute
    or collective of that step is issued: returns (T, all_ok) with T = the
    smallest per-clip sample count any rank holds for this step and all_ok =
    every rank produced a batch.  Readers cut files into pieces
    (audio_reader.py:167-174), so ranks routinely see different tail lengths:
    a rank skipping a step on its own (`if T < 2: continue`) or failing in its
    reader thread would leave the others inside the gradient all-reduce.  With
    the common T every rank truncates to, the per-rank B*T are equal and the
    1/N average of the module docstring is exact."""
    if not is_distributed():
        return int(n_samples), bool(ok)
    group, device = _ctl(device)
    t = torch.tensor([int(n_samples), 1 if ok else 0], dtype=torch.int64,
                     device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    t = t.cpu()
    return int(t[0]), bool(int(t[1]))


def any_rank(flag, device='cpu'):
    """True on EVERY rank when `flag` is true on at least one (all-reduce MAX):
    for decisions that must be taken by all ranks in the same step (aborting
    on a device error one rank saw)."""
    if not is_distributed():
        return bool(flag)
    group, device = _ctl(device)
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.cpu()[0]))
