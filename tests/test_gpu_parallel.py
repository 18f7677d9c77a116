"""Data-parallel path with the REAL model on the GPU: N ranks x
B/N clips through net.loss -> optimizer.minimize must land on the same
parameters as one process with the whole batch; the RCCL backend initialises
and reduces the device gradient bucket; bench.py refuses a rank-count
mismatch and starts its own ranks when run plainly."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import ROOT

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(spec, world, extra_env=None, timeout=600):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.update(extra_env or {})
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, 'dp_worker.py'),
             json.dumps(spec)], env=env, stdout=subprocess.PIPE,
            stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o.decode(errors='replace'))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


@pytest.mark.parametrize('opt,gc', [('adam', False), ('sgd', True)])
def test_two_ranks_real_model_equal_single_process(hip_lib, tmp_path, opt, gc):
    """2 ranks x 2 clips (sharing this box's GPU over gloo) == 1 process x 4
    clips after 2 full training steps, parameters <= 1e-6."""
    spec = dict(mode='dp', B=4, T=300, steps=2, opt=opt, lr=1e-3,
                cfg=dict(global_condition_channels=4,
                         global_condition_cardinality=5) if gc else {})
    spec['out'] = str(tmp_path / 'dp2.npz')
    _run_ranks(spec, 2, dict(WN_SHARE_GPU='1', WN_DIST_BACKEND='gloo'))
    one = dict(spec, out=str(tmp_path / 'dp1.npz'))
    _run_ranks(one, 1)
    a = np.load(spec['out'])
    b = np.load(one['out'])
    assert np.abs(a['losses'] - b['losses']).max() < 1e-6
    assert np.abs(a['params'] - b['params']).max() <= 1e-6


# The GPU box admits at most 6 processes on its card (this pytest process is
# one of them), so the many-rank rehearsals below run FOUR ranks on the one
# GPU over gloo; the 8-rank arithmetic (shard ranges, 1/N, gc ids of 64 clips,
# broadcast, step agreement) is covered on CPU by tests/test_parallel_gloo.py.
REHEARSAL_RANKS = int(os.environ.get('WN_REHEARSE_RANKS', 4))


def test_many_ranks_real_model_equal_single_process(hip_lib, tmp_path):
    """REHEARSAL_RANKS ranks x 1 clip == 1 process x that many clips after two
    full training steps (parameters and per-step global losses <= 1e-6)."""
    n = REHEARSAL_RANKS
    spec = dict(mode='dp', B=n, T=300, steps=2, opt='adam', lr=1e-3,
                cfg=dict(global_condition_channels=4,
                         global_condition_cardinality=5))
    spec['out'] = str(tmp_path / 'dpn.npz')
    _run_ranks(spec, n, dict(WN_SHARE_GPU='1', WN_DIST_BACKEND='gloo'))
    one = dict(spec, out=str(tmp_path / 'dp1.npz'))
    _run_ranks(one, 1)
    a = np.load(spec['out'])
    b = np.load(one['out'])
    assert np.abs(a['losses'] - b['losses']).max() < 1e-6
    assert np.abs(a['params'] - b['params']).max() <= 1e-6


@pytest.mark.parametrize('B,T', [(4, 300), (8, 17000)], ids=['side_stream_gemms', 'one_stream'])
def test_tail_allreduce_inside_backward_equals_one_call(hip_lib, tmp_path, B, T):
    """`dp_overlap_allreduce` (the skip / post-processing gradients' all-reduce
    issued from inside the backward pass on a communication stream, the rest at
    the update) against one all-reduce of the whole bucket: REHEARSAL_RANKS
    ranks of the real model on this GPU over gloo, three Adam steps through
    eager, recorded and replayed launch plans.  Same values summed per element;
    gloo's ring may add a chunk's terms in another order (1 ulp of a gradient),
    so the parameters agree to 1e-7, and to 1e-6 with one process x the whole
    batch.  One clip of 300 samples per rank runs the weight-gradient GEMMs on
    the side stream (the all-reduce starts after their join); two clips of
    17000 (more than 1024 tiles) on the main stream (it starts before the
    backward stack launch)."""
    n = REHEARSAL_RANKS
    spec = dict(mode='dp', B=n * (B // 4), T=T, steps=3, opt='adam', lr=1e-3,
                cfg=dict(global_condition_channels=4, global_condition_cardinality=5))
    env = dict(WN_SHARE_GPU='1', WN_DIST_BACKEND='gloo')
    res = {}
    for ov in (False, True):
        out = str(tmp_path / ('ov%d.npz' % ov))
        _run_ranks(dict(spec, overlap=ov, out=out), n, env)
        res[ov] = np.load(out)
    one = str(tmp_path / 'one.npz')
    _run_ranks(dict(spec, out=one), 1)
    one = np.load(one)
    assert np.abs(res[True]['losses'] - res[False]['losses']).max() <= 1e-7
    assert np.abs(res[True]['params'] - res[False]['params']).max() <= 1e-7
    assert np.abs(res[True]['params'] - one['params']).max() <= 1e-6
    assert np.abs(res[True]['losses'] - one['losses']).max() < 1e-6


def _bench_json(args, env, timeout=900):
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args
                       + ['--no-secondary', '--no-cpu-baseline'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1]
    return json.loads(line), p.stderr.decode()


@pytest.mark.parametrize('gc', [False, True], ids=['plain', 'gc'])
def test_bench_many_ranks_rehearsal(hip_lib, gc):
    """`python bench.py --gpus N [--gc]` (configs[2] / configs[3] with N ranks
    of one clip each, sharing this box's GPU over gloo): every rank is seen,
    the global batch and the per-rank speaker ids are what the 8-GPU run will
    use, every rank names its device before the first step, the JSON says
    which collective library / settings ran -- and the loss of the global
    batch after the same number of steps equals ONE process with N clips."""
    n = REHEARSAL_RANKS
    env = dict(os.environ, WN_SHARE_GPU='1', WN_DIST_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    common = ['--steps', '2', '--warmup', '1', '--samples', '4000'] + \
        (['--gc'] if gc else [])
    r, err = _bench_json(['--gpus', str(n), '--batch', '1'] + common, env)
    assert r['n_gpus'] == n and r['ranks_seen'] == n
    assert r['config']['global_batch'] == n
    assert r['allreduce_bytes'] == 4 * (1630432 if gc else 1515968)
    assert r['collective']['rccl_version'] and 'NCCL_ALGO' in r['collective']
    assert r['allreduce_us_per_step'] > 0
    # both gradient-exchange schedules were timed in the same process, the
    # line carries both and says which one `value` is from
    sch = r['allreduce_schedules']
    assert set(sch) == {'one_call', 'two_call'} and r['overlap_failed'] is False
    assert r['allreduce_calls'] in (1, 2)
    best = min(sch.values(), key=lambda e: e['ms_per_step'])
    assert r['ms_per_step'] == best['ms_per_step']
    assert r['allreduce_calls'] == (2 if best is sch['two_call'] else 1)
    assert all(e['ms_per_step'] > 0 and np.isfinite(e['final_loss']) for e in sch.values())
    for k in range(n):
        assert '[bench] rank %d/%d' % (k, n) in err and 'PCI' in err
    if gc:
        assert r['config']['gc_ids'] == [(37 * b) % 377 for b in range(n)]
        assert len(set(r['config']['gc_ids'])) == n
    env1 = {k: v for k, v in env.items() if k not in ('WN_SHARE_GPU', 'WN_DIST_BACKEND')}
    one, _ = _bench_json(['--gpus', '1', '--batch', str(n)] + common, env1)
    assert one['config']['global_batch'] == n
    # (the weight-gradient slabs of N x 1 clip and 1 x N clips group the tiles
    # differently: after three optimizer steps the float32 losses, ~5.3, may
    # differ by a few units in the last place, 4.8e-7 each)
    # (N > 1 goes on to time the two-call schedule: the comparison is with the
    # one-call schedule's loss, taken after the same number of steps)
    assert abs(one['config']['global_loss'] -
               r['allreduce_schedules']['one_call']['global_loss']) <= 5e-6
    if gc:
        assert one['config']['gc_ids'] == r['config']['gc_ids']


@pytest.mark.parametrize('how', ['raise', 'hang', 'nan'])
def test_bench_overlap_trial_failure_falls_back_in_process(hip_lib, how):
    """First-contact guard of `bench.py --gpus N`: when the two-call schedule
    raises, hangs (watchdog) or produces a non-finite loss, the SAME processes
    still print ONE line -- the one-call figures, "overlap_failed": true, the
    reason -- and exit 0."""
    env = dict(os.environ, WN_SHARE_GPU='1', WN_DIST_BACKEND='gloo',
               WN_BENCH_INJECT_OVERLAP_FAILURE=how, WN_OVERLAP_TRIAL_TIMEOUT='20')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                        '--batch', '1', '--steps', '2', '--warmup', '1', '--samples',
                        '4000', '--no-secondary', '--no-cpu-baseline'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r['overlap_failed'] is True and r['allreduce_calls'] == 1
    assert set(r['allreduce_schedules']) == {'one_call'}
    assert r['ms_per_step'] == r['allreduce_schedules']['one_call']['ms_per_step']
    assert r['value'] > 0 and r['ranks_seen'] == 2
    want = {'raise': 'injected failure', 'hang': 'did not return within', 'nan': 'non-finite'}
    assert want[how] in r['overlap_failure']


@pytest.mark.parametrize('B,T', [(1, 4000), (8, 16000)], ids=['side_stream_gemms', 'one_stream'])
def test_rccl_world_one_two_call_schedule_beside_the_backward_stack(hip_lib, tmp_path, B, T):
    """The two-call schedule through RCCL itself (backend "nccl", world size
    1, parallel.rehearse_world_one): the tail's all-reduce KERNEL runs on the
    communication stream beside the persistent backward-stack launch of the
    default stack (B = 8: issued before the stack launch; B = 1: after the
    side-stream GEMMs' join), the head's at the update; three Adam steps
    through eager, recorded and replayed launch plans.  A sum over one rank is
    the identity, so losses and parameters equal the one-call run BITWISE, no
    dependency wait expires and no tail is left dangling."""
    res = {}
    for ov in (False, True):
        out = str(tmp_path / ('w1_%d.npz' % ov))
        _run_ranks(dict(mode='nccl1_overlap', B=B, T=T, steps=3, overlap=ov, out=out), 1)
        res[ov] = np.load(out)
    assert res[True]['comm_stream_used'] and not res[False]['comm_stream_used']
    assert np.array_equal(res[True]['losses'], res[False]['losses'])
    assert np.array_equal(res[True]['params'], res[False]['params'])
    assert np.isfinite(res[True]['losses']).all()


def test_rccl_world_one_allreduces_the_gradient_bucket(hip_lib, tmp_path):
    """backend "nccl" (RCCL) comes up on the device and all-reduces /
    broadcasts the flat fp32 buckets of the real model."""
    spec = dict(mode='nccl1', B=2, T=200, steps=1, out=str(tmp_path / 'n.json'))
    _run_ranks(spec, 1)
    r = json.load(open(spec['out']))
    assert r['ok'] and r['backend'] == 'nccl' and r['absmax'] > 0


def test_bench_self_launches_two_ranks(hip_lib):
    """`python bench.py --gpus 2` run plainly starts two rank processes
    (sharing this box's one GPU over gloo here; RCCL on a multi-GPU node) and
    reports both in `ranks_seen`."""
    env = dict(os.environ, WN_SHARE_GPU='1', WN_DIST_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'),
                        '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--batch', '2', '--samples', '4000',
                        '--no-secondary', '--no-cpu-baseline'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1]
    r = json.loads(line)
    assert r['n_gpus'] == 2 and r['ranks_seen'] == 2
    assert r['config']['global_batch'] == 4
    assert r['value'] > 0
    # what the first multi-GPU run must tell: all-reduce share, rank spread
    assert r['allreduce_us_per_step'] > 0
    assert 0 < r['step_ms_min'] <= r['step_ms_max']
    assert 0 < r['step_frac'] < 1 and r['roofline']['frac'] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2,
                    reason='needs >= 2 GPUs (RCCL refuses two ranks per device)')
def test_bench_two_gpus_over_rccl(hip_lib):
    """On a multi-GPU box: `python bench.py --gpus 2` with the default
    backend ("nccl" = RCCL over xGMI), one rank per device."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'WN_SHARE_GPU',
              'WN_DIST_BACKEND', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'),
                        '--gpus', '2', '--steps', '3', '--warmup', '2',
                        '--batch', '2', '--samples', '8000',
                        '--no-secondary', '--no-cpu-baseline'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1]
    r = json.loads(line)
    assert r['dist_backend'] == 'nccl'
    assert r['n_gpus'] == 2 and r['ranks_seen'] == 2
    assert r['allreduce_us_per_step'] > 0


def test_train_py_two_ranks_on_wav_data(hip_lib, tmp_path):
    """train.py itself under two ranks (one GPU, gloo) on a wav corpus whose
    files are sharded over the ranks: the ranks see different piece lengths
    (tail pieces), so every step goes through parallel.agree_step; the run
    must finish on both ranks (no collective left hanging), log a mean loss
    per step and leave one checkpoint."""
    from scipy.io import wavfile
    data = tmp_path / 'corpus'
    rate = 16000
    for spk, n in ((225, 3), (226, 2)):
        os.makedirs(str(data / ('p%d' % spk)))
        for rec in range(n):
            t = np.arange(int(rate * (0.55 + 0.17 * rec + 0.05 * spk % 3))) / rate
            sig = 0.5 * np.sin(2 * np.pi * (200 + 40 * rec) * t)
            wavfile.write(str(data / ('p%d' % spk) / ('p%d_%03d.wav' % (spk, rec))),
                          rate, (sig * 32767).astype(np.int16))
    params = dict(filter_width=2, sample_rate=16000,
                  dilations=[1, 2, 4, 8, 16, 32] * 2, residual_channels=32,
                  dilation_channels=32, quantization_channels=256,
                  skip_channels=64, use_biases=True, scalar_input=False,
                  initial_filter_width=32)
    pj = str(tmp_path / 'params.json')
    json.dump(params, open(pj, 'w'))
    logdir = str(tmp_path / 'run')
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   WN_DIST_BACKEND='gloo', WN_DIST_TIMEOUT='120')
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, 'train.py'), '--data_dir',
             str(data), '--sample_size', '3000', '--batch_size', '1',
             '--wavenet_params', pj, '--logdir', logdir, '--num_steps', '8',
             '--silence_threshold', '0', '--learning_rate', '0.002'],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o.decode(errors='replace'))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    ev = [json.loads(l) for l in open(os.path.join(logdir, 'events.jsonl'))]
    assert len(ev) >= 5 and all(np.isfinite(e['loss']) for e in ev)
    assert any(f.startswith('model.ckpt-') for f in os.listdir(logdir))
