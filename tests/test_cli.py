"""train.py / generate.py counterparts: flags, directory rules, checkpoint
naming (reference train.py:37-180, generate.py:24-116).  CPU part."""
import os
import sys

import pytest

from util import ROOT

sys.path.insert(0, ROOT)
import train  # noqa: E402
import generate  # noqa: E402


def test_train_flag_defaults():
    a = train.get_arguments([])
    assert (a.batch_size, a.checkpoint_every, a.num_steps) == (1, 50, 100000)
    assert (a.learning_rate, a.sample_size, a.optimizer) == (1e-3, 100000, 'adam')
    assert (a.momentum, a.silence_threshold) == (0.9, 0.3)
    assert a.l2_regularization_strength == 0 and a.gc_channels is None
    assert a.data_dir == './VCTK-Corpus' and a.histograms is False


def test_validate_directories_rules():
    a = train.get_arguments(['--logdir', 'x', '--logdir_root', 'y'])
    with pytest.raises(ValueError, match='cannot be specified at the same'):
        train.validate_directories(a)
    a = train.get_arguments(['--logdir', 'x', '--restore_from', 'y'])
    with pytest.raises(ValueError, match='unexpected overwrites'):
        train.validate_directories(a)
    d = train.validate_directories(train.get_arguments(['--logdir', 'x']))
    assert d == {'logdir': 'x', 'logdir_root': None, 'restore_from': 'x'}
    d = train.validate_directories(train.get_arguments(
        ['--logdir_root', 'r', '--restore_from', 'old']))
    assert d['logdir'].startswith(os.path.join('r', 'train'))
    assert d['restore_from'] == 'old' and d['logdir'] != d['restore_from']
    d = train.validate_directories(train.get_arguments([]))
    assert d['logdir'].startswith(os.path.join('./logdir', 'train'))


def test_checkpoint_naming(tmp_path):
    assert train.checkpoint_path('d', 150) == os.path.join('d', 'model.ckpt-150')
    assert train.latest_checkpoint(str(tmp_path)) is None
    for s in (3, 20, 100):
        open(train.checkpoint_path(str(tmp_path), s), 'w').close()
    assert train.latest_checkpoint(str(tmp_path)).endswith('model.ckpt-100')
    with open(os.path.join(str(tmp_path), 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model.ckpt-20"\n')
    assert train.latest_checkpoint(str(tmp_path)).endswith('model.ckpt-20')


def test_latest_checkpoint_ignores_v2_data_shards_and_meta(tmp_path):
    """A logdir holding only a step-0 TensorFlow V2 checkpoint and no
    `checkpoint` marker: the prefix is the checkpoint, not the data shard
    (whose name also ends in digits) nor the .meta graph."""
    d = str(tmp_path)
    for name in ('model.ckpt-0.index', 'model.ckpt-0.data-00000-of-00001',
                 'model.ckpt-0.meta', 'model.ckpt-7.data-00000-of-00002',
                 'model.ckpt-7.data-00001-of-00002', 'model.ckpt-7.index',
                 'model.ckpt-best'):
        open(os.path.join(d, name), 'w').close()
    assert train.latest_checkpoint(d) == os.path.join(d, 'model.ckpt-7')
    os.remove(os.path.join(d, 'model.ckpt-7.index'))
    assert train.latest_checkpoint(d) == os.path.join(d, 'model.ckpt-0')


def test_generate_flags():
    a = generate.get_arguments(['ck'])
    assert (a.samples, a.temperature, a.window) == (16000, 1.0, 8000)
    assert a.fast_generation is True and a.save_every is None
    assert generate.get_arguments(['ck', '--fast_generation', 'false']
                                  ).fast_generation is False
    with pytest.raises(ValueError, match='gc_cardinality'):
        generate.get_arguments(['ck', '--gc_channels', '32'])
    with pytest.raises(ValueError, match='gc_id'):
        generate.get_arguments(['ck', '--gc_channels', '32',
                                '--gc_cardinality', '377'])
    with pytest.raises(SystemExit):
        generate.get_arguments(['ck', '--temperature', '-1'])


def test_synthetic_reader():
    r = train.SyntheticReader(1000, gc_cardinality=7)
    a = r.dequeue(3)
    g = r.dequeue_gc(3)
    assert tuple(a.shape) == (3, 1000, 1) and float(a.abs().max()) <= 1
    assert tuple(g.shape) == (3,) and int(g.max()) < 7


def _run_snippet(code, timeout=120):
    import subprocess
    p = subprocess.run([sys.executable, '-c', code], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_bench_emitter_watchdog_prints_the_prepared_line_and_exits_zero():
    """bench.py's first-contact guard (N > 1: the two-call all-reduce trial):
    a guarded region that does not return within its bound makes rank 0 print
    the line prepared from the schedule already measured -- with the reason in
    place of the marker -- exactly once, and the process leaves with status 0."""
    code = (
        "import time, bench\n"
        "e = bench._Emitter()\n"
        "e.fallback = '{\"value\": 1.0, \"overlap_failed\": true, \"overlap_failure\": \"@WHY@\"}'\n"
        "e.guarded(lambda: time.sleep(60), 'the two-call all-reduce trial', bound=0.5)\n"
        "print('not reached')\n")
    rc, out, err = _run_snippet(code)
    assert rc == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and 'not reached' not in out
    import json
    r = json.loads(lines[0])
    assert r['overlap_failed'] is True
    assert 'did not return within' in r['overlap_failure'] and 'two-call' in r['overlap_failure']


def test_bench_emitter_prints_once_and_cancels_its_watchdog():
    """The normal path: the guarded region returns, its watchdog is cancelled,
    and only the first `line` call prints."""
    code = (
        "import time, bench\n"
        "e = bench._Emitter()\n"
        "e.fallback = 'FALLBACK @WHY@'\n"
        "assert e.guarded(lambda: 7, 'x', bound=0.3) == 7\n"
        "time.sleep(0.8)\n"
        "e.line('first')\n"
        "e.line('second')\n")
    rc, out, err = _run_snippet(code)
    assert rc == 0, err[-2000:]
    assert out.split() == ['first']
