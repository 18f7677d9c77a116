"""wavenet/tf_checkpoint.py: the TensorFlow-checkpoint reader against files
written by tests/tf_ckpt_writer.py (both restate the published formats; no
TensorFlow here, so this is a self-consistency check, see the module
docstring), the bias-name mapping of model.py:28, and the train.py /
generate.py entry points."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import tf_ckpt_writer as W  # noqa: E402
from util import model_kwargs  # noqa: E402
from wavenet import WaveNetModel, tf_checkpoint as C  # noqa: E402


def _tensors(seed=0):
    rng = np.random.RandomState(seed)
    t = {'wavenet/causal_layer/filter': rng.randn(2, 16, 8).astype(np.float32),
         'wavenet/dilated_stack/layer0/Variable': rng.randn(8).astype(np.float32),
         'wavenet/dilated_stack/layer0/Variable_1': rng.randn(8).astype(np.float32),
         'big': rng.randn(300, 40).astype(np.float32),          # several 4 KiB blocks
         'doubles': rng.randn(3, 2).astype(np.float64),
         'step': np.asarray([12345, -7], dtype=np.int64),
         'ints': np.arange(-5, 6, dtype=np.int32),
         'scalar': np.asarray(2.5, dtype=np.float32)}
    for i in range(40):                                        # prefix-compressed keys
        t['wavenet/dilated_stack/layer%d/filter' % i] = rng.randn(2, 4, 4).astype(np.float32)
    return t


@pytest.mark.parametrize('snappy', [False, True])
@pytest.mark.parametrize('fmt', ['v1', 'v2'])
def test_reader_round_trip(tmp_path, fmt, snappy):
    want = _tensors()
    path = str(tmp_path / 'model.ckpt-77')
    (W.write_v1 if fmt == 'v1' else W.write_v2)(path, want, snappy=snappy)
    assert C.checkpoint_format(path) == fmt
    got = C.read_checkpoint(path)
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        np.testing.assert_array_equal(got[k], want[k])


def test_v1_short_value_lists_and_foreign_dtypes(tmp_path):
    """TensorProto semantics a real Saver file may use: a typed value list
    shorter than the tensor repeats its LAST value (a constant is stored as one
    value), an empty one means zeros; a tensor of a dtype that cannot be a
    model weight (DT_STRING = 7) is skipped in V1 as it is in V2."""
    import struct
    path = str(tmp_path / 'model.ckpt-3')
    raw = {'const': (1, (2, 3), W.field(5, 2, struct.pack('<f', 0.25))),
           'ramp_then_flat': (1, (5,), W.field(5, 2, struct.pack('<2f', 1.0, 2.0))),
           'empty': (1, (4,), b''),
           'label': (7, (1,), W.field(8, 2, b'hello'))}
    W.write_v1(path, {'w': np.arange(6, dtype=np.float32).reshape(2, 3)},
               raw_tensors=raw)
    got = C.read_checkpoint(path)
    assert sorted(got) == ['const', 'empty', 'ramp_then_flat', 'w']
    np.testing.assert_array_equal(got['const'], np.full((2, 3), 0.25, np.float32))
    np.testing.assert_array_equal(got['ramp_then_flat'],
                                  np.array([1, 2, 2, 2, 2], np.float32))
    np.testing.assert_array_equal(got['empty'], np.zeros(4, np.float32))


def test_known_answers():
    assert C.crc32c(b'123456789') == 0xe3069283           # the CRC-32C check value
    assert C.crc32c(b'\0' * 32) == 0x8a9136aa             # RFC 3720 B.4
    # snappy: literal "abcd", then a copy of length 6 at offset 4
    assert C.snappy_decompress(bytes([10, 3 << 2]) + b'abcd' + bytes([((6 - 4) << 2) | 1, 4])) \
        == b'abcdabcdab'
    assert C.checkpoint_format('/nonexistent/model.ckpt-1') is None


def test_corruption_is_detected(tmp_path):
    path = str(tmp_path / 'model.ckpt-1')
    W.write_v1(path, _tensors())
    raw = bytearray(open(path, 'rb').read())
    raw[100] ^= 0x40
    open(path, 'wb').write(bytes(raw))
    with pytest.raises(ValueError):
        C.read_checkpoint(path)
    open(path, 'wb').write(bytes(raw[:-8]) + b'notmagic')
    assert C.checkpoint_format(path) is None


def _small_net(**over):
    cfg = dict(batch_size=1, dilations=[1, 2, 4], filter_width=2, residual_channels=8,
               dilation_channels=8, skip_channels=12, quantization_channels=16,
               use_biases=True, scalar_input=False, initial_filter_width=2)
    cfg.update(over)
    return WaveNetModel(device='cpu', seed=3, **model_kwargs(cfg))


@pytest.mark.parametrize('fmt', ['v1', 'v2'])
@pytest.mark.parametrize('bias_names', ['reference', 'intended'])
def test_load_into_model(tmp_path, fmt, bias_names):
    """A checkpoint with the reference's variable names -- biases as
    `Variable[_n]` per scope (model.py:28) or under their intended names --
    restores every variable of the model."""
    src = _small_net(global_condition_channels=4, global_condition_cardinality=3)
    want = {n: v.detach().numpy().copy() for n, v in src.named_variables()}
    ck = {}
    order = {'filter_bias': 0, 'gate_bias': 1, 'dense_bias': 2, 'slip_bias': 3,
             'postprocess1_bias': 0, 'postprocess2_bias': 1}
    for n, a in want.items():
        scope, leaf = n.rsplit('/', 1)
        if bias_names == 'reference' and leaf in order:
            k = order[leaf]
            n = scope + ('/Variable' if k == 0 else '/Variable_%d' % k)
        ck[n] = a
    ck['wavenet/optimizer/beta1_power'] = np.asarray(0.9, dtype=np.float32)   # ignored
    path = str(tmp_path / 'model.ckpt-5')
    (W.write_v1 if fmt == 'v1' else W.write_v2)(path, ck)
    dst = _small_net(global_condition_channels=4, global_condition_cardinality=3)
    unused = C.load_into(dst, path)
    assert unused == ['wavenet/optimizer/beta1_power']
    for n, v in dst.named_variables():
        np.testing.assert_array_equal(v.detach().numpy(), want[n])
    # a model of another shape refuses the file
    with pytest.raises((ValueError, KeyError)):
        C.load_into(_small_net(residual_channels=4), path)


def test_train_py_restores_a_tf_checkpoint(tmp_path):
    """train.load (train.py:117-134) picks up a TensorFlow checkpoint left in
    the log directory by the reference."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    import train
    src = _small_net()
    want = {n: v.detach().numpy().copy() for n, v in src.named_variables()}
    logdir = str(tmp_path / 'logdir')
    os.makedirs(logdir)
    W.write_v1(os.path.join(logdir, 'model.ckpt-42'), want)
    with open(os.path.join(logdir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model.ckpt-42"\n'
                'all_model_checkpoint_paths: "model.ckpt-42"\n')
    dst = _small_net()
    assert train.load(dst, logdir) == 42
    for n, v in dst.named_variables():
        np.testing.assert_array_equal(v.detach().numpy(), want[n])
    # V2: only model.ckpt-43.index / .data-* exist
    W.write_v2(os.path.join(logdir, 'model.ckpt-43'), want)
    with open(os.path.join(logdir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model.ckpt-43"\n')
    assert train.load(_small_net(), logdir) == 43
