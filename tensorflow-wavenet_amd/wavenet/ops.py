"""Host-side mirror of the reference's wavenet/ops.py on top of the HIP C ABI.

Same names, argument meaning and error behaviour as
/root/reference/wavenet/ops.py: `optimizer_factory` (:6-24), `time_to_batch`
(:27-34), `batch_to_time` (:37-43), `causal_conv` (:46-62), `mu_law_encode`
(:65-73), `mu_law_decode` (:76-85).  Inputs may be numpy arrays or torch
tensors (the reference's tests pass numpy arrays, test_causal_conv.py:13-17);
results are torch tensors on the GPU.  Every function launches hand-written
HIP kernels; nothing here computes on the CPU.
"""
import math

import numpy as np
import torch

from . import _lib

_TABLES = {}


def _dev():
    _lib.require_gpu()
    return torch.device('cuda', torch.cuda.current_device())


def _as_dev(x, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(device=_dev(), dtype=dtype).contiguous()
    return torch.as_tensor(np.asarray(x), dtype=dtype).to(_dev()).contiguous()


def mu_law_tables(quantization_channels):
    """(thresholds[Q-1], decode_table[Q]) device tensors, cached per (Q, dev).
    Built once on the host by the library from the float32 chain of
    ops.py:65-85 (constant tables, like FFT twiddles -- not a compute path)."""
    dev = _dev()
    key = (int(quantization_channels), dev.index)
    if key not in _TABLES:
        q = int(quantization_channels)
        if q < 2:
            raise ValueError('quantization_channels must be >= 2, got %d' % q)
        thr = np.empty(q - 1, np.float32)
        lut = np.empty(q, np.float32)
        _lib.call('wn_mu_law_thresholds_host', q, thr.ctypes.data)
        _lib.call('wn_mu_law_decode_table_host', q, lut.ctypes.data)
        _TABLES[key] = (torch.from_numpy(thr).to(dev),
                        torch.from_numpy(lut).to(dev))
    return _TABLES[key]


def mu_law_encode(audio, quantization_channels):
    '''Quantizes waveform amplitudes (ops.py:65-73).  int32, same shape.'''
    a = _as_dev(audio, torch.float32)
    out = torch.empty(a.shape, dtype=torch.int32, device=a.device)
    if a.numel() == 0:
        return out
    thr, _ = mu_law_tables(quantization_channels)
    _lib.call('wn_mu_law_encode', _lib.ptr(a), _lib.ptr(out), a.numel(),
              _lib.ptr(thr), int(quantization_channels), _lib.stream())
    return out


def mu_law_decode(output, quantization_channels):
    '''Recovers waveform from quantized values (ops.py:76-85).  float32.'''
    c = _as_dev(output, torch.int32)
    out = torch.empty(c.shape, dtype=torch.float32, device=c.device)
    if c.numel() == 0:
        return out
    _, lut = mu_law_tables(quantization_channels)
    _lib.call('wn_mu_law_decode', _lib.ptr(c), _lib.ptr(out), c.numel(),
              _lib.ptr(lut), int(quantization_channels), _lib.stream())
    return out


def time_to_batch(value, dilation, name=None):
    """ops.py:27-34: [B,T,C] -> [B*dilation, ceil(T/dilation), C]."""
    v = _as_dev(value, torch.float32)
    b, t, c = v.shape
    u = (t + dilation - 1) // dilation
    out = torch.empty((b * dilation, u, c), dtype=torch.float32,
                      device=v.device)
    _lib.call('wn_time_to_batch', _lib.ptr(v), _lib.ptr(out), b, t, c,
              int(dilation), _lib.stream())
    return out


def batch_to_time(value, dilation, name=None):
    """ops.py:37-43: [B*dilation, U, C] -> [B, U*dilation, C]."""
    v = _as_dev(value, torch.float32)
    bd, u, c = v.shape
    if bd % dilation:
        raise ValueError('batch %d not divisible by dilation %d'
                         % (bd, dilation))
    b = bd // dilation
    out = torch.empty((b, u * dilation, c), dtype=torch.float32,
                      device=v.device)
    _lib.call('wn_batch_to_time', _lib.ptr(v), _lib.ptr(out), b, u, c,
              int(dilation), _lib.stream())
    return out


def causal_conv(value, filter_, dilation, name='causal_conv'):
    """ops.py:46-62: value [B,T,Cin], filter_ [K,Cin,Cout] -> [B,T,Cout].
    Any K (incl. TF's 'SAME' centring delay for K > 2), any channel counts."""
    v = _as_dev(value, torch.float32)
    w = _as_dev(filter_, torch.float32)
    if v.dim() != 3 or w.dim() != 3 or v.shape[2] != w.shape[1]:
        raise ValueError('causal_conv: value %s / filter %s mismatch'
                         % (tuple(v.shape), tuple(w.shape)))
    b, t, cin = v.shape
    k, _, cout = w.shape
    out = torch.empty((b, t, cout), dtype=torch.float32, device=v.device)
    _lib.call('wn_causal_conv', _lib.ptr(v), _lib.ptr(w), _lib.ptr(out), b, t,
              cin, cout, k, int(dilation), _lib.stream())
    return out


# ---------------------------------------------------------------------------
# optimizers (ops.py:6-24) with TensorFlow-0.10 update rules
# ---------------------------------------------------------------------------


class _Optimizer(object):
    """Fused flat-buffer optimizer.  `minimize(loss)` mirrors
    tf.train.Optimizer.minimize: `loss` is what WaveNetModel.loss returned
    (its gradients already sit in the model's flat gradient bucket); under
    torch.distributed the bucket is all-reduced (RCCL) and averaged first."""

    def __init__(self):
        self._slots = None
        self._step = 0

    def _make_slots(self, model):
        raise NotImplementedError

    def _apply(self, model, grad_scale):
        raise NotImplementedError

    def minimize(self, loss, var_list=None):
        model = getattr(loss, '_wn_model', None)
        if model is None:
            raise ValueError('minimize() needs the tensor returned by '
                             'WaveNetModel.loss()')
        if not getattr(loss, '_wn_has_grads', False):
            raise ValueError('loss was computed with backward=False')
        from . import parallel
        scale = parallel.allreduce_gradients(model)
        if self._slots is None:
            self._make_slots(model)
        self._step += 1
        self._apply(model, scale)
        return loss


class AdamOptimizer(_Optimizer):
    # tf.train.AdamOptimizer(learning_rate, epsilon=1e-4); `momentum` ignored
    def __init__(self, learning_rate, epsilon=1e-4, beta1=0.9, beta2=0.999):
        super(AdamOptimizer, self).__init__()
        self.lr, self.eps, self.b1, self.b2 = learning_rate, epsilon, beta1, beta2

    def _make_slots(self, model):
        self._slots = [torch.zeros_like(model.params),
                       torch.zeros_like(model.params)]

    def _apply(self, model, grad_scale):
        t = self._step
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        m, v = self._slots
        _lib.call('wn_adam', _lib.ptr(model.params), _lib.ptr(model.grads),
                  _lib.ptr(m), _lib.ptr(v), model.params.numel(), lr_t,
                  self.b1, self.b2, self.eps, grad_scale, 0.0, None,
                  _lib.stream())


class MomentumOptimizer(_Optimizer):
    def __init__(self, learning_rate, momentum):
        super(MomentumOptimizer, self).__init__()
        self.lr, self.mom = learning_rate, momentum

    def _make_slots(self, model):
        self._slots = [torch.zeros_like(model.params)]

    def _apply(self, model, grad_scale):
        _lib.call('wn_momentum', _lib.ptr(model.params), _lib.ptr(model.grads),
                  _lib.ptr(self._slots[0]), model.params.numel(), self.lr,
                  self.mom, grad_scale, 0.0, None, _lib.stream())


class RMSPropOptimizer(_Optimizer):
    # tf.train.RMSPropOptimizer(lr, decay=0.9, momentum, epsilon=1e-5);
    # the `rms` slot starts at ONE (TensorFlow), `momentum` slot at zero.
    def __init__(self, learning_rate, momentum, epsilon=1e-5, decay=0.9):
        super(RMSPropOptimizer, self).__init__()
        self.lr, self.mom, self.eps, self.decay = (learning_rate, momentum,
                                                   epsilon, decay)

    def _make_slots(self, model):
        self._slots = [torch.ones_like(model.params),
                       torch.zeros_like(model.params)]

    def _apply(self, model, grad_scale):
        _lib.call('wn_rmsprop', _lib.ptr(model.params), _lib.ptr(model.grads),
                  _lib.ptr(self._slots[0]), _lib.ptr(self._slots[1]),
                  model.params.numel(), self.lr, self.decay, self.mom,
                  self.eps, grad_scale, 0.0, None, _lib.stream())


def create_adam_optimizer(learning_rate, momentum):
    return AdamOptimizer(learning_rate=learning_rate, epsilon=1e-4)


def create_sgd_optimizer(learning_rate, momentum):
    return MomentumOptimizer(learning_rate=learning_rate, momentum=momentum)


def create_rmsprop_optimizer(learning_rate, momentum):
    return RMSPropOptimizer(learning_rate=learning_rate, momentum=momentum,
                            epsilon=1e-5)


optimizer_factory = {'adam': create_adam_optimizer,
                     'sgd': create_sgd_optimizer,
                     'rmsprop': create_rmsprop_optimizer}
