"""CPU ORACLE (second formulation) -- TEST INFRASTRUCTURE ONLY.

The reference's TensorFlow graph restated OP FOR OP in PyTorch-CPU, unfused:
pad -> time_to_batch -> width-K conv ('SAME') -> batch_to_time -> slice,
separately for filter and gate (wavenet/ops.py:27-62, model.py:269-270), bias,
tanh*sigmoid, dense 1x1, skip 1x1 (model.py:286-312), the skip tensors
materialised and summed sequentially (model.py:430), ReLU/1x1/ReLU/1x1
(model.py:431-440), shifted one-hot softmax cross-entropy mean
(model.py:654-666); gradients by torch.autograd.

Two uses, both as a checker/baseline only:
  * an independent cross-check of oracle/wavenet_oracle.py (forward, loss and
    analytic gradients) -- tests/test_oracle_crosscheck.py;
  * bench.py's "cpu_baseline" leg (kind "port": TensorFlow 0.10 cannot be
    installed here, so this is a CPU restatement of the reference graph, not
    TensorFlow itself).
Never imported by the shipped path.
"""
import math

import torch
import torch.nn.functional as F


def to_torch(var, dtype=torch.float32, requires_grad=False):
    if isinstance(var, dict):
        return {k: to_torch(v, dtype, requires_grad) for k, v in var.items()}
    if isinstance(var, list):
        return [to_torch(v, dtype, requires_grad) for v in var]
    t = torch.tensor(var, dtype=dtype)
    t.requires_grad_(requires_grad)
    return t


def leaves(var):
    if isinstance(var, dict):
        out = []
        for v in var.values():
            out += leaves(v)
        return out
    if isinstance(var, list):
        out = []
        for v in var:
            out += leaves(v)
        return out
    return [var]


def time_to_batch(value, dilation):
    b, t, c = value.shape
    pad_elements = dilation - 1 - (t + dilation - 1) % dilation
    padded = F.pad(value, (0, 0, 0, pad_elements))
    reshaped = padded.reshape(-1, dilation, c)
    transposed = reshaped.permute(1, 0, 2)
    return transposed.reshape(b * dilation, -1, c)


def batch_to_time(value, dilation):
    b, t, c = value.shape
    prepared = value.reshape(dilation, -1, c)
    transposed = prepared.permute(1, 0, 2)
    return transposed.reshape(b // dilation, -1, c)


def conv1d_same(x, w):
    """tf.nn.conv1d(x[B,T,Cin], w[K,Cin,Cout], stride=1, 'SAME')."""
    k = w.shape[0]
    pl = (k - 1) // 2
    pr = k - 1 - pl
    xp = F.pad(x, (0, 0, pl, pr)).permute(0, 2, 1)          # [B,Cin,T+K-1]
    return F.conv1d(xp, w.permute(2, 1, 0)).permute(0, 2, 1)


def causal_conv(value, filter_, dilation):
    k = filter_.shape[0]
    padded = F.pad(value, (0, 0, (k - 1) * dilation, 0))
    if dilation > 1:
        transformed = time_to_batch(padded, dilation)
        conv = conv1d_same(transformed, filter_)
        restored = batch_to_time(conv, dilation)
    else:
        restored = conv1d_same(padded, filter_)
    return restored[:, :value.shape[1], :]


def network(cfg, var, net_in, gc_emb=None):
    use_b = cfg.get('use_biases', False)
    dil = cfg['dilations']
    x = causal_conv(net_in, var['causal_layer']['filter'], 1)
    outputs = []
    for i, d in enumerate(dil):
        v = var['dilated_stack'][i]
        cf = causal_conv(x, v['filter'], d)
        cg = causal_conv(x, v['gate'], d)
        if gc_emb is not None:
            cf = cf + conv1d_same(gc_emb, v['gc_filtweights'])
            cg = cg + conv1d_same(gc_emb, v['gc_gateweights'])
        if use_b:
            cf = cf + v['filter_bias']
            cg = cg + v['gate_bias']
        out = torch.tanh(cf) * torch.sigmoid(cg)
        skip = conv1d_same(out, v['skip'])
        if i != len(dil) - 1:
            tr = conv1d_same(out, v['dense'])
            if use_b:
                tr = tr + v['dense_bias']
        if use_b:
            skip = skip + v['skip_bias']
        outputs.append(skip)
        if i != len(dil) - 1:
            x = x + tr
    total = 0
    for o in outputs:               # python sum(outputs): sequential adds
        total = total + o
    p = var['postprocessing']
    t1 = F.relu(total)
    c1 = conv1d_same(t1, p['postprocess1'])
    if use_b:
        c1 = c1 + p['postprocess1_bias']
    t2 = F.relu(c1)
    if cfg.get('residual_postproc', False):
        t2 = t2 + total
    c2 = conv1d_same(t2, p['postprocess2'])
    if use_b:
        c2 = c2 + p['postprocess2_bias']
    return c2


def loss(cfg, var, q, audio=None, gc_ids=None, l2=None,
         tf_bias_name_quirk=True, names=None):
    """q: int64 [B,T] mu-law codes (encoding itself is integer work pinned
    elsewhere); audio only for scalar_input."""
    Q = cfg.get('quantization_channels', 256)
    B = cfg['batch_size']
    dtype = var['causal_layer']['filter'].dtype
    enc = F.one_hot(q.reshape(B, -1), Q).to(dtype)
    if cfg.get('scalar_input', False):
        net_in = audio.to(dtype).reshape(B, -1, 1)
    else:
        net_in = enc
    gc_emb = None
    if cfg.get('global_condition_cardinality') is not None and \
            gc_ids is not None:
        gc_emb = var['embeddings']['gc_embedding'][gc_ids.reshape(-1)]
        gc_emb = gc_emb.reshape(B, 1, cfg['global_condition_channels'])
    raw = network(cfg, var, net_in, gc_emb)
    shifted = F.pad(enc[:, 1:, :], (0, 0, 0, 1))
    pred = raw.reshape(-1, Q)
    lab = shifted.reshape(-1, Q)
    row = -(lab * F.log_softmax(pred, dim=-1)).sum(-1)
    reduced = row.mean()
    if l2 is None:
        return reduced
    terms = []
    for n, v in names:
        if 'bias' in n.split('/')[-1] and not tf_bias_name_quirk:
            continue
        terms.append((v ** 2).sum() / 2)
    return reduced + l2 * sum(terms)


class TFAdam(object):
    """tf.train.AdamOptimizer(lr, epsilon=1e-4) update rule (ops.py:6-8)."""

    def __init__(self, params, lr, eps=1e-4, b1=0.9, b2=0.999):
        self.p, self.lr, self.eps, self.b1, self.b2 = params, lr, eps, b1, b2
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / \
            (1 - self.b1 ** self.t)
        for p, m, v in zip(self.p, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.sub_(lr_t * m / (v.sqrt() + self.eps))
            p.grad = None


def train_step_fn(cfg, var_np, lr=1e-3):
    """Returns step(q[int64 B,T]) -> loss float: one full training step
    (forward, loss, autograd backward, TF-Adam) of the op-for-op graph."""
    var = to_torch(var_np, torch.float32, requires_grad=True)
    params = leaves(var)
    opt = TFAdam(params, lr)

    def step(q):
        l = loss(cfg, var, q)
        l.backward()
        opt.step()
        return float(l.detach())

    return step
