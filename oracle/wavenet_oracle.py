"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

A numpy restatement of the arithmetic of the reference's hot path
(jyegerlehner/tensorflow-wavenet): wavenet/ops.py and wavenet/model.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the shipped path (tensorflow-wavenet_amd/) never does.

Parity status
-------------
* PINNED by the reference's own known-answer tests (restated in
  tests/test_oracle_pins.py with the literal vectors):
    - causal_conv          test/test_causal_conv.py:11-27, 29-58
    - mu_law_encode/decode test/test_mu_law.py:37-51, 53-68, 113-124
* PARITY UNPINNED (the reference holds no golden tensor and TensorFlow 0.10,
  the third-party runtime holding the arithmetic, is absent from
  /root/reference and from this image, so the reference cannot be run here):
    residual-block outputs, full-stack logits, loss value, gradients,
    optimizer trajectories.  For these the oracle follows the reference
    source line by line (citations below) and is cross-checked against an
    independent second formulation (oracle/torch_graph.py: the reference
    graph op for op in PyTorch-CPU, gradients by autograd).

Everything is written for clarity, not speed; `dtype` selects float32 (the
reference's precision) or float64 (tight self-checks).
"""
import math

import numpy as np

# --------------------------------------------------------------------------
# mu-law  (wavenet/ops.py:65-85)
# --------------------------------------------------------------------------


def _log_f32(x32):
    """float32 log, defined as the correctly rounded result (float64 log
    rounded once to float32) so that the oracle does not depend on which SIMD
    log the host's numpy build dispatches to."""
    return np.log(x32.astype(np.float64)).astype(np.float32)


def mu_law_encode(audio, quantization_channels):
    """ops.py:65-73.  Pure float32 chain, C-style truncation to int32."""
    x = np.asarray(audio, dtype=np.float32)
    mu = np.float32(quantization_channels - 1)
    one = np.float32(1.0)
    # magnitude = log(1 + mu*|x|) / log(1. + mu)              ops.py:70
    num = _log_f32(one + mu * np.abs(x))
    den = _log_f32(np.asarray(one + mu, dtype=np.float32))
    magnitude = (num / den).astype(np.float32)
    signal = (np.sign(x).astype(np.float32) * magnitude).astype(np.float32)
    # cast((signal + 1) / 2 * mu + 0.5, int32)                 ops.py:73
    v = (signal + one).astype(np.float32)
    v = (v / np.float32(2.0)).astype(np.float32)
    v = (v * mu).astype(np.float32)
    v = (v + np.float32(0.5)).astype(np.float32)
    return v.astype(np.int32)  # truncation toward zero


def mu_law_decode(output, quantization_channels):
    """ops.py:76-85.  float32 result."""
    mu = np.float32(quantization_channels - 1)
    one = np.float32(1.0)
    casted = np.asarray(output).astype(np.float32)
    signal = (np.float32(2.0) * (casted / mu).astype(np.float32) - one
              ).astype(np.float32)
    # magnitude = (1/mu) * ((1+mu)**|signal| - 1)              ops.py:84
    base = np.float64(one + mu)
    p = np.power(base, np.abs(signal).astype(np.float64)).astype(np.float32)
    magnitude = ((one / mu).astype(np.float32) * (p - one).astype(np.float32)
                 ).astype(np.float32)
    return (np.sign(signal).astype(np.float32) * magnitude).astype(np.float32)


def mu_law_thresholds(quantization_channels):
    """The Q-1 float32 decision thresholds of mu_law_encode: thr[k-1] is the
    smallest float32 x in [-1, 1] with encode(x) >= k.  encode is a monotone
    step function of x, so encode(x) == #{k : thr[k-1] <= x} on [-1, 1].
    Found by bisection over the float32 bit pattern (order-preserving map)."""
    q = quantization_channels

    def to_key(f):
        u = int(np.asarray(f, dtype=np.float32).view(np.uint32))
        return (~u & 0xFFFFFFFF) if (u & 0x80000000) else (u | 0x80000000)

    def from_key(k):
        u = (k & 0x7FFFFFFF) if (k & 0x80000000) else (~k & 0xFFFFFFFF)
        return np.array(u, dtype=np.uint32).view(np.float32)

    lo0 = to_key(np.float32(-1.0))
    hi0 = to_key(np.float32(1.0))
    thr = np.empty(q - 1, dtype=np.float32)
    for k in range(1, q):
        lo, hi = lo0, hi0  # encode(lo) < k <= encode(hi)
        if mu_law_encode(from_key(lo), q) >= k:
            thr[k - 1] = from_key(lo)
            continue
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if mu_law_encode(from_key(mid), q) >= k:
                hi = mid
            else:
                lo = mid
        thr[k - 1] = from_key(hi)
    return thr


# --------------------------------------------------------------------------
# time_to_batch / batch_to_time / causal_conv  (wavenet/ops.py:27-62)
# --------------------------------------------------------------------------


def time_to_batch(value, dilation):
    """ops.py:27-34."""
    b, t, c = value.shape
    pad_elements = dilation - 1 - (t + dilation - 1) % dilation
    padded = np.pad(value, [[0, 0], [0, pad_elements], [0, 0]])
    reshaped = padded.reshape(-1, dilation, c)
    transposed = reshaped.transpose(1, 0, 2)
    return transposed.reshape(b * dilation, -1, c)


def batch_to_time(value, dilation):
    """ops.py:37-43."""
    b, t, c = value.shape
    prepared = value.reshape(dilation, -1, c)
    transposed = prepared.transpose(1, 0, 2)
    return transposed.reshape(b // dilation, -1, c)


def _conv1d_same(x, w):
    """tf.nn.conv1d(x, w, stride=1, padding='SAME'): cross-correlation with
    pad_left=(K-1)//2, pad_right=K-1-pad_left (TF SAME rule)."""
    k = w.shape[0]
    pl = (k - 1) // 2
    pr = k - 1 - pl
    xp = np.pad(x, [[0, 0], [pl, pr], [0, 0]])
    t = x.shape[1]
    out = np.zeros((x.shape[0], t, w.shape[2]), dtype=np.result_type(x, w))
    for kk in range(k):
        out += xp[:, kk:kk + t, :] @ w[kk]
    return out


def causal_conv_literal(value, filter_, dilation):
    """ops.py:46-62 op for op (pad, time_to_batch, conv1d SAME, batch_to_time,
    slice).  Used to validate the closed form below."""
    k = filter_.shape[0]
    padded = np.pad(value, [[0, 0], [(k - 1) * dilation, 0], [0, 0]])
    if dilation > 1:
        transformed = time_to_batch(padded, dilation)
        conv = _conv1d_same(transformed, filter_)
        restored = batch_to_time(conv, dilation)
    else:
        restored = _conv1d_same(padded, filter_)
    return restored[:, :value.shape[1], :]


def causal_conv(value, filter_, dilation):
    """Closed form of ops.py:46-62:
        y[b,t] = sum_k x[b, t - (K-1-k + (K-1)//2) * d] @ W[k],  x[t<0] = 0.
    For K=2: y[t] = x[t-d] @ W[0] + x[t] @ W[1]   (W[0] = past tap,
    consistent with model.py:335-336).  For K>2 TF's SAME centring adds a
    (K-1)//2 * d delay (the reference's behaviour, kept on purpose)."""
    value = np.asarray(value)
    filter_ = np.asarray(filter_)
    k = filter_.shape[0]
    b, t, _ = value.shape
    out = np.zeros((b, t, filter_.shape[2]),
                   dtype=np.result_type(value, filter_))
    for kk in range(k):
        shift = (k - 1 - kk + (k - 1) // 2) * dilation
        if shift >= t:
            continue
        if shift == 0:
            out += value @ filter_[kk]
        else:
            out[:, shift:, :] += value[:, :t - shift, :] @ filter_[kk]
    return out


# --------------------------------------------------------------------------
# Parameters  (wavenet/model.py:7-28, 118-225)
# --------------------------------------------------------------------------


def _xavier(rng, shape, dtype):
    """tf.contrib.layers.xavier_initializer_conv2d (uniform): limit
    sqrt(6/(fan_in+fan_out)), fan_in = prod(shape[:-2])*shape[-2],
    fan_out = prod(shape[:-2])*shape[-1]   (model.py:10) [inferred-TF]."""
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def create_variables(cfg, seed=0, dtype=np.float32, bias_scale=0.0):
    """Same nested dict, names and [K,Cin,Cout] shapes as
    WaveNetModel._create_variables (model.py:118-225).  bias_scale>0 draws
    N(0,bias_scale) biases (the reference zero-inits them, model.py:27; zero
    biases hide bias bugs in parity runs)."""
    rng = np.random.default_rng(seed)
    R, D, S = (cfg['residual_channels'], cfg['dilation_channels'],
               cfg['skip_channels'])
    Q = cfg.get('quantization_channels', 256)
    K = cfg['filter_width']
    G = cfg.get('global_condition_channels')
    card = cfg.get('global_condition_cardinality')
    use_b = cfg.get('use_biases', False)

    def bias(n):
        if bias_scale > 0:
            return (rng.standard_normal(n) * bias_scale).astype(dtype)
        return np.zeros(n, dtype)

    var = {}
    if card is not None:
        if card == G:
            emb = np.identity(card, dtype=dtype)          # model.py:16-19
        else:
            emb = _xavier(rng, (card, G), dtype)
        var['embeddings'] = {'gc_embedding': emb}
    if cfg.get('scalar_input', False):
        c0, k0 = 1, cfg.get('initial_filter_width', 32)
    else:
        c0, k0 = Q, K
    var['causal_layer'] = {'filter': _xavier(rng, (k0, c0, R), dtype)}
    var['dilated_stack'] = []
    for _ in cfg['dilations']:
        cur = {
            'filter': _xavier(rng, (K, R, D), dtype),
            'gate': _xavier(rng, (K, R, D), dtype),
            'dense': _xavier(rng, (1, D, R), dtype),
            'skip': _xavier(rng, (1, D, S), dtype),
        }
        if G is not None:
            cur['gc_gateweights'] = _xavier(rng, (1, G, D), dtype)
            cur['gc_filtweights'] = _xavier(rng, (1, G, D), dtype)
        if use_b:
            cur['filter_bias'] = bias(D)
            cur['gate_bias'] = bias(D)
            cur['dense_bias'] = bias(R)
            cur['skip_bias'] = bias(S)
        var['dilated_stack'].append(cur)
    post = {'postprocess1': _xavier(rng, (1, S, S), dtype),
            'postprocess2': _xavier(rng, (1, S, Q), dtype)}
    if use_b:
        post['postprocess1_bias'] = bias(S)
        post['postprocess2_bias'] = bias(Q)
    var['postprocessing'] = post
    return var


def cast_variables(var, dtype):
    if isinstance(var, dict):
        return {k: cast_variables(v, dtype) for k, v in var.items()}
    if isinstance(var, list):
        return [cast_variables(v, dtype) for v in var]
    return np.asarray(var).astype(dtype)


def zeros_like_variables(var):
    if isinstance(var, dict):
        return {k: zeros_like_variables(v) for k, v in var.items()}
    if isinstance(var, list):
        return [zeros_like_variables(v) for v in var]
    return np.zeros_like(var)


def flatten_variables(var, prefix='wavenet'):
    """(name, array) pairs in the reference's creation order."""
    out = []
    if 'embeddings' in var:
        out.append((prefix + '/embeddings/gc_embedding',
                    var['embeddings']['gc_embedding']))
    out.append((prefix + '/causal_layer/filter',
                var['causal_layer']['filter']))
    order = ['filter', 'gate', 'dense', 'skip', 'gc_gateweights',
             'gc_filtweights', 'filter_bias', 'gate_bias', 'dense_bias',
             'skip_bias']
    for i, cur in enumerate(var['dilated_stack']):
        for k in order:
            if k in cur:
                out.append(('%s/dilated_stack/layer%d/%s' % (prefix, i, k),
                            cur[k]))
    for k in ['postprocess1', 'postprocess2', 'postprocess1_bias',
              'postprocess2_bias']:
        if k in var['postprocessing']:
            out.append((prefix + '/postprocessing/' + k,
                        var['postprocessing'][k]))
    return out


# --------------------------------------------------------------------------
# Forward  (wavenet/model.py:227-330, 389-442, 518-562)
# --------------------------------------------------------------------------


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def one_hot(q, depth, dtype):
    """model.py:518-531; out-of-range index -> all-zero row (tf.one_hot)."""
    q = np.asarray(q)
    out = np.zeros(q.shape + (depth,), dtype=dtype)
    ok = (q >= 0) & (q < depth)
    idx = np.nonzero(ok)
    out[idx + (q[ok],)] = 1
    return out


def embed_gc(cfg, var, gc_ids, batch):
    """model.py:533-562 (integer-id path; the dense-vector branch of the
    reference cannot run: model.py:547,553)."""
    if cfg.get('global_condition_cardinality') is None or gc_ids is None:
        return None
    ids = np.asarray(gc_ids).reshape(-1)
    emb = var['embeddings']['gc_embedding'][ids]
    return emb.reshape(batch, 1, cfg['global_condition_channels'])


def network_forward(cfg, var, net_in, gc_emb=None, keep=False):
    """_create_network (model.py:389-442).  net_in: [B,T,Cin0] (one-hot or
    scalar).  Returns logits [B,T,Q] (+ cache when keep)."""
    dil = cfg['dilations']
    use_b = cfg.get('use_biases', False)
    cache = {'layers': []}
    x = causal_conv(net_in, var['causal_layer']['filter'], 1)  # model.py:400
    total = None
    for i, d in enumerate(dil):
        v = var['dilated_stack'][i]
        last = i == len(dil) - 1
        a_f = causal_conv(x, v['filter'], d)                   # model.py:269
        a_g = causal_conv(x, v['gate'], d)                     # model.py:270
        if gc_emb is not None:                                 # model.py:272-284
            a_f = a_f + gc_emb @ v['gc_filtweights'][0]
            a_g = a_g + gc_emb @ v['gc_gateweights'][0]
        if use_b:                                              # model.py:286-290
            a_f = a_f + v['filter_bias']
            a_g = a_g + v['gate_bias']
        th, sg = np.tanh(a_f), _sigmoid(a_g)
        z = th * sg                                            # model.py:292
        skip = z @ v['skip'][0]                                # model.py:303-305
        if use_b:
            skip = skip + v['skip_bias']                       # model.py:311-312
        total = skip if total is None else total + skip        # model.py:430
        if keep:
            cache['layers'].append({'x': x, 'tanh': th, 'sig': sg, 'z': z,
                                    'skip': skip})
        if not last:                                           # model.py:294-300
            dense = z @ v['dense'][0]
            if use_b:
                dense = dense + v['dense_bias']
            x = x + dense                                      # model.py:330
    p = var['postprocessing']
    h1 = np.maximum(total, 0)                                  # model.py:431
    c1 = h1 @ p['postprocess1'][0]
    if use_b:
        c1 = c1 + p['postprocess1_bias']
    h2 = np.maximum(c1, 0)                                     # model.py:435
    if cfg.get('residual_postproc', False):
        h2 = h2 + total                                        # model.py:436-437
    logits = h2 @ p['postprocess2'][0]
    if use_b:
        logits = logits + p['postprocess2_bias']
    if keep:
        cache.update(total=total, h1=h1, c1=c1, h2=h2, net_in=net_in,
                     gc_emb=gc_emb)
        return logits, cache
    return logits


def _prep_inputs(cfg, var, audio, gc_ids, dtype):
    Q = cfg.get('quantization_channels', 256)
    audio = np.asarray(audio, dtype=np.float32)
    B = cfg['batch_size']
    audio = audio.reshape(B, -1)
    q = mu_law_encode(audio, Q)                                # model.py:639
    enc = one_hot(q, Q, dtype)                                 # model.py:644
    if cfg.get('scalar_input', False):
        net_in = audio.astype(dtype).reshape(B, -1, 1)         # model.py:646-648
    else:
        net_in = enc
    gc_emb = embed_gc(cfg, var, gc_ids, B)
    return q, enc, net_in, gc_emb


def _l2_names(var):
    """Variables the reference's L2 term covers: all trainables whose *name*
    lacks 'bias' (model.py:674-676).  Because create_bias_variable passes the
    name as tf.Variable's 2nd positional arg (= trainable, model.py:28), bias
    variables get default names 'Variable[_n]' and are NOT excluded
    [inferred-TF] -> tf_bias_name_quirk=True includes them."""
    return [n for n, _ in flatten_variables(var)]


def loss(cfg, var, audio, gc_ids=None, l2=None, dtype=np.float32,
         tf_bias_name_quirk=True, keep=False):
    """WaveNetModel.loss (model.py:628-685).  cfg needs 'batch_size'."""
    var = cast_variables(var, dtype)
    q, enc, net_in, gc_emb = _prep_inputs(cfg, var, audio, gc_ids, dtype)
    out = network_forward(cfg, var, net_in, gc_emb, keep=keep)
    logits, cache = out if keep else (out, None)
    B, T, Q = logits.shape
    # targets: one-hot shifted left by one, zero row appended (model.py:657-659)
    shifted = np.concatenate([enc[:, 1:, :], np.zeros((B, 1, Q), dtype)], 1)
    m = logits.max(-1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(-1, keepdims=True))
    logp = logits - lse
    row = -(shifted * logp).sum(-1)                            # model.py:663
    reduced = row.mean()                                       # model.py:666
    l2_loss = None
    total_loss = reduced
    if l2 is not None:
        terms = []
        for n, a in flatten_variables(var):
            if 'bias' in n.split('/')[-1] and not tf_bias_name_quirk:
                continue
            terms.append((a.astype(dtype) ** 2).sum() / 2)     # tf.nn.l2_loss
        l2_loss = sum(terms)
        total_loss = reduced + l2 * l2_loss                    # model.py:679-680
    if keep:
        cache.update(q=q, enc=enc, shifted=shifted, logits=logits, logp=logp,
                     reduced=reduced, l2_loss=l2_loss)
        return total_loss, cache
    return total_loss


# --------------------------------------------------------------------------
# Backward (the reference relies on TF autodiff; restated analytically,
# SURVEY.md section 8a "Backward")
# --------------------------------------------------------------------------


def loss_and_grads(cfg, var, audio, gc_ids=None, l2=None, dtype=np.float32,
                   tf_xent_zero_label_quirk=True, tf_bias_name_quirk=True,
                   relu_masks=None, return_cache=False):
    """Returns (loss, grads) with grads in the same nested layout as var.

    relu_masks: optional {'total': bool [B,T,S], 'c1': bool [B,T,S]} used as
    the ReLU derivative of the two post-processing ReLUs (model.py:431,435)
    instead of this run's own (value > 0).  The gradient is discontinuous in
    the forward values at the ReLU kink: an implementation whose `total` or
    `c1` differs by one rounding from this float64 run may legitimately sit on
    the other side of 0 at a few of the B*T*S positions (float32 numpy vs
    float64 numpy of THIS file differ by 6e-4 of a variable's gradient at
    T=5200 for exactly that reason).  A checker passes the implementation's
    masks here after verifying that they differ from the oracle's only where
    the oracle's value is within rounding of 0.

    tf_xent_zero_label_quirk: TF's fused softmax-xent kernel returns
    backprop = softmax - labels, so the all-zero-label last row of every clip
    back-propagates softmax/(B*T) instead of 0 [inferred-TF]; False gives the
    mathematically exact gradient of the loss value."""
    var = cast_variables(var, dtype)
    total_loss, c = loss(cfg, var, audio, gc_ids, l2, dtype,
                         tf_bias_name_quirk, keep=True)
    dil = cfg['dilations']
    use_b = cfg.get('use_biases', False)
    B, T, Q = c['logits'].shape
    g = zeros_like_variables(var)
    p = np.exp(c['logp'])
    if tf_xent_zero_label_quirk:
        dlogits = (p - c['shifted']) / (B * T)
    else:
        dlogits = (p * c['shifted'].sum(-1, keepdims=True) - c['shifted']
                   ) / (B * T)
    pp = var['postprocessing']
    f2 = lambda a: a.reshape(-1, a.shape[-1])
    g['postprocessing']['postprocess2'][0] = f2(c['h2']).T @ f2(dlogits)
    if use_b:
        g['postprocessing']['postprocess2_bias'] = f2(dlogits).sum(0)
    dh2 = dlogits @ pp['postprocess2'][0].T
    dtotal = np.zeros_like(c['total'])
    if cfg.get('residual_postproc', False):
        dtotal = dtotal + dh2
    m_c1 = (c['c1'] > 0) if relu_masks is None else \
        np.asarray(relu_masks['c1']).reshape(c['c1'].shape)
    m_total = (c['total'] > 0) if relu_masks is None else \
        np.asarray(relu_masks['total']).reshape(c['total'].shape)
    dc1 = dh2 * m_c1
    g['postprocessing']['postprocess1'][0] = f2(c['h1']).T @ f2(dc1)
    if use_b:
        g['postprocessing']['postprocess1_bias'] = f2(dc1).sum(0)
    dh1 = dc1 @ pp['postprocess1'][0].T
    dtotal = dtotal + dh1 * m_total

    K = cfg['filter_width']
    dx = None  # gradient wrt the layer's output x'
    gc_emb = c['gc_emb']
    demb = None if gc_emb is None else np.zeros_like(gc_emb)
    for i in reversed(range(len(dil))):
        d = dil[i]
        v, gv, lc = var['dilated_stack'][i], g['dilated_stack'][i], \
            c['layers'][i]
        last = i == len(dil) - 1
        z, th, sg, x = lc['z'], lc['tanh'], lc['sig'], lc['x']
        gv['skip'][0] = f2(z).T @ f2(dtotal)
        if use_b:
            gv['skip_bias'] = f2(dtotal).sum(0)
        dz = dtotal @ v['skip'][0].T
        if not last:
            gv['dense'][0] = f2(z).T @ f2(dx)
            if use_b:
                gv['dense_bias'] = f2(dx).sum(0)
            dz = dz + dx @ v['dense'][0].T
        da_f = dz * sg * (1 - th * th)
        da_g = dz * th * sg * (1 - sg)
        if use_b:
            gv['filter_bias'] = f2(da_f).sum(0)
            gv['gate_bias'] = f2(da_g).sum(0)
        if gc_emb is not None:
            sf, sgg = da_f.sum(1), da_g.sum(1)               # [B,D]
            e = gc_emb[:, 0, :]                               # [B,G]
            gv['gc_filtweights'][0] = e.T @ sf
            gv['gc_gateweights'][0] = e.T @ sgg
            demb[:, 0, :] += sf @ v['gc_filtweights'][0].T + \
                sgg @ v['gc_gateweights'][0].T
        dxin = np.zeros_like(x) if last else dx.copy()
        for kk in range(K):
            shift = (K - 1 - kk + (K - 1) // 2) * d
            if shift >= T:
                continue
            xs = x[:, :T - shift, :]
            gv['filter'][kk] = f2(xs).T @ f2(da_f[:, shift:, :])
            gv['gate'][kk] = f2(xs).T @ f2(da_g[:, shift:, :])
            dxin[:, :T - shift, :] += da_f[:, shift:, :] @ v['filter'][kk].T \
                + da_g[:, shift:, :] @ v['gate'][kk].T
        dx = dxin
    # causal layer (model.py:227-234), K0 taps, dilation 1
    w0 = var['causal_layer']['filter']
    K0 = w0.shape[0]
    net_in = c['net_in']
    for kk in range(K0):
        shift = (K0 - 1 - kk + (K0 - 1) // 2)
        if shift >= T:
            continue
        g['causal_layer']['filter'][kk] = \
            f2(net_in[:, :T - shift, :]).T @ f2(dx[:, shift:, :])
    if gc_emb is not None:
        ids = np.asarray(gc_ids).reshape(-1)
        np.add.at(g['embeddings']['gc_embedding'], ids, demb[:, 0, :])
    if l2 is not None:
        def add_l2(gv, vv, name):
            if 'bias' in name and not tf_bias_name_quirk:
                return
            gv += l2 * vv
        for (n, ga), (_, va) in zip(flatten_variables(g),
                                    flatten_variables(var)):
            add_l2(ga, va, n.split('/')[-1])
    if return_cache:
        return total_loss, g, c
    return total_loss, g


# --------------------------------------------------------------------------
# Prediction  (wavenet/model.py:564-626)
# --------------------------------------------------------------------------


def _softmax64(logits):
    """softmax in float64 then cast to float32 (model.py:584-585, 620-621)."""
    l = np.asarray(logits, dtype=np.float64)
    e = np.exp(l - l.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


def predict_proba(cfg, var, waveform, gc_ids=None, dtype=np.float32):
    """model.py:564-590: already-quantised int samples -> next-sample
    distribution [Q] for the LAST position."""
    var = cast_variables(var, dtype)
    Q = cfg.get('quantization_channels', 256)
    B = cfg['batch_size']
    w = np.asarray(waveform)
    if cfg.get('scalar_input', False):
        enc = mu_law_decode(w, Q).astype(dtype).reshape(B, -1, 1)
    else:
        enc = one_hot(w.reshape(B, -1), Q, dtype)
    gc_emb = embed_gc(cfg, var, gc_ids, B)
    logits = network_forward(cfg, var, enc, gc_emb)
    return _softmax64(logits.reshape(-1, Q))[-1]


class IncrementalGenerator(object):
    """_create_generator + predict_proba_incremental (model.py:332-387,
    444-516, 592-626): one time step with per-layer FIFO state.  Quirks kept:
    dense is computed for every layer incl. the last (model.py:377-380);
    residual_postproc is ignored here (model.py:511 vs 436-437)."""

    def __init__(self, cfg, var, dtype=np.float32):
        if cfg['filter_width'] > 2:
            raise NotImplementedError("Incremental generation does not "
                                      "support filter_width > 2.")
        if cfg.get('scalar_input', False):
            raise NotImplementedError("Scalar input is not supported by "
                                      "fast generation.")
        self.cfg, self.var, self.dtype = cfg, cast_variables(var, dtype), dtype
        self.init_ops()

    def init_ops(self):
        """net.init_ops: queues pre-filled with zeros (model.py:457-458,
        477-479)."""
        cfg = self.cfg
        B = cfg['batch_size']
        Q = cfg.get('quantization_channels', 256)
        self.q0 = [np.zeros((B, Q), self.dtype)]
        self.queues = [[np.zeros((B, cfg['residual_channels']), self.dtype)
                        for _ in range(d)] for d in cfg['dilations']]

    def step(self, sample, gc_ids=None, push=True):
        cfg, var = self.cfg, self.var
        Q = cfg.get('quantization_channels', 256)
        B = cfg['batch_size']
        use_b = cfg.get('use_biases', False)
        cur = one_hot(np.asarray(sample).reshape(-1), Q, self.dtype)
        cur = cur.reshape(-1, Q)
        gc_emb = embed_gc(cfg, var, gc_ids, B)
        pushes = []
        state = self.q0[0]
        pushes.append((self.q0, cur))
        w = var['causal_layer']['filter']
        x = state @ w[0] + cur @ w[1]                          # model.py:335-338
        total = None
        for i, d in enumerate(cfg['dilations']):
            v = var['dilated_stack'][i]
            state = self.queues[i][0]
            pushes.append((self.queues[i], x))
            a_f = state @ v['filter'][0] + x @ v['filter'][1]
            a_g = state @ v['gate'][0] + x @ v['gate'][1]
            if gc_emb is not None:
                e = gc_emb.reshape(1, -1)                      # model.py:360-361
                a_f = a_f + e @ v['gc_filtweights'][0]
                a_g = a_g + e @ v['gc_gateweights'][0]
            if use_b:
                a_f = a_f + v['filter_bias']
                a_g = a_g + v['gate_bias']
            z = np.tanh(a_f) * _sigmoid(a_g)
            dense = z @ v['dense'][0]
            if use_b:
                dense = dense + v['dense_bias']
            skip = z @ v['skip'][0]
            if use_b:
                skip = skip + v['skip_bias']
            total = skip if total is None else total + skip
            x = x + dense
        p = var['postprocessing']
        c1 = np.maximum(total, 0) @ p['postprocess1'][0]
        if use_b:
            c1 = c1 + p['postprocess1_bias']
        logits = np.maximum(c1, 0) @ p['postprocess2'][0]
        if use_b:
            logits = logits + p['postprocess2_bias']
        if push:                                               # net.push_ops
            for qu, val in pushes:
                qu.pop(0)
                qu.append(val)
        return _softmax64(logits.reshape(-1, Q))[-1]


# --------------------------------------------------------------------------
# Optimizers with TensorFlow-0.10 update rules  (wavenet/ops.py:6-24)
# [inferred-TF]: they differ from torch.optim.
# --------------------------------------------------------------------------


class TFOptimizer(object):
    def __init__(self, kind, learning_rate, momentum=0.9):
        self.kind, self.lr, self.mom = kind, learning_rate, momentum
        self.t = 0
        self.slots = None

    def apply(self, w, g):
        """w, g: flat float arrays (updated copy of w is returned)."""
        dt = w.dtype.type
        if self.slots is None:
            if self.kind == 'adam':
                self.slots = [np.zeros_like(w), np.zeros_like(w)]
            elif self.kind == 'sgd':
                self.slots = [np.zeros_like(w)]
            else:  # rmsprop: ms initialised to ONE, mom to zero
                self.slots = [np.ones_like(w), np.zeros_like(w)]
        self.t += 1
        lr = dt(self.lr)
        if self.kind == 'adam':                                # ops.py:6-8
            b1, b2, eps = dt(0.9), dt(0.999), dt(1e-4)
            m, v = self.slots
            m[:] = b1 * m + (dt(1) - b1) * g
            v[:] = b2 * v + (dt(1) - b2) * g * g
            lr_t = dt(self.lr * math.sqrt(1 - 0.999 ** self.t) /
                      (1 - 0.9 ** self.t))
            return w - lr_t * m / (np.sqrt(v) + eps)
        if self.kind == 'sgd':                                 # ops.py:11-13
            acc, = self.slots
            acc[:] = dt(self.mom) * acc + g
            return w - lr * acc
        if self.kind == 'rmsprop':                             # ops.py:16-19
            ms, mom = self.slots
            rho, eps = dt(0.9), dt(1e-5)
            ms[:] = rho * ms + (dt(1) - rho) * g * g
            mom[:] = dt(self.mom) * mom + lr * g / np.sqrt(ms + eps)
            return w - mom
        raise KeyError(self.kind)


def pack(var):
    return np.concatenate([a.reshape(-1) for _, a in flatten_variables(var)])


def unpack_into(var, flat):
    off = 0
    for _, a in flatten_variables(var):
        n = a.size
        a[...] = flat[off:off + n].reshape(a.shape)
        off += n
    return var
