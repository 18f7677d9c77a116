"""BASELINE.json configs[4] (fast generation on the DEFAULT stack: L=50,
S=512, Q=256, dilations up to 512) checked for correctness, not only speed:
the device paths of WaveNetModel.generate / predict_proba_incremental against
the float64 restatement of the reference's generator
(oracle.IncrementalGenerator = model.py:332-387, 444-516, 592-626;
teacher-forced like test/test_generation.py:50-72, but over > 2 wraps of the
dilation-512 ring buffers and past the receptive field 5117)."""
import json
import os

import numpy as np
import pytest
import torch

from util import O, ROOT, cfg_with, build_pair

pytestmark = pytest.mark.gpu
TOL = 1e-5            # probabilities, float32 device path vs float64 oracle
N_TRACE = 1200        # > 2 * 512 (two wraps of the longest ring) + graph replay
N_SEED = 6000         # > receptive field 5117


def default_cfg(**kw):
    p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
    c = {k: p[k] for k in p if k != 'sample_rate'}
    c['batch_size'] = 1
    c.update(kw)
    return c


def _wave(n, seed):
    """Teacher-forcing codes that exercise all bins: a mu-law-coded sine plus
    uniformly random codes."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = 0.6 * np.sin(2 * np.pi * 220.0 * t / 16000.0)
    w = O.mu_law_encode(x.astype(np.float32), 256)
    r = rng.integers(0, 256, n)
    return np.where(rng.uniform(size=n) < 0.5, w, r).astype(np.int32)


@pytest.fixture(scope='module', params=['plain', 'gc'])
def traced(request, hip_lib):
    """(net, cfg, var, wave, gc id, oracle probabilities of the first N_TRACE
    steps, oracle generator advanced over the whole N_SEED-sample seed)."""
    gc = request.param == 'gc'
    cfg = default_cfg(**(dict(global_condition_channels=32,
                              global_condition_cardinality=377) if gc else {}))
    net, var = build_pair(cfg)                 # biases ~ N(0, 0.1)
    gid = 123 if gc else None
    wave = _wave(N_SEED, 11)
    gen = O.IncrementalGenerator(cfg, var, dtype=np.float64)
    ids = None if gid is None else np.array([gid])
    probs = np.empty((N_SEED, 256), np.float32)
    for i, s in enumerate(wave):
        probs[i] = gen.step(int(s), ids)
    return net, cfg, var, wave, gid, probs, gen


@pytest.mark.parametrize('multi,persist', [(False, False), (True, False), (True, True)],
                         ids=['one_wg', 'multi_cu_graph', 'multi_cu_persistent'])
def test_default_stack_teacher_forced_trace(traced, multi, persist):
    """Every one of N_TRACE next-sample distributions of the three device paths
    (single-workgroup kernel; multi-CU step kernels replayed from a hipGraph;
    ONE persistent multi-CU launch for the whole run) equals the oracle's."""
    net, cfg, var, wave, gid, probs, _ = traced
    net.fastgen_multi_cu = multi
    net.fastgen_persistent = persist
    net.fastgen_graph_steps = 200              # 1199 steps -> 4 graph replays
    seed = wave[:N_TRACE]
    out, pr = net.generate(0, seed_samples=seed, return_proba_every=1,
                           global_condition=gid)
    pr = pr.cpu().numpy()
    assert np.array_equal(out.cpu().numpy(), seed)
    assert pr.shape == (N_TRACE - 1, 256)
    err = np.abs(pr - probs[:N_TRACE - 1]).max(axis=1)
    assert err.max() < TOL, (int(err.argmax()), float(err.max()))
    # the ring buffers wrapped: steps past 2 * 512 are covered
    assert err[1024:].max() < TOL


def test_default_stack_incremental_api_and_naive(traced):
    """predict_proba_incremental one sample at a time (the reference's calling
    pattern, generate.py:213-226) over 1100 steps, then the naive forward
    (model.py:564-590) on the same history: all equal to the oracle."""
    net, cfg, var, wave, gid, probs, _ = traced
    net.reset_generator()
    n = 1100
    worst = 0.0
    for i in range(n):
        p = net.predict_proba_incremental(int(wave[i]), global_condition=gid)
        if i % 50 == 0 or i >= n - 40:
            worst = max(worst, float(np.abs(p.cpu().numpy() - probs[i]).max()))
    assert worst < TOL
    # naive path over the full seed (longer than the receptive field, so it
    # sees exactly what the queues hold)
    naive = net.predict_proba(wave, None if gid is None else [gid])
    assert np.abs(naive.cpu().numpy() - probs[N_SEED - 1]).max() < TOL


def test_default_stack_prime_from_6000_sample_seed(traced):
    """prime_generator (ONE forward pass over a seed longer than the receptive
    field; the reference's TODO generate.py:199-201) then one incremental
    step: equals the oracle generator that consumed the seed sample by
    sample; and the sampling path continues from it."""
    net, cfg, var, wave, gid, probs, gen = traced
    net.prime_generator(wave[:N_SEED - 1], global_condition=gid)
    p = net.predict_proba_incremental(int(wave[N_SEED - 1]),
                                      global_condition=gid, push=False)
    assert np.abs(p.cpu().numpy() - probs[N_SEED - 1]).max() < TOL
    # generate() from the long seed: the first drawn sample's distribution is
    # probs[N_SEED-1]; teacher-force the drawn continuation through the oracle
    # and require it to be likely under the oracle's distributions
    for multi, persist in ((False, False), (True, False), (True, True)):
        net.fastgen_multi_cu = multi
        net.fastgen_persistent = persist
        net.fastgen_graph_steps = 50
        out = net.generate(120, seed_samples=wave, seed=5,
                           global_condition=gid).cpu().numpy()
        assert out.shape == (N_SEED + 120,)
        assert np.array_equal(out[:N_SEED], wave)
        g2 = _clone_generator(gen)
        ids = None if gid is None else np.array([gid])
        ll = np.log(probs[N_SEED - 1][out[N_SEED]] + 1e-30)
        for i in range(119):
            pi = g2.step(int(out[N_SEED + i]), ids)
            ll += np.log(pi[out[N_SEED + i + 1]] + 1e-30)
        # random-init network: near-uniform predictions; a drawn sequence
        # cannot be much less likely than uniform under the true distribution
        assert ll / 120 > np.log(1.0 / 256) - 1.0


def _clone_generator(gen):
    import copy
    g = copy.copy(gen)
    g.q0 = [a.copy() for a in gen.q0]
    g.queues = [[a.copy() for a in q] for q in gen.queues]
    return g


def test_persistent_launch_failure_restores_state_and_falls_back(hip_lib, monkeypatch):
    """A persistent generation launch whose workgroups cannot all be resident
    (CUs held by another process: invisible to the launch-time occupancy check)
    ends with an expired hand-over wait after it has rewritten part of the
    generator's state.  The host must restore its snapshot, warn, and finish
    the run on the step kernels: same samples as a model that took the step
    kernels from the start, also on the following call."""
    from wavenet import _lib
    cfg = default_cfg()
    a, _ = build_pair(cfg)
    b, _ = build_pair(cfg)
    a.fastgen_graph_steps = b.fastgen_graph_steps = 40
    b.fastgen_persistent = False
    lib = _lib.load()
    real = lib.wn_fastgen_persist
    calls = []

    def failing(*args):
        # the real launch runs (state, cursors, codes are rewritten), then
        # reports what an expired wait reports
        code = real(*args)
        torch.cuda.synchronize()
        a._gen['fgp_sync'][12] = 1
        calls.append(code)
        return code
    monkeypatch.setattr(lib, 'wn_fastgen_persist', failing)
    seed = _wave(700, 3)
    with pytest.warns(UserWarning, match='state restored'):
        out_a = a.generate(150, seed_samples=seed, seed=9).cpu().numpy()
    out_b = b.generate(150, seed_samples=seed, seed=9).cpu().numpy()
    assert calls == [0] and a._gen_launch_failed['persist']
    assert np.array_equal(out_a, out_b)
    assert a._gen['steps'] == b._gen['steps']
    more_a = a.continue_generation(90, int(out_a[-1]), seed=4).cpu().numpy()
    more_b = b.continue_generation(90, int(out_b[-1]), seed=4).cpu().numpy()
    assert calls == [0]                     # not tried again on this generator
    assert np.array_equal(more_a, more_b)
    assert torch.equal(a._gen['state'], b._gen['state'])
