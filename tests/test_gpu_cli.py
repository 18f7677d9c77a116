"""End-to-end: train.py (synthetic + wav data, checkpoint / resume) and
generate.py (fast + naive, wav seed, save_every) on the GPU."""
import json
import os
import sys

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from util import ROOT

sys.path.insert(0, ROOT)
import train  # noqa: E402
import generate  # noqa: E402
from test_audio_reader import make_wavs  # noqa: E402

pytestmark = pytest.mark.gpu

SMALL = {"filter_width": 2, "sample_rate": 16000,
         "dilations": [1, 2, 4, 8, 16, 32, 1, 2, 4, 8, 16, 32],
         "residual_channels": 32, "dilation_channels": 32,
         "quantization_channels": 256, "skip_channels": 64,
         "use_biases": True, "scalar_input": False,
         "initial_filter_width": 32, "residual_postproc": False}


@pytest.fixture()
def params(tmp_path):
    p = str(tmp_path / 'params.json')
    json.dump(SMALL, open(p, 'w'))
    return p


def test_train_checkpoint_resume_and_generate(hip_lib, tmp_path, params, capsys):
    logdir = str(tmp_path / 'run')
    common = ['--synthetic', '--sample_size', '3000', '--batch_size', '2',
              '--wavenet_params', params, '--logdir', logdir,
              '--checkpoint_every', '2', '--learning_rate', '0.002']
    assert train.main(common + ['--num_steps', '5']) == 0
    out = capsys.readouterr().out
    assert 'step 0 - loss = ' in out and 'sec/step)' in out
    assert os.path.exists(os.path.join(logdir, 'model.ckpt-4'))
    assert train.latest_checkpoint(logdir).endswith('model.ckpt-4')
    ev = [json.loads(l) for l in open(os.path.join(logdir, 'events.jsonl'))]
    assert [e['step'] for e in ev] == [0, 1, 2, 3, 4]
    assert ev[-1]['loss'] < ev[0]['loss']
    # resume: continues at step 5 from the restored weights
    assert train.main(common + ['--num_steps', '7']) == 0
    out = capsys.readouterr().out
    assert 'Global step was: 4' in out and 'step 5 - loss' in out
    assert 'step 4 - loss' not in out
    ck = os.path.join(logdir, 'model.ckpt-6')
    assert os.path.exists(ck)
    sd = torch.load(ck, map_location='cpu')
    assert sd['step'] == 6
    assert 'wavenet/dilated_stack/layer3/filter' in sd['variables']
    # generate: fast path with save_every chunks
    wav = str(tmp_path / 'out.wav')
    assert generate.main([ck, '--samples', '300', '--wavenet_params', params,
                          '--wav_out_path', wav, '--save_every', '120',
                          '--logdir', str(tmp_path / 'gen')]) == 0
    rate, data = wavfile.read(wav)
    assert rate == 16000 and data.shape[0] == 301 and np.abs(data).max() <= 1
    # seeded by a wav file, naive windowed path
    seed = str(tmp_path / 'seed.wav')
    t = np.arange(4000) / 16000.0
    wavfile.write(seed, 16000, (0.4 * np.sin(2 * np.pi * 330 * t)
                                ).astype(np.float32))
    wav2 = str(tmp_path / 'out2.wav')
    assert generate.main([ck, '--samples', '20', '--wavenet_params', params,
                          '--wav_out_path', wav2, '--wav_seed', seed,
                          '--window', '200', '--fast_generation', 'false',
                          '--logdir', str(tmp_path / 'gen')]) == 0
    _, d2 = wavfile.read(wav2)
    assert d2.shape[0] == 200 + 20
    # and seeded fast path (priming in-kernel)
    assert generate.main([ck, '--samples', '50', '--wavenet_params', params,
                          '--wav_out_path', wav2, '--wav_seed', seed,
                          '--window', '500',
                          '--logdir', str(tmp_path / 'gen')]) == 0
    _, d3 = wavfile.read(wav2)
    assert d3.shape[0] == 500 + 50


def test_chunked_generation_equals_single_call(hip_lib):
    from wavenet import WaveNetModel
    kw = {k: SMALL[k] for k in SMALL if k not in ('sample_rate',)}
    net = WaveNetModel(batch_size=1, seed=4, **kw)
    net.fastgen_graph_steps = 40
    a = net.generate(200, seed_samples=[7, 9], seed=3).cpu().numpy()
    b1 = net.generate(80, seed_samples=[7, 9], seed=3).cpu().numpy()
    b2 = net.continue_generation(120, int(b1[-1]), 1.0, None, 3).cpu().numpy()
    assert np.array_equal(a, np.concatenate([b1, b2]))


def test_train_on_wav_directory_with_gc(hip_lib, tmp_path, params, capsys):
    data = str(tmp_path / 'corpus')
    os.makedirs(data)
    make_wavs(data)
    logdir = str(tmp_path / 'run2')
    assert train.main(['--data_dir', data, '--sample_size', '4000',
                       '--batch_size', '2', '--wavenet_params', params,
                       '--logdir', logdir, '--num_steps', '3',
                       '--gc_channels', '8', '--silence_threshold', '0.05',
                       '--optimizer', 'sgd', '--learning_rate', '0.01',
                       '--l2_regularization_strength', '1e-6']) == 0
    out = capsys.readouterr().out
    assert 'Detected --gc_cardinality=227' in out
    assert 'step 2 - loss' in out
    sd = torch.load(train.latest_checkpoint(logdir), map_location='cpu')
    assert tuple(sd['variables']['wavenet/embeddings/gc_embedding'].shape) \
        == (227, 8)


def test_train_histogram_summaries(hip_lib, tmp_path, params):
    """--histograms True: the reference's per-layer weight histograms
    (model.py:314-325 tags) are written next to each checkpoint."""
    logdir = str(tmp_path / 'run_h')
    assert train.main(['--synthetic', '--sample_size', '2000', '--batch_size',
                       '1', '--wavenet_params', params, '--logdir', logdir,
                       '--checkpoint_every', '2', '--num_steps', '3',
                       '--histograms', 'True']) == 0
    z = np.load(os.path.join(logdir, 'histograms-2.npz'))
    L = len(SMALL['dilations'])
    for tag in ('layer0_filter', 'layer0_gate', 'layer0_dense', 'layer0_skip',
                'layer3_biases_filter', 'layer3_biases_skip',
                'layer%d_skip' % (L - 1)):
        c = z[tag + '/counts']
        assert c.shape == (30,) and c.sum() > 0
        lo, hi = z[tag + '/range']
        assert hi > lo
    assert 'layer%d_dense/counts' % (L - 1) not in z.files   # model.py:318
    assert z['layer0_filter/counts'].sum() == 2 * 32 * 32


def test_bench_line_carries_the_contract(hip_lib):
    """`python bench.py` (small shape): ONE JSON line with the driver's keys,
    the roofline object measured live (NN GEMMs + the TN GEMMs beside them),
    the whole-step fraction, the CPU baseline with its core count and CPU
    model, and the secondary figures (GC step, B = 1 step, fast generation)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, WN_CPU_THREADS='4')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'),
                        '--steps', '2', '--warmup', '1', '--batch', '2',
                        '--samples', '4000'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup',
              'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in r, k
    assert r['n_gpus'] == 1 and r['steps'] == 2 and r['dtype'] == 'f32'
    assert r['vs_baseline'] is None and 'workload' in r['config']
    rf = r['roofline']
    assert rf['bound'] == 'mfma' and rf['peak'] == 157.3
    assert 0 < rf['frac'] < 1 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    # (five NN GEMMs at this small shape: the skip sum runs inside the forward
    # stack launch, wn_stack_fwd_skip; six at the headline shape)
    assert rf['launches_per_step'] == 5 and rf['tn_gemms']['launches_per_step'] == 3
    # the two persistent stack launches are timed live too; their bound is
    # issue (f32 MFMA + vector ALU: one resource), the PMC figures are quoted
    # at the default shape only, and no fraction of a "bound" that can exceed 1
    sl = rf['stack_launches']
    assert sl['wn_stack_fwd']['avg_launch_us'] > 0 and sl['wn_stack_bwd']['avg_launch_us'] > 0
    assert sl['wn_stack_bwd']['bound'] == 'issue' and sl['wn_stack_bwd']['traffic'] is None
    assert 0 < sl['wn_stack_bwd']['mfma_busy_frac_live_at_2p4ghz'] < 1
    assert not any('frac_of_mix_stream' in v for v in sl.values())
    assert 'traffic_stale' in rf
    assert 0 < r['step_frac'] < 1
    cb = r['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['cpu_model']
    assert cb['value'] > 0 and r['gpu_over_cpu'] > 1
    sec = r['secondary']
    for k in ('forward_only_samples_per_s', 'fastgen_us_per_sample',
              'gc_ms_per_step', 'b1_ms_per_step'):
        assert sec[k] > 0, k
