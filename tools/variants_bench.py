"""Step time of the off-default model variants at the bench shape,
to spot a variant whose extra kernels cost more than they should."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import torch
from bench import synth_audio
from wavenet import WaveNetModel, optimizer_factory

p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
base = dict(batch_size=8, dilations=p['dilations'], filter_width=2, residual_channels=32,
            dilation_channels=32, skip_channels=512, quantization_channels=256, use_biases=True)
audio = torch.from_numpy(synth_audio(8, 16000)).cuda()
ids = torch.arange(8, dtype=torch.int32).cuda()
cases = [('default', {}, None, None),
         ('l2', {}, None, 1e-4),
         ('gc 32x377', dict(global_condition_channels=32, global_condition_cardinality=377), ids, None),
         ('residual_postproc', dict(residual_postproc=True), None, None),
         ('scalar_input k32', dict(scalar_input=True, initial_filter_width=32), None, None),
         ('no biases', dict(use_biases=False), None, None),
         ('filter_width 3', dict(filter_width=3), None, None),
         ('64 channels', dict(residual_channels=64, dilation_channels=64), None, None),
         ('128 channels', dict(residual_channels=128, dilation_channels=128), None, None),
         ('256 channels', dict(residual_channels=256, dilation_channels=256), None, None),
         ('momentum', {}, None, None), ('rmsprop', {}, None, None)]
only = os.environ.get('KB_ONLY')
if only:
    cases = [c for c in cases if only in c[0]]
for name, kw, gc, l2 in cases:
    cfg = dict(base); cfg.update(kw)
    net = WaveNetModel(seed=0, **cfg)
    optname = {'momentum': 'sgd', 'rmsprop': 'rmsprop'}.get(name, 'adam')
    opt = optimizer_factory[optname](learning_rate=1e-3, momentum=0.9)
    for _ in range(4):
        opt.minimize(net.loss(audio, gc, l2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        opt.minimize(net.loss(audio, gc, l2))
    torch.cuda.synchronize()
    print('%-20s %7.2f ms/step' % (name, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    del net
    torch.cuda.empty_cache()
