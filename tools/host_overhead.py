"""Host time to ISSUE one training step (no device sync) vs its GPU time:
with launch plans (default) and with the eager Python path."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import torch
from bench import synth_audio
from wavenet import WaveNetModel, optimizer_factory

p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
net = WaveNetModel(batch_size=8, dilations=p['dilations'], filter_width=p['filter_width'],
                   residual_channels=p['residual_channels'], dilation_channels=p['dilation_channels'],
                   skip_channels=p['skip_channels'], quantization_channels=p['quantization_channels'],
                   use_biases=p['use_biases'], seed=0)
opt = optimizer_factory['adam'](learning_rate=1e-3, momentum=0.9)
audio = torch.from_numpy(synth_audio(8, 16000)).cuda()
for plans in (True, False, True):
    net.use_launch_plans = plans
    for _ in range(4):
        opt.minimize(net.loss(audio))
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        opt.minimize(net.loss(audio))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('launch plans %-5s: host issue %.2f ms/step, wall %.2f ms/step' %
          (plans, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
