"""One clip of 16000 samples (the reference's own batch size) on the default
stack: N training steps, for a kernel trace of the small-batch step.
    rocprofv3 --kernel-trace --stats -d out -- python tools/b1_step.py [steps] [clips]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import WaveNetModel, optimizer_factory  # noqa: E402
from util import model_kwargs  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
cfg = {k: p[k] for k in p if k != 'sample_rate'}
cfg['batch_size'] = B
if os.environ.get('KB_CH'):     # more than 32 channels: the channel-block path
    cfg['residual_channels'] = cfg['dilation_channels'] = int(os.environ['KB_CH'])
net = WaveNetModel(seed=0, **model_kwargs(cfg))
if os.environ.get('KB_SPLIT_FRAC'):
    net.overlap_tn_split_frac = float(os.environ['KB_SPLIT_FRAC'])
if os.environ.get('KB_OVERLAP_TN'):
    net.overlap_tn = os.environ['KB_OVERLAP_TN'] == '1'
opt = optimizer_factory['adam'](learning_rate=1e-3, momentum=0.9)
audio = torch.from_numpy(np.random.default_rng(0).uniform(
    -1, 1, (B, 16000)).astype(np.float32)).cuda()
for _ in range(5):
    opt.minimize(net.loss(audio))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    opt.minimize(net.loss(audio))
torch.cuda.synchronize()
print('B=%d overlap_tn=%s: %.3f ms per step' % (B, net.overlap_tn, (time.perf_counter() - t0) / steps * 1e3))
