"""Hash of the loss and the whole gradient bucket of one training step of the
default stack (for bitwise A/B of two library builds in separate processes):
    WN_LIB_PATH=... python tools/grad_hash.py [B] [T] [variant]"""
import hashlib
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
from util import model_kwargs, synth_audio  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
cfg = {k: p[k] for k in p if k != 'sample_rate'}
cfg['batch_size'] = B
if len(sys.argv) > 3:
    WaveNetModel.DEFAULT_STACK_VARIANT = int(sys.argv[3], 0)
net = WaveNetModel(seed=0, **model_kwargs(cfg))
audio = synth_audio(B, T)
loss = float(net.loss(audio))
g = net.grads.cpu().numpy()
h = hashlib.sha256(g.tobytes()).hexdigest()[:16]
opt = optimizer_factory['adam'](learning_rate=1e-3, momentum=0.9)
for _ in range(5):
    opt.minimize(net.loss(audio))
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        opt.minimize(net.loss(audio))
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 10 * 1e3)
print('%s: B %d T %d loss %.9g grads sha %s abssum %.9g  step %.3f ms (runs %s)' % (
    os.path.basename(os.environ.get('WN_LIB_PATH', 'default lib')), B, T, loss, h,
    float(np.abs(g).sum(dtype=np.float64)), sorted(ts)[1], ' '.join('%.3f' % t for t in ts)))
