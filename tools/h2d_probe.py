"""Where a train.py iteration's host time goes (one GPU): what a pageable /
pinned / device-resident input costs per step, and the pieces of an iteration
(reader, staging copy, net.loss / minimize issue, the delayed loss fetch) with
a 2.2 ms host pause, the synthetic reader's numpy work, and the reader itself.
Round 6 findings: float(loss) waits for everything queued on the stream -- the
NEXT step included -- so a loss read one step late still serialises host and
device (12.7 ms per iteration) unless it goes through an event + pinned scalar
(train.py: 9.52 ms); H2D from a pinned buffer costs 33 ms per step here,
pageable 9.6.
    python tools/h2d_probe.py"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import train  # noqa: E402
from wavenet import WaveNetModel, optimizer_factory  # noqa: E402
from util import model_kwargs, synth_audio  # noqa: E402

p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
cfg = {k: p[k] for k in p if k != 'sample_rate'}
cfg['batch_size'] = 8
net = WaveNetModel(seed=0, **model_kwargs(cfg))
opt = optimizer_factory['adam'](learning_rate=1e-3, momentum=0.9)
host = torch.from_numpy(synth_audio(8, 16000))
pin = torch.empty(host.shape).pin_memory()
dev = host.cuda()
cs = torch.cuda.Stream()


def run(mode, n=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prev = None
    for i in range(n):
        if mode == 'pageable':
            a = host
        elif mode == 'pinned_async':
            pin.copy_(host)
            a = pin.to('cuda', non_blocking=True)
        elif mode == 'copy_stream':
            with torch.cuda.stream(cs):
                a = host.to('cuda')
            torch.cuda.current_stream().wait_stream(cs)
        else:
            a = dev
        loss = net.loss(a)
        opt.minimize(loss)
        if prev is not None:
            float(prev)
        prev = loss
    float(prev)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for m in ('device', 'pageable', 'copy_stream', 'pinned_async', 'device'):
    run(m, 5)
    print('%-13s %.2f ms/step (loss fetched one step late)' % (m, run(m)))

def pieces(kind):
    rd = train.SyntheticReader(16000)
    T = dict(dequeue=0.0, stage=0.0, loss=0.0, minimize=0.0, fetch=0.0)
    prev = None
    n = 60
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for i in range(n):
        t = time.perf_counter()
        if kind == 'sleep':
            time.sleep(0.0022)
            a = host
        elif kind == 'numpy_unused':
            rd.dequeue(8)
            a = host
        else:
            a = rd.dequeue(8)
        T['dequeue'] += time.perf_counter() - t
        t = time.perf_counter()
        with torch.cuda.stream(cs):
            d = a.reshape(8, -1).contiguous().to('cuda')
        torch.cuda.current_stream().wait_stream(cs)
        T['stage'] += time.perf_counter() - t
        t = time.perf_counter()
        loss = net.loss(d)
        T['loss'] += time.perf_counter() - t
        t = time.perf_counter()
        opt.minimize(loss)
        T['minimize'] += time.perf_counter() - t
        t = time.perf_counter()
        if prev is not None:
            float(prev)
        prev = loss
        T['fetch'] += time.perf_counter() - t
    torch.cuda.synchronize()
    print('%-13s iteration %.2f ms: ' % (kind, (time.perf_counter() - t_all) / n * 1e3) +
          ', '.join('%s %.2f' % (k, v / n * 1e3) for k, v in T.items()))


for kind in ('sleep', 'numpy_unused', 'reader', 'sleep'):
    pieces(kind)
