"""Experiment: one training step (net.loss + optimizer.minimize) captured in a
hipGraph and replayed, against the launch-plan replay the model does by itself.
    python tools/graph_step.py [B] [T]
(timing only: Adam's bias-corrected rate is a by-value argument, frozen in the
graph -- the numbers say what a graph-safe optimizer step would gain)"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
cfg = {k: p[k] for k in p if k != 'sample_rate'}
cfg['batch_size'] = B
net = WaveNetModel(seed=0, **model_kwargs(cfg))
opt = optimizer_factory['adam'](learning_rate=1e-3, momentum=0.9)
audio = torch.from_numpy(synth_audio(B, T)).cuda()


def timed(fn, n=20, rounds=3):
    ts = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2], ts


def step():
    opt.minimize(net.loss(audio))


for _ in range(5):
    step()
plan_ms, plan_all = timed(step)
# capture: a side stream warm-up as torch asks for, then the graph
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
for _ in range(3):
    g.replay()
graph_ms, graph_all = timed(g.replay)
plan2_ms, plan2_all = timed(step)
print('B %d T %d: launch plan %.3f ms (%s)  hipGraph replay %.3f ms (%s)  launch plan again %.3f ms' % (
    B, T, plan_ms, ' '.join('%.3f' % t for t in plan_all), graph_ms,
    ' '.join('%.3f' % t for t in graph_all), plan2_ms))
