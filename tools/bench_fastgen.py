"""Secondary metric (BASELINE.json config[4]): fast generation of 16000
samples, batch 1, default wavenet_params.json stack, one persistent kernel.
Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
import torch
from wavenet import WaveNetModel

p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
net = WaveNetModel(batch_size=1, dilations=p['dilations'], filter_width=p['filter_width'],
                   residual_channels=p['residual_channels'], dilation_channels=p['dilation_channels'],
                   skip_channels=p['skip_channels'], quantization_channels=p['quantization_channels'],
                   use_biases=p['use_biases'], seed=0)
net.generate(200, seed_samples=[128], seed=1)       # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
out = net.generate(n, seed_samples=[128], temperature=1.0, seed=2)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({'metric': 'fast-generation audio samples/sec (batch 1, default stack)',
                  'value': n / dt, 'unit': 'audio samples/s', 'us_per_sample': dt / n * 1e6,
                  'samples': n, 'seconds': dt,
                  'distinct_codes': int(torch.unique(out).numel())}))
