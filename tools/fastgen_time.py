"""Fast-generation rate of the default stack (one process, one library):
    WN_LIB_PATH=build/ab/lib_new.so python tools/fastgen_time.py [samples]
Prints us per sample (median of 3 runs) and a checksum of the drawn samples."""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import WaveNetModel  # noqa: E402
from util import model_kwargs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
cfg = {k: p[k] for k in p if k != 'sample_rate'}
cfg['batch_size'] = 1
if os.environ.get('KB_CH'):     # more than 32 channels: the wide generator
    cfg['residual_channels'] = cfg['dilation_channels'] = int(os.environ['KB_CH'])
gen = WaveNetModel(seed=0, **model_kwargs(cfg))
gen.generate(200, seed_samples=[128], seed=1)
torch.cuda.synchronize()
ts = []
for r in range(3):
    t0 = time.perf_counter()
    out = gen.generate(n, seed_samples=[128], temperature=1.0, seed=2)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / n * 1e6)
out = out.cpu().numpy() if hasattr(out, 'cpu') else np.asarray(out)
print('%s: %.2f us/sample (runs %s)  checksum %d' % (
    os.environ.get('WN_LIB_PATH', 'default lib'), float(np.median(ts)),
    ' '.join('%.2f' % t for t in ts), int(out.astype(np.int64).sum())))
