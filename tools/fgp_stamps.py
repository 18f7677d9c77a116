"""In-kernel stamps of fg_persist_kernel (diagnostic build:
tools/mk_variants.sh wn_fastgen.hip stamps:-DFGP_STAMPS;
WN_LIB_PATH=.../build/ab/lib_stamps.so python tools/fgp_stamps.py [steps]):
where a sample's time goes between the roles of the persistent launch."""
import ctypes
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import WaveNetModel, _lib  # noqa: E402
from util import model_kwargs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
p = json.load(open(os.path.join(ROOT, 'wavenet_params.json')))
cfg = {k: p[k] for k in p if k != 'sample_rate'}
cfg['batch_size'] = 1
gen = WaveNetModel(seed=0, **model_kwargs(cfg))
gen.fastgen_persistent = True
lib = _lib.load()
dbg = torch.zeros((n + 8) * 16, dtype=torch.int64, device='cuda')
lib.wn_diag_fgp_dbg.argtypes = [ctypes.c_void_p]
lib.wn_diag_fgp_dbg(dbg.data_ptr())
gen.generate(200, seed_samples=[128], seed=1)
dbg.zero_()
gen.generate(n, seed_samples=[128], temperature=1.0, seed=2)
torch.cuda.synchronize()
s = dbg.cpu().numpy().reshape(-1, 16)[:n].astype(np.float64) / 100.0   # us
s = s[5:n - 2]
nseg = 5
period = np.median(np.diff(s[:, 0]))
print('steps %d; period (seg 0 got its code, step to step) median %.2f us' % (len(s), period))
rows = [('draw flag -> seg 0 has the code', s[:, 0] - np.roll(s[:, 15], 1))]
prev = s[:, 0]
for k in range(nseg):
    rows.append(('seg %d: start of layers (x + pre in) after previous event' % k, s[:, 1 + k] - prev))
    rows.append(('seg %d: layers + publish' % k, s[:, 6 + k] - s[:, 1 + k]))
    prev = s[:, 6 + k]
rows += [('last segment published -> skip wg 0 published h1', s[:, 11] - s[:, 6 + nseg - 1]),
         ('h1 -> post1 wg 0 published h2', s[:, 12] - s[:, 11]),
         ('h2 -> logits wg 0 published', s[:, 13] - s[:, 12]),
         ('logits -> draw wave has them', s[:, 14] - s[:, 13]),
         ('draw (softmax f64, sample, publish)', s[:, 15] - s[:, 14])]
tot = 0.0
for name, v in rows:
    v = v[1:]
    print('%-62s median %6.2f  p90 %6.2f' % (name, np.median(v), np.percentile(v, 90)))
    tot += np.median(v)
print('sum of medians %.2f us' % tot)
