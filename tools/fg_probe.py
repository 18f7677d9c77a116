import json, os, sys, time
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tensorflow-wavenet_amd'))
import torch
from wavenet import WaveNetModel
def run(name, L=50, S=512, Q=256, n=4000):
    net = WaveNetModel(batch_size=1, dilations=([2**i for i in range(10)]*5)[:L], filter_width=2, residual_channels=32,
                       dilation_channels=32, skip_channels=S, quantization_channels=Q, use_biases=True, seed=0)
    net.generate(200, seed_samples=[1], seed=1); torch.cuda.synchronize()
    t0=time.perf_counter(); net.generate(n, seed_samples=[1], seed=2); torch.cuda.synchronize()
    dt=time.perf_counter()-t0
    print('%-22s %7.1f us/sample' % (name, dt/n*1e6))
run('default L50 S512 Q256')
run('L50 S32 Q256', S=32)
run('L50 S32 Q16', S=32, Q=16)
run('L10 S512 Q256', L=10)
run('L10 S32 Q16', L=10, S=32, Q=16)
run('L1 S32 Q16', L=1, S=32, Q=16)
