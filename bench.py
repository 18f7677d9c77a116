#!/usr/bin/env python
"""Headline benchmark: audio samples/sec of a full WaveNet training step
(mu-law encode + forward + loss + backward + gradient all-reduce + Adam) on
the default wavenet_params.json stack, fp32, synthetic 16 kHz clips resident
in HBM (BASELINE.json metric; config[1]: B=8 clips x 16000 samples per GPU).

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL.  Under torch.distributed.run (RANK /
WORLD_SIZE in the environment) this process IS one rank; run plainly
(`python bench.py --gpus N`) it starts the N rank processes itself, before
touching the GPU, and exits with their status.  The per-GPU work is fixed (weak
scaling), `value` is the whole-job aggregate; `ranks_seen` is an RCCL
all-reduce of ones.  WORLD_SIZE != --gpus is an error.

Extra objects on the JSON line:
  roofline     - the dominant kernel (gemm_nn3_kernel: the skip-sum /
                 post-processing fp32 MFMA GEMMs and their data gradients),
                 timed live with HIP events on the launch stream in a short
                 instrumented pass of the same step right after the headline
                 loop (the headline loop itself carries no instrumentation);
                 achieved = algorithmic FLOPs / kernel time; traffic (PMC,
                 committed profile) next to traffic_algorithmic (operands in,
                 outputs out, epilogue masks / pre-activations) per launch.
  step_tflops / step_frac - the WHOLE step against the fp32 MFMA peak
                 (8.804 MFLOP per audio sample).
  N > 1 only: allreduce_us_per_step (HIP events around the gradient
                 all-reduce, instrumented pass), step_ms_min / step_ms_max
                 over ranks.
  cpu_baseline - the reference graph restated op for op in PyTorch-CPU
                 (oracle/torch_graph.py, kind "port": TF 0.10 cannot be
                 installed here) timed on this host's cores on a bounded
                 sample (config[0]: B=1, T=16000), rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tensorflow-wavenet_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
STEP_FLOP_PER_SAMPLE = 8.804e6  # fwd 2.935 MFLOP x 3, default stack


def synth_audio(B, T, first_clip=0, seed=1234, sample_rate=16000):
    """clip b = 0.5 sin(2 pi 110 2^(b/12) t / 16000) + 0.05 N(0,1),
    clipped to [-1, 1], noise from ONE default_rng(1234) stream drawn clip after
    clip; b is the GLOBAL clip index.  A rank of a data-parallel job draws (and
    drops) the noise of the clips before its own, so that N ranks x B clips
    hold exactly the batch one process x N B clips holds."""
    rng = np.random.default_rng(seed)
    t = np.arange(T)
    out = np.empty((B, T), np.float32)
    for g in range(first_clip + B):
        noise = rng.standard_normal(T)
        if g < first_clip:
            continue
        f = 110.0 * 2 ** (g / 12.0)
        x = 0.5 * np.sin(2 * np.pi * f * t / sample_rate) + 0.05 * noise
        out[g - first_clip] = np.clip(x, -1, 1)
    return out


def rank_gc_ids(rank, B):
    """BASELINE.json configs[3]: speaker id of GLOBAL clip b is (37 b) mod 377;
    rank r of a data-parallel job owns clips [r B, (r + 1) B)."""
    return [(37 * (rank * B + b)) % 377 for b in range(B)]


def collective_env():
    """What shapes the gradient all-reduce on this run: the RCCL version and
    every NCCL_* / RCCL_* variable found in the environment (SURVEY section 5
    asks for ring vs tree to be comparable from the driver's line)."""
    try:
        ver = torch.cuda.nccl.version()
        ver = '.'.join(str(v) for v in ver) if isinstance(ver, tuple) else str(ver)
    except Exception as e:      # noqa: BLE001 (a CPU-only torch build)
        ver = 'unavailable: %s' % type(e).__name__
    env = {k: v for k, v in sorted(os.environ.items())
           if k.startswith(('NCCL_', 'RCCL_'))}
    return {'rccl_version': ver,
            'NCCL_ALGO': os.environ.get('NCCL_ALGO'),
            'NCCL_PROTO': os.environ.get('NCCL_PROTO'),
            'env': env}


def host_cores(cap=16):
    """CPU threads this process may really use: affinity mask, cgroup quota,
    and the GPU box's per-GPU CPU share (16), whichever is smallest."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') \
        else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.999)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get('WN_CPU_THREADS', cap))))


def cpu_model():
    """CPU model string of this host (/proc/cpuinfo), SURVEY 8(d)."""
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or 'unknown'


def log(msg):
    sys.stderr.write('[bench] %s\n' % msg)
    sys.stderr.flush()


def cpu_baseline(params, T, max_seconds=20.0, max_steps=20):
    """Time the op-for-op CPU restatement of the reference graph (checker /
    baseline only; never part of the measured GPU path)."""
    from oracle import wavenet_oracle as O, torch_graph as TG
    cores = host_cores()
    torch.set_num_threads(cores)
    log('cpu baseline on %d threads' % cores)
    cfg = dict(params)
    cfg['batch_size'] = 1
    var = O.create_variables(cfg, seed=0, dtype=np.float32)
    step = TG.train_step_fn(cfg, var)
    q = torch.tensor(O.mu_law_encode(synth_audio(1, T), 256).astype(np.int64))
    t0 = time.time()
    step(q)                                   # warm-up (allocator, threads)
    log('cpu warm-up step %.1f s' % (time.time() - t0))
    n, t0 = 0, time.time()
    # bounded sample: about 10-20 s of CPU work (0.5-0.8 s per step here)
    while n < max_steps and (time.time() - t0) < max_seconds - 8.0:
        step(q)
        n += 1
        log('cpu step %d done (%.1f s elapsed)' % (n, time.time() - t0))
    dt = (time.time() - t0) / max(n, 1)
    return {'value': T / dt, 'unit': 'audio samples/s', 'cores': cores,
            'cpu_model': cpu_model(), 'kind': 'port',
            'sample': '%d full training steps (fwd+bwd+TF-Adam) of the '
                      'op-for-op PyTorch-CPU restatement of the reference '
                      'graph, default wavenet_params.json, B=1, T=%d '
                      '(%.2f s/step)' % (n, T, dt)}


def _csrc_sha16():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        from csrc_hash import csrc_hash
        return csrc_hash()
    except Exception:      # noqa: BLE001 (tools/ not shipped: unknown, never "fresh")
        return None


def _newest_profile(pattern, kernel):
    """(entry of `kernel`, relative path, stale) from the newest committed
    summary profiles/<pattern> that holds the kernel.  stale: the summary
    records the hash of the kernel sources it was collected on
    (tools/csrc_hash.py); True when it differs from this tree's (or when the
    summary predates the field), i.e. the figure is a constant read from an
    older build, not an observation of this run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
            e = d[kernel]
        except (KeyError, ValueError, OSError):
            continue
        sha = (d.get('_meta') or {}).get('csrc_sha16')
        cur = _csrc_sha16()
        return e, os.path.relpath(path, ROOT), (sha is None or cur is None or sha != cur)
    return None, None, None


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, source file, stale) from the newest
    committed PMC summary (profiles/*_pmc_traffic.json, written by
    tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes of this same command, with the gfx950 FETCH_SIZE x2 correction).
    (None, None, None) when no summary holds the kernel."""
    e, src, stale = _newest_profile('*_pmc_traffic.json', kernel)
    return (None, None, None) if e is None else (e['hbm_bytes_per_launch'], src, stale)


def rocprof_avg_us(kernel_prefix):
    """(average launch duration in us of the kernel whose name starts with
    `kernel_prefix`, source file, stale) from the newest committed
    `rocprofv3 --kernel-trace --stats` summary profiles/*_bench_kernel_stats.csv
    (tools/collect_profiles.sh: the same bench.py command).  stale as in
    _newest_profile, from the PMC summary of the same tag."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_bench_kernel_stats.csv')))
    for path in reversed(files):
        try:
            with open(path) as f:
                rows = [r for r in csv.DictReader(f) if r['Name'].startswith(kernel_prefix)]
        except (OSError, KeyError):
            continue
        if not rows:
            continue
        calls = sum(int(r['Calls']) for r in rows)
        total = sum(float(r['TotalDurationNs']) for r in rows)
        stale = True
        try:
            with open(path.replace('_bench_kernel_stats.csv', '_pmc_traffic.json')) as f:
                sha = (json.load(f).get('_meta') or {}).get('csrc_sha16')
            cur = _csrc_sha16()
            stale = sha is None or cur is None or sha != cur
        except (OSError, ValueError):
            pass
        return total / calls / 1e3, os.path.relpath(path, ROOT), stale
    return None, None, None


def pmc_issue(kernel):
    """(issue-slot summary of `kernel`, source file, stale) from the newest
    profiles/*_issue.json (tools/pmc_issue.py), or (None, None, None)."""
    return _newest_profile('*_issue.json', kernel)


def secondary(net, audio, gc_ids, kw, B, T, gen_samples=16000, opt=None):
    """SURVEY 8(d) secondary figures (rank 0, N = 1, after the timed region):
    forward-only samples/s on the same batch, and fast generation
    (BASELINE.json configs[4]: batch 1, seed 128, temperature 1)."""
    from wavenet import WaveNetModel
    for _ in range(2):
        net.loss(audio, gc_ids, backward=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        net.loss(audio, gc_ids, backward=False)
    torch.cuda.synchronize()
    fwd = B * T * 10 / (time.perf_counter() - t0)
    log('forward only: %.0f samples/s' % fwd)
    # opt-in split-bf16 GEMM mode on the same batch (NOT the headline: fp32
    # operands rebuilt from 6 bf16 piece products, same error vs float64 as the
    # fp32 MFMA kernels)
    optin, optin_roof = None, None
    if net.gemm_mode == 'fp32' and opt is not None:
        net.gemm_mode = 'bf16x6'
        for _ in range(3):
            opt.minimize(net.loss(audio, gc_ids))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            opt.minimize(net.loss(audio, gc_ids))
        torch.cuda.synchronize()
        optin = (time.perf_counter() - t0) / 10 * 1e3
        # its own roofline: the six NN GEMMs timed live (HIP events), in
        # fp32-EQUIVALENT TFLOP/s (the contraction's flops, not the six piece
        # products') against the dense bf16 peak / 6
        net._gemm_events = []
        for _ in range(3):
            opt.minimize(net.loss(audio, gc_ids))
        torch.cuda.synchronize()
        ev = [e for e in net._gemm_events if e[3] == 'wn_gemm_nn_split']
        net._gemm_events = None
        t_nn = sum(e[0].elapsed_time(e[1]) for e in ev) * 1e-3
        f_nn = sum(e[2] for e in ev)
        optin_roof = {'ms_per_step': optin,
                      'kernel': 'gemm_nn_split_kernel<6>',
                      'gemm_tflops_equiv': f_nn / t_nn / 1e12 if t_nn > 0 else None,
                      'peak': 2500.0 / 6, 'peak_note': 'dense bf16 MFMA 2500 TFLOP/s / 6 piece products',
                      'frac': f_nn / t_nn / 1e12 / (2500.0 / 6) if t_nn > 0 else None,
                      'nn_gemm_us_per_step': t_nn / 3 * 1e6}
        net.gemm_mode = 'fp32'
        log('opt-in bf16x6 GEMM mode: %.2f ms/step, NN GEMMs %.0f fp32-equivalent TFLOP/s'
            % (optin, optin_roof['gemm_tflops_equiv'] or 0))
    from wavenet import optimizer_factory

    def timed_steps(model, a, ids, n=10, warm=3):
        o = optimizer_factory['adam'](learning_rate=1e-3, momentum=0.9)
        for _ in range(warm):
            o.minimize(model.loss(a, ids))
        # median of three timed rounds: a round is only 20 - 100 ms, one host
        # hiccup (allocator, garbage collection) would double it
        rounds = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                o.minimize(model.loss(a, ids))
            torch.cuda.synchronize()
            rounds.append((time.perf_counter() - t0) / n * 1e3)
        return sorted(rounds)[1]

    # BASELINE.json configs[3] on ONE GPU: the same step with global
    # conditioning 32 x 377 (skipped when the headline run already has it)
    gc_ms = None
    if not kw.get('global_condition_channels'):
        kwg = dict(kw, global_condition_channels=32,
                   global_condition_cardinality=377)
        netg = WaveNetModel(seed=0, **kwg)
        idsg = torch.tensor([(37 * b) % 377 for b in range(B)],
                            dtype=torch.int32, device=audio.device)
        gc_ms = timed_steps(netg, audio, idsg)
        del netg
        log('global conditioning 32x377: %.2f ms/step' % gc_ms)
    # configs[0] shape (one clip) on the GPU
    kw1 = dict(kw)
    kw1['batch_size'] = 1
    net1 = WaveNetModel(seed=0, **kw1)
    ids1 = None if gc_ids is None else gc_ids[:1]
    b1_ms = timed_steps(net1, audio[:1], ids1)
    log('B=1: %.2f ms/step' % b1_ms)
    # the reference's own training shape: one piece of `--sample_size` 100 000
    # samples (train.py:30; SURVEY 8d asks for it as a secondary figure)
    t100k_ms = None
    if T != 100000:
        a100 = torch.from_numpy(synth_audio(1, 100000)).to(audio.device)
        t100k_ms = timed_steps(net1, a100, ids1, n=5, warm=2)
        del a100
        log('B=1, T=100000: %.2f ms/step' % t100k_ms)
    del net1
    torch.cuda.empty_cache()
    gen = WaveNetModel(seed=0, **kw1)
    gc = 5 if kw.get('global_condition_channels') else None
    gen.generate(200, seed_samples=[128], seed=1, global_condition=gc)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gen.generate(gen_samples, seed_samples=[128], temperature=1.0, seed=2,
                 global_condition=gc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    log('fast generation: %.0f samples/s' % (gen_samples / dt))
    return {'forward_only_samples_per_s': fwd,
            'fastgen_samples_per_s': gen_samples / dt,
            'fastgen_us_per_sample': dt / gen_samples * 1e6,
            'fastgen_samples': gen_samples,
            # one persistent multi-CU launch for the run (wn_fastgen_persist) or
            # four step kernels per sample replayed from a hipGraph
            'fastgen_path': 'persistent launch' if gen.fastgen_persistent
            else 'step kernels in a hipGraph',
            'gc_ms_per_step': gc_ms,
            'gc_samples_per_s': None if gc_ms is None else B * T / gc_ms * 1e3,
            'b1_ms_per_step': b1_ms,
            'b1_samples_per_s': T / b1_ms * 1e3,
            't100k_ms_per_step': t100k_ms,
            't100k_samples_per_s': None if t100k_ms is None else 100000 / t100k_ms * 1e3,
            'optin_bf16x6_ms_per_step': optin,
            'optin_bf16x6': optin_roof}


class _Emitter:
    """Prints the ONE JSON line, exactly once, from whichever comes first: the
    normal end of the run or the watchdog of a guarded region (N > 1: the
    two-call all-reduce trial and every collective after it).  When a guarded
    region does not return within its bound -- a collective that never
    completes on some rank -- rank 0 prints the line it prepared from the
    schedule already measured (`fallback`, "overlap_failed": true) and every
    rank leaves with os._exit(0): a process whose device queue is stuck cannot
    be unwound, must not be re-executed (GPU box rule), and the measured
    figures are still valid."""

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.done = False
        self.fallback = None

    def line(self, text):
        with self.lock:
            if not self.done:
                self.done = True
                sys.stdout.write(text + '\n')
                sys.stdout.flush()

    def _expired(self, what, bound):
        log('%s did not return within %.0f s: giving up on it' % (what, bound))
        if self.fallback is not None:
            self.line(self.fallback.replace(
                '@WHY@', '%s did not return within %.0f s' % (what, bound)))
        sys.stderr.flush()
        os._exit(0)

    def guarded(self, fn, what='a collective', bound=None):
        """fn() under a watchdog thread."""
        import threading
        if bound is None:
            bound = float(os.environ.get('WN_OVERLAP_TRIAL_TIMEOUT', '120'))
        t = threading.Timer(bound, self._expired, (what, bound))
        t.daemon = True
        t.start()
        try:
            return fn()
        finally:
            t.cancel()


def overlap_trial(net, parallel, timed, instrumented, over_ranks, args, isteps,
                  dev, emit):
    """N > 1: the two-call gradient exchange (`dp_overlap_allreduce`: the skip /
    post-processing tail all-reduced on a communication stream beside the
    backward stack, the head at the update) timed in THIS process after the
    one-call schedule: 2 warm-up steps (the launch plans are re-recorded), K
    timed steps, the instrumented pass.  Guarded three ways, each leaving the
    process usable for the one-call figures already in hand:
      * an exception (a collective error) is caught, a dangling tail joined;
      * a non-finite loss or an expired dependency wait inside a persistent
        stack launch (the collective's kernel held CUs for seconds) is a
        failure, agreed over the ranks with an all-reduce(MAX);
      * the whole trial runs under the emitter's watchdog (a hang).
    Returns {'failed': reason} or the schedule's figures."""
    import math

    def body():
        why = None
        res = {}
        try:
            net.dp_overlap_allreduce = True
            for _ in range(2):
                net_step_loss = timed(1)[1]
            # (tests: WN_BENCH_INJECT_OVERLAP_FAILURE=raise|hang|nan exercises
            # the three guards on a box where the schedule itself works)
            inject = os.environ.get('WN_BENCH_INJECT_OVERLAP_FAILURE')
            if inject == 'raise':
                raise RuntimeError('injected failure')
            if inject == 'hang':
                time.sleep(1e6)
            dt, loss = timed(args.steps)
            _, ar_us = instrumented(isteps)
            lossf = float('nan') if inject == 'nan' else float(loss)
            if not math.isfinite(lossf) or not math.isfinite(float(net_step_loss)):
                why = 'non-finite loss under the two-call schedule'
            try:
                net.check_device_errors()
            except Exception as e:         # noqa: BLE001
                why = 'expired dependency wait in a stack launch: %s' % (e,)
                net.reset_device_errors()
            res = dict(dt=dt, loss=loss, ar_us=ar_us)
        except Exception as e:             # noqa: BLE001
            why = 'exception under the two-call schedule: %r' % (e,)
            parallel.abandon_tail_allreduce(net)
        # the ranks agree (a failure on one is a failure of the schedule)
        bad = parallel.any_rank(why is not None, device=dev)
        if bad:
            net.dp_overlap_allreduce = False
            log('two-call all-reduce trial failed: %s' % (why or 'on another rank'))
            return {'failed': why or 'failed on another rank'}
        dt_max, dt_min, ar = over_ranks(res['dt'], res['ar_us'])
        gl = float(parallel.allreduce_mean_scalar(res['loss'].reshape(1).float())[0])
        entry = {'ms_per_step': dt_max / args.steps * 1e3,
                 'step_ms_min': dt_min / args.steps * 1e3,
                 'step_ms_max': dt_max / args.steps * 1e3,
                 'allreduce_us_per_step': ar, 'final_loss': float(res['loss']),
                 'global_loss': gl}
        return {'entry': entry, 'dt_max': dt_max, 'dt_min': dt_min, 'ar_us': ar,
                'final_loss': float(res['loss']), 'global_loss': gl}

    return emit.guarded(body, 'the two-call all-reduce trial')


def launch_ranks(n):
    """Start `n` copies of this script as rank processes (what
    `python -m torch.distributed.run --nproc-per-node n` would do) and return
    the worst exit status.  A rank that fails takes the others down."""
    import socket
    import subprocess
    port = os.environ.get('MASTER_PORT')     # a preset port is honoured
    if not port:
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                       'HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable,
                                       os.path.abspath(__file__)]
                                      + sys.argv[1:], env=env))
    status = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                status = status or rc
                for q in alive:              # our own children, by handle
                    q.terminate()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='clips per GPU')
    ap.add_argument('--samples', type=int, default=16000, help='T per clip')
    ap.add_argument('--gc', action='store_true',
                    help='config[3]: global conditioning 32 x 377')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--gemm-mode', default='fp32',
                    choices=['fp32', 'bf16x3', 'bf16x6', 'bf16x9'],
                    help='opt-in split-bf16 NN GEMMs (default: fp32 MFMA)')
    ap.add_argument('--set', action='append', default=[], metavar='ATTR=VALUE',
                    help='A/B runs: set a WaveNetModel attribute (a Python '
                         'literal), e.g. --set stack_fwd=False --set '
                         'stack_variant=0x1010; repeatable')
    ap.add_argument('--no-overlap-trial', action='store_true',
                    help='N > 1: time the one-call gradient all-reduce only '
                         '(default: both schedules, value from the faster)')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the forward-only and fast-generation figures')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # run plainly: start the N ranks ourselves.  Nothing in this process
        # has touched the GPU yet (importing torch does not), the children are
        # ordinary subprocesses, and the parent only waits for them.
        sys.exit(launch_ranks(args.gpus))

    from wavenet import WaveNetModel, optimizer_factory, parallel
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit('bench.py: WORLD_SIZE %d != --gpus %d (launch with '
                         '--nproc-per-node %d, or run `python bench.py --gpus '
                         '%d` plainly so that it starts the ranks itself)'
                         % (world, args.gpus, args.gpus, args.gpus))
    # one rank per GPU; WN_SHARE_GPU=1 (rehearsal only) maps every rank to the
    # devices that exist
    ndev = torch.cuda.device_count()
    if local >= ndev and os.environ.get('WN_SHARE_GPU') != '1':
        raise SystemExit('rank %d has no GPU (found %d)' % (local, ndev))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    with open(os.path.join(ROOT, 'wavenet_params.json')) as f:
        params = json.load(f)
    B, T = args.batch, args.samples
    kw = dict(batch_size=B, dilations=params['dilations'],
              filter_width=params['filter_width'],
              residual_channels=params['residual_channels'],
              dilation_channels=params['dilation_channels'],
              skip_channels=params['skip_channels'],
              quantization_channels=params['quantization_channels'],
              use_biases=params['use_biases'],
              scalar_input=params['scalar_input'],
              initial_filter_width=params['initial_filter_width'],
              residual_postproc=params.get('residual_postproc', False))
    gc_ids = None
    if args.gc:
        kw.update(global_condition_channels=32,
                  global_condition_cardinality=377)
        gc_ids = torch.tensor(rank_gc_ids(rank, B), dtype=torch.int32,
                              device=dev)
    net = WaveNetModel(seed=0, **kw)
    net.gemm_mode = args.gemm_mode
    # data-parallel: N > 1 times BOTH gradient-exchange schedules in this
    # process (one all-reduce at the update; or the bucket's tail beside the
    # backward stack + the head at the update, wavenet/parallel.py) and reports
    # the faster; --set dp_overlap_allreduce=... pins one
    for item in args.set:        # A/B knobs: explicit model attributes
        import ast
        name, _, val = item.partition('=')
        if not hasattr(net, name):
            raise SystemExit('bench.py --set: WaveNetModel has no attribute %r' % name)
        setattr(net, name, ast.literal_eval(val))
    parallel.broadcast_parameters(net)
    opt = optimizer_factory['adam'](learning_rate=1e-3, momentum=0.9)
    audio = torch.from_numpy(synth_audio(B, T, first_clip=rank * B)).to(dev)

    def step():
        loss = net.loss(audio, gc_ids)
        opt.minimize(loss)
        return loss

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # one line per rank before the first step: which device this rank drives
    prop = torch.cuda.get_device_properties(local)
    bus = '%04x:%02x:%02x' % (getattr(prop, 'pci_domain_id', 0),
                              getattr(prop, 'pci_bus_id', 0),
                              getattr(prop, 'pci_device_id', 0))
    sys.stderr.write('[bench] rank %d/%d (local %d): device %d = %s, PCI %s, '
                     '%d CUs, clips [%d, %d)\n'
                     % (rank, world, int(os.environ.get('LOCAL_RANK', 0)), local,
                        prop.name, bus, prop.multi_processor_count,
                        rank * B, (rank + 1) * B))
    sys.stderr.flush()
    log('rank %d/%d: model built, warming up' % (rank, world))

    def timed(k):
        """EXACTLY k steps between barrier + device sync on both sides."""
        sync_all()
        t0 = time.perf_counter()
        for _ in range(k):
            loss = step()
        sync_all()
        return time.perf_counter() - t0, loss

    def instrumented(n):
        """(NOT part of `value`) the same step with HIP events around every
        GEMM / stack launch and around the gradient all-reduce"""
        net._gemm_events = []
        parallel.timing_events = [] if world > 1 else None
        for _ in range(n):
            step()
        sync_all()
        ev = net._gemm_events or []
        net._gemm_events = None
        ar = parallel.timing_events or []
        parallel.timing_events = None
        us = sum(a.elapsed_time(b) for a, b in ar) / n * 1e3 if ar else None
        return ev, us

    def over_ranks(dt, ar_us):
        """(max, min) of the timed region and max of the all-reduce time"""
        if world == 1:
            return dt, dt, ar_us
        tt = torch.tensor([dt, -dt, ar_us or 0.0], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        return float(tt[0]), -float(tt[1]), float(tt[2])

    # N > 1: the ONE-call schedule (a single all-reduce of the whole bucket at
    # the update) is timed first, unless --set names a schedule: it is the
    # path every rehearsal has run, and its figures are in hand before the
    # two-call schedule -- whose tail collective runs on a communication
    # stream beside the persistent backward-stack launch and has never met
    # RCCL with N >= 2 -- is tried in this same process (overlap_trial below).
    explicit = any(i.partition('=')[0] == 'dp_overlap_allreduce' for i in args.set)
    if world > 1 and not explicit:
        net.dp_overlap_allreduce = False
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log('warm-up done, timing %d steps' % args.steps)
    net._gemm_events = None          # the headline loop is un-instrumented
    dt, loss = timed(args.steps)
    isteps = max(1, min(5, args.steps))
    events, ar_us = instrumented(isteps)
    ranks_seen = 1
    dt_max, dt_min, ar_us = over_ranks(dt, ar_us)
    dt = dt_max
    global_loss, ids_all = float(loss), None
    if world > 1:
        gl = parallel.allreduce_mean_scalar(loss.reshape(1).float())
        global_loss = float(gl[0])
        if gc_ids is not None:
            got = [torch.empty_like(gc_ids) for _ in range(world)]
            torch.distributed.all_gather(got, gc_ids)
            ids_all = [int(v) for t in got for v in t.cpu().tolist()]
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        torch.distributed.all_reduce(ones)          # RCCL: every rank adds 1
        ranks_seen = int(round(float(ones[0])))
        if ranks_seen != args.gpus:
            raise SystemExit('all-reduce saw %d ranks, expected %d'
                             % (ranks_seen, args.gpus))
    cur = {'dt': dt, 'dt_min': dt_min, 'dt_max': dt_max, 'ar_us': ar_us,
           'final_loss': float(loss), 'global_loss': global_loss,
           'calls': 2 if net.dp_overlap_allreduce else 1,
           'schedules': {}, 'overlap_failed': None, 'overlap_failure': None}
    cur['schedules']['two_call' if cur['calls'] == 2 else 'one_call'] = {
        'ms_per_step': dt / args.steps * 1e3,
        'step_ms_min': dt_min / args.steps * 1e3,
        'step_ms_max': dt_max / args.steps * 1e3,
        'allreduce_us_per_step': ar_us, 'final_loss': float(loss),
        'global_loss': global_loss}

    def result_line(cur):
        """The JSON line from the figures of the schedule in `cur` (the
        roofline objects come from the instrumented pass of the first
        schedule: the kernels are the same under both)."""
        dt, dt_min, dt_max, ar_us = cur['dt'], cur['dt_min'], cur['dt_max'], cur['ar_us']
        final_loss, global_loss = cur['final_loss'], cur['global_loss']
        value = world * B * T * args.steps / dt
        # dominant kernel = the NN GEMM launches (skip sum, post1, post2 and
        # their data gradients); the TN weight-gradient GEMMs are reported beside
        nn = [e for e in events if 'gemm_nn' in e[3]]
        tn = [e for e in events if 'gemm_tn' in e[3]]
        flops = sum(e[2] for e in nn)
        ktime = sum(e[0].elapsed_time(e[1]) for e in nn) * 1e-3
        achieved = flops / ktime / 1e12 if ktime > 0 else 0.0
        nlaunch = len(nn)
        tn_flops = sum(e[2] for e in tn)
        tn_time = sum(e[0].elapsed_time(e[1]) for e in tn) * 1e-3
        # small batches run the TN GEMMs on a side stream beside the backward
        # stack (net.overlap_tn): the event pairs, recorded on the main stream,
        # then bracket nothing -- no TN figure rather than a meaningless one
        tn_side = any(net._overlap_tn_on(w) for w in net._ws.values() if w.training)
        if tn_side:
            tn_time = 0.0
        step_tflops = STEP_FLOP_PER_SAMPLE * B * T / (dt / args.steps) / 1e12
        peak = FP32_MFMA_PEAK_TFLOPS if args.gemm_mode == 'fp32' else \
            2500.0 / int(args.gemm_mode[-1])
        dom = 'gemm_nn3_kernel' if args.gemm_mode == 'fp32' else \
            'gemm_nn_split_kernel'
        Sk, Qc = int(params['skip_channels']), int(params['quantization_channels'])
        LC = len(params['dilations']) * 32
        algo_bytes = 4.0 * B * T * (
            (LC + Sk) + (3 * Sk) + (Sk + Qc) + (Qc + 2 * Sk) + (3 * Sk) + (Sk + LC)) / 6
        # (the committed PMC summary was collected at the default shape only)
        traffic, traffic_src, traffic_stale = pmc_traffic(dom) if (B, T) == (8, 16000) \
            else (None, None, None)
        # the two persistent residual-stack launches.  Their bound is ISSUE (on
        # gfx950 the f32 MFMA and the vector ALU are one resource: DESIGN.md,
        # profiles/*_mfma_valu.txt), neither HBM nor the matrix pipe alone: time
        # live (HIP events), the issue-slot fractions and the fabric bytes from the
        # committed PMC summaries of the kernel that RAN (name = what the library's
        # shape / variant rules pick), each tagged stale when the kernel sources
        # changed since it was collected.  The TB/s figure is information, not a
        # fraction of a bound.
        rp_us, rp_src, rp_stale = rocprof_avg_us(dom) if (B, T) == (8, 16000) \
            else (None, None, None)
        stacks = {}
        lib_ = net  # (kernel names follow wn_stack_tile_rows / the variant word)
        tws = [w for w in net._ws.values() if w.training]
        rows_f = tws[0].stack_rows if tws else 32
        rows_b = rows_f
        kern = {'wn_stack_fwd': 'void stack_fwd_kernel<2, 16>' if rows_f == 32
                else 'void stack_fwd16_kernel<2, 8>',
                'wn_stack_bwd': 'void stack_bwd_kernel<8>' if rows_b == 32
                else 'void stack_bwd16_kernel<8>'}
        tile_layers = B * ((T + 31) // 32) * len(params['dilations'])
        mfma_cyc = {'wn_stack_fwd': 80 * 64, 'wn_stack_bwd': 160 * 64}    # per 32-row tile and layer
        for ev_name in ('wn_stack_fwd', 'wn_stack_bwd'):
            evs = [e for e in events if e[3] == ev_name or
                   (ev_name == 'wn_stack_fwd' and e[3] == 'wn_stack_fwd_skip')]
            if not evs:
                continue
            us = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs) * 1e3
            kname = kern[ev_name]
            by, src, stale = pmc_traffic(kname) if (B, T) == (8, 16000) else (None, None, None)
            iss, isrc, istale = pmc_issue(kname) if (B, T) == (8, 16000) else (None, None, None)
            stacks[ev_name] = {
                'kernel': kname.replace('void ', ''), 'avg_launch_us': us,
                'bound': 'issue',
                # live: the launch's MFMA work (algorithmic) over its time,
                # against 1024 SIMDs at the 2.4 GHz peak clock
                'mfma_busy_frac_live_at_2p4ghz': tile_layers * mfma_cyc[ev_name] / 1024.0 / (us * 2400.0),
                'issue': None if iss is None else {
                    k: iss.get(k) for k in ('mfma_busy_frac', 'valu_issue_frac', 'issue_frac',
                                            'clock_ghz', 'avg_us')},
                'issue_source': isrc, 'issue_stale': istale,
                'traffic': by, 'traffic_source': src, 'traffic_stale': stale,
                'fabric_tb_s_info': None if by is None else by / us / 1e6}
        out = {
            'metric': 'audio samples/sec (train, default wavenet_params.json)',
            'value': value, 'unit': 'audio samples/s', 'n_gpus': world,
            'ranks_seen': ranks_seen,
            'dist_backend': torch.distributed.get_backend() if world > 1 else None,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'step_tflops': step_tflops,
            'step_frac': step_tflops / FP32_MFMA_PEAK_TFLOPS,
            'step_tflops_note': 'whole step per GPU: 8.804 MFLOP per audio sample '
                                '(SURVEY 8d) x samples / step time, against the '
                                '157.3 TFLOP/s fp32 MFMA peak',
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.gemm_mode == 'fp32' else
                     'f32 (NN GEMM products rebuilt from %s split pieces on bf16 '
                     'MFMA, fp32 accumulate; opt-in)' % args.gemm_mode,
            'data': 'synthetic',
            'config': {'workload': 'default wavenet_params.json stack (50 dilation '
                                   'layers, R=D=32, S=512, Q=256), full training '
                                   'step, %d clips x %d samples per GPU, fp32%s'
                                   % (B, T, ', global conditioning 32x377'
                                      if args.gc else ''),
                       'clips_per_gpu': B, 'samples_per_clip': T,
                       'global_batch': world * B,
                       'parallelism': 'dp%d' % world,
                       'final_loss': final_loss,
                       # mean over ranks = the loss of the global batch (equal
                       # per-rank B*T, wavenet/parallel.py)
                       'global_loss': global_loss,
                       'gc_ids': ids_all if ids_all is not None else
                       (None if gc_ids is None else gc_ids.cpu().tolist())},
            'roofline': {'bound': 'mfma',
                         'kernel': dom,
                         'achieved': achieved, 'peak': peak,
                         'unit': 'TFLOP/s' if args.gemm_mode == 'fp32' else
                                 'TFLOP/s (fp32-equivalent; peak = bf16 dense '
                                 '2500 / piece products)',
                         'frac': achieved / peak,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'traffic_stale': traffic_stale,
                         'traffic_unit': 'bytes per launch (rocprofv3 PMC pass)',
                         # what the six launches of a step must move (fp32 path):
                         # A operands in, outputs out, the pre-activation plane the
                         # ReLU backward needs out, its two masks in; per launch
                         'traffic_algorithmic': algo_bytes,
                         'traffic_ratio': None if traffic is None
                         else traffic / algo_bytes,
                         'launches_per_step': nlaunch // isteps,
                         'avg_launch_us': ktime / max(nlaunch, 1) * 1e6,
                         # the same fraction from the committed rocprofv3 --stats
                         # summary (its average launch duration; the profiler's
                         # own overhead makes it a few per cent lower than live)
                         'frac_rocprof': None if not (rp_us and nlaunch) else
                         flops / nlaunch / (rp_us * 1e-6) / 1e12 / peak,
                         'rocprof_avg_launch_us': rp_us, 'rocprof_source': rp_src,
                         'rocprof_stale': rp_stale,
                         'flops_per_step': flops / isteps,
                         'measured_over': '%d instrumented steps after the timed '
                                          'region' % isteps,
                         'tn_gemms': {
                             'kernel': 'gemm_tn3_kernel',
                             'achieved': tn_flops / tn_time / 1e12
                             if tn_time > 0 else None,
                             'frac': tn_flops / tn_time / 1e12 / peak
                             if tn_time > 0 else None,
                             'launches_per_step': len(tn) // isteps,
                             'us_per_step': tn_time / isteps * 1e6
                             if tn_time > 0 else None,
                             'note': 'on a side stream beside the backward stack '
                                     '(small batch): not timed' if tn_side else None},
                         'stack_launches': stacks},
        }
        if world > 1:
            out['allreduce_us_per_step'] = ar_us
            out['allreduce_note'] = ('max over ranks of the HIP-event time around '
                                     'the gradient all-reduce issued at the update '
                                     '(includes waiting for the slowest rank); with '
                                     'allreduce_calls == 2 that is the bucket\'s head '
                                     'only: the skip / post-processing tail was '
                                     'reduced beside the backward stack.  N > 1 '
                                     'times BOTH schedules in this process, K steps '
                                     'each (allreduce_schedules); value is the '
                                     'faster one\'s')
            # which schedule `value` / `ms_per_step` are from (the faster of the
            # two timed in this process), both schedules' figures, and whether
            # the two-call trial failed (then: one call, value from it)
            out['allreduce_calls'] = cur['calls']
            out['allreduce_schedules'] = cur['schedules']
            out['overlap_failed'] = cur['overlap_failed']
            out['overlap_failure'] = cur['overlap_failure']
            out['allreduce_tail_bytes'] = int((net.grads.numel() - parallel.tail_start(net))
                                              * net.grads.element_size())
            out['allreduce_bytes'] = int(net.grads.numel() * net.grads.element_size())
            out['collective'] = collective_env()
            out['step_ms_min'] = dt_min / args.steps * 1e3
            out['step_ms_max'] = dt_max / args.steps * 1e3
        return out

    emit = _Emitter()
    if world > 1 and not explicit and not args.no_overlap_trial:
        # everything the line needs from the one-call schedule is in hand:
        # rank 0 prints it from the watchdog if the trial never returns
        # (as TEXT, built now: no HIP call is made from the watchdog thread)
        if rank == 0:
            emit.fallback = json.dumps(result_line(dict(
                cur, overlap_failed=True, overlap_failure='@WHY@')))
        two = overlap_trial(net, parallel, timed, instrumented, over_ranks,
                            args, isteps, dev, emit)
        if two.get('failed'):
            cur['overlap_failed'], cur['overlap_failure'] = True, two['failed']
        else:
            cur['overlap_failed'] = False
            cur['schedules']['two_call'] = two['entry']
            if two['dt_max'] < cur['dt']:
                # the faster schedule is the job's throughput
                cur.update(dt=two['dt_max'], dt_min=two['dt_min'], dt_max=two['dt_max'],
                           ar_us=two['ar_us'], final_loss=two['final_loss'],
                           global_loss=two['global_loss'], calls=2)
    if rank != 0:
        if world > 1:
            emit.guarded(torch.distributed.barrier)
            torch.distributed.destroy_process_group()
        return
    out = result_line(cur)
    log('gpu: %.0f samples/s, %.2f ms/step' % (out['value'], out['ms_per_step']))
    if world == 1 and not args.no_secondary:
        out['secondary'] = secondary(net, audio, gc_ids, kw, B, T, opt=opt)
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(params, T)
        out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
    emit.line(json.dumps(out))
    if world > 1:
        emit.guarded(torch.distributed.barrier)
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
