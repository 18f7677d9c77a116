#!/usr/bin/env python
"""Training script: MI355X counterpart of the reference's train.py.

Same flags, defaults, logdir rules and progress line as the reference
(train.py:22-101 flags, :142-180 directory validation, :104-134 checkpoint
save / restore with the step parsed from the file name, :310-311
`step N - loss = x, (y sec/step)`), driving the HIP WaveNetModel.  New:
  * --synthetic : train on generated sine clips instead of --data_dir wavs;
  * data-parallel when launched by torchrun (one rank per GPU, RCCL);
  * --store_metadata dumps a kernel-level Chrome trace (torch.profiler) every
    50th step in place of TF's RunMetadata timeline.
Checkpoints are torch files `model.ckpt-<step>` holding {reference variable
name: tensor}; scalars go to <logdir>/events.jsonl (TensorBoard is TF-only).
"""
from __future__ import print_function

import argparse
import glob
import json
import os
import re
import sys
import time
from datetime import datetime

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import tf_checkpoint  # noqa: E402

BATCH_SIZE = 1
DATA_DIRECTORY = './VCTK-Corpus'
LOGDIR_ROOT = './logdir'
CHECKPOINT_EVERY = 50
NUM_STEPS = int(1e5)
LEARNING_RATE = 1e-3
WAVENET_PARAMS = './wavenet_params.json'
STARTED_DATESTRING = "{0:%Y-%m-%dT%H-%M-%S}".format(datetime.now())
SAMPLE_SIZE = 100000
L2_REGULARIZATION_STRENGTH = 0
SILENCE_THRESHOLD = 0.3
EPSILON = 0.001
MOMENTUM = 0.9


def _str_to_bool(s):
    if s.lower() not in ('true', 'false'):
        raise ValueError('Argument needs to be a boolean, got {}'.format(s))
    return s.lower() == 'true'


def get_arguments(argv=None):
    p = argparse.ArgumentParser(description='WaveNet example network')
    p.add_argument('--batch_size', type=int, default=BATCH_SIZE,
                   help='How many wav files to process at once (per GPU).')
    p.add_argument('--data_dir', type=str, default=DATA_DIRECTORY,
                   help='The directory containing the VCTK corpus.')
    p.add_argument('--store_metadata', type=bool, default=False,
                   help='Store a kernel trace every 50 steps.')
    p.add_argument('--logdir', type=str, default=None,
                   help='Directory for logs / checkpoints; continues training '
                   'if it holds a model. Not with --logdir_root / '
                   '--restore_from.')
    p.add_argument('--logdir_root', type=str, default=None,
                   help='Root under which a dated logdir is created.')
    p.add_argument('--restore_from', type=str, default=None,
                   help='Directory to restore the model from (new logdir).')
    p.add_argument('--checkpoint_every', type=int, default=CHECKPOINT_EVERY)
    p.add_argument('--num_steps', type=int, default=NUM_STEPS)
    p.add_argument('--learning_rate', type=float, default=LEARNING_RATE)
    p.add_argument('--wavenet_params', type=str, default=WAVENET_PARAMS)
    p.add_argument('--sample_size', type=int, default=SAMPLE_SIZE,
                   help='Concatenate and cut audio samples to this many '
                   'samples.')
    p.add_argument('--l2_regularization_strength', type=float,
                   default=L2_REGULARIZATION_STRENGTH)
    p.add_argument('--silence_threshold', type=float,
                   default=SILENCE_THRESHOLD)
    p.add_argument('--optimizer', type=str, default='adam',
                   choices=['adam', 'sgd', 'rmsprop'])
    p.add_argument('--momentum', type=float, default=MOMENTUM)
    p.add_argument('--histograms', type=_str_to_bool, default=False)
    p.add_argument('--gc_channels', type=int, default=None,
                   help='Number of global condition channels.')
    p.add_argument('--dp_overlap_allreduce', type=_str_to_bool, default=False,
                   help='Data-parallel runs: all-reduce the skip / '
                        'post-processing gradients on a communication stream '
                        'beside the backward stack (two calls per step) '
                        'instead of one all-reduce at the update.  Off by '
                        'default: `bench.py --gpus N` times both schedules on '
                        'the node at hand -- switch it on where that run '
                        'reports allreduce_calls == 2 and overlap_failed false.')
    p.add_argument('--synthetic', action='store_true',
                   help='Train on synthetic sine clips (no --data_dir needed).')
    p.add_argument('--gc_cardinality', type=int, default=None,
                   help='Only with --synthetic: number of speaker ids.')
    return p.parse_args(argv)


def checkpoint_path(logdir, step):
    return os.path.join(logdir, 'model.ckpt-{}'.format(step))


def save(net, logdir, step):
    print('Storing checkpoint to {} ...'.format(logdir), end="")
    sys.stdout.flush()
    os.makedirs(logdir, exist_ok=True)
    path = checkpoint_path(logdir, step)
    torch.save({'variables': net.state_dict(), 'step': step}, path)
    with open(os.path.join(logdir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "{}"\n'.format(os.path.basename(path)))
    print(' Done.')


def latest_checkpoint(logdir):
    """Newest `model.ckpt-<step>` in logdir, or None."""
    marker = os.path.join(logdir, 'checkpoint')
    if os.path.exists(marker):
        name = open(marker).read().split('"')[1]
        path = os.path.join(logdir, name)
        # (a TensorFlow V2 checkpoint is a prefix: model.ckpt-N.index / .data-*)
        if os.path.exists(path) or tf_checkpoint.checkpoint_format(path):
            return path
    # only `model.ckpt-<step>` itself or a V2 prefix's `.index`: a V2 data
    # shard (`model.ckpt-N.data-00000-of-00001`) and `.meta` also end in
    # digits / start with the prefix, and must not be taken for a checkpoint
    found = {}
    for f in glob.glob(os.path.join(logdir, 'model.ckpt-*')):
        base = f[:-len('.index')] if f.endswith('.index') else f
        m = re.match(r'model\.ckpt-(\d+)$', os.path.basename(base))
        if m:
            found[base] = int(m.group(1))
    return max(found, key=found.get) if found else None


def load(net, logdir):
    print("Trying to restore saved checkpoints from {} ...".format(logdir),
          end="")
    path = latest_checkpoint(logdir) if os.path.isdir(logdir) else None
    if path is None:
        print(" No checkpoint found.")
        return None
    print("  Checkpoint found: {}".format(path))
    global_step = int(path.split('/')[-1].split('-')[-1])
    print("  Global step was: {}".format(global_step))
    print("  Restoring...", end="")
    if tf_checkpoint.checkpoint_format(path):
        # written by the reference's tf.train.Saver (train.py:104-114 there)
        tf_checkpoint.load_into(net, path)
    else:
        net.load_state_dict(torch.load(path, map_location='cpu')['variables'])
    print(" Done.")
    return global_step


def get_default_logdir(logdir_root):
    return os.path.join(logdir_root, 'train', STARTED_DATESTRING)


def validate_directories(args):
    """Validate and arrange directory related arguments (train.py:142-180)."""
    if args.logdir and args.logdir_root:
        raise ValueError("--logdir and --logdir_root cannot be "
                         "specified at the same time.")
    if args.logdir and args.restore_from:
        raise ValueError(
            "--logdir and --restore_from cannot be specified at the same "
            "time. This is to keep your previous model from unexpected "
            "overwrites.\nUse --logdir_root to specify the root of the "
            "directory which will be automatically created with current date "
            "and time, or use only --logdir to just continue the training "
            "from the last checkpoint.")
    logdir_root = args.logdir_root or LOGDIR_ROOT
    logdir = args.logdir
    if logdir is None:
        logdir = get_default_logdir(logdir_root)
        print('Using default logdir: {}'.format(logdir))
    restore_from = args.restore_from or logdir
    return {'logdir': logdir, 'logdir_root': args.logdir_root,
            'restore_from': restore_from}


class SyntheticReader(object):
    """Sine-plus-noise clips of `sample_size` samples (BASELINE.md data)."""

    def __init__(self, sample_size, gc_cardinality=None, rank=0, seed=1234):
        self.T, self.card = sample_size, gc_cardinality
        self.rng = np.random.default_rng(seed + rank)
        self.gc_category_cardinality = gc_cardinality
        self.count = rank * 1000003

    def _clip(self):
        self.count += 1
        f = 110.0 * 2 ** ((self.count % 36) / 12.0)
        t = np.arange(self.T)
        x = 0.5 * np.sin(2 * np.pi * f * t / 16000.0) + \
            0.05 * self.rng.standard_normal(self.T)
        return np.clip(x, -1, 1).astype(np.float32), self.count

    def dequeue(self, n):
        clips = [self._clip() for _ in range(n)]
        self._last_ids = [c[1] for c in clips]
        return torch.from_numpy(np.stack([c[0] for c in clips]))[..., None]

    def dequeue_gc(self, n):
        return torch.tensor([(37 * i) % self.card for i in self._last_ids],
                            dtype=torch.int32)

    def start_threads(self, *a, **k):
        return []


def main(argv=None):
    args = get_arguments(argv)
    try:
        directories = validate_directories(args)
    except ValueError as e:
        print("Some arguments are wrong:")
        print(str(e))
        return 1
    logdir = directories['logdir']
    restore_from = directories['restore_from']
    # a restored model written somewhere else counts as a new training
    is_overwritten_training = logdir != restore_from

    from wavenet import WaveNetModel, AudioReader, optimizer_factory, parallel
    from wavenet.audio_reader import Coordinator
    rank, world, local = parallel.init_from_env(host_control_plane=True)
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))

    with open(args.wavenet_params, 'r') as f:
        wavenet_params = json.load(f)

    coord = Coordinator()
    # The reference computes `silence_threshold = None if below EPSILON` and
    # then passes the RAW flag to the reader (train.py:211-220; SURVEY Appendix
    # A: "do not fix silently") -- same here: `--silence_threshold 0` trims with
    # a threshold of 0 (only exactly silent frames go) instead of skipping the
    # trimming.
    silence_threshold = args.silence_threshold
    gc_enabled = args.gc_channels is not None
    if args.synthetic:
        reader = SyntheticReader(args.sample_size,
                                 args.gc_cardinality if gc_enabled else None,
                                 rank=rank)
    else:
        reader = AudioReader(args.data_dir, coord,
                             sample_rate=wavenet_params['sample_rate'],
                             gc_enabled=gc_enabled,
                             sample_size=args.sample_size,
                             silence_threshold=silence_threshold,
                             rank=rank, world=world, seed=rank)

    net = WaveNetModel(
        batch_size=args.batch_size,
        dilations=wavenet_params["dilations"],
        filter_width=wavenet_params["filter_width"],
        residual_channels=wavenet_params["residual_channels"],
        dilation_channels=wavenet_params["dilation_channels"],
        skip_channels=wavenet_params["skip_channels"],
        quantization_channels=wavenet_params["quantization_channels"],
        use_biases=wavenet_params["use_biases"],
        scalar_input=wavenet_params["scalar_input"],
        initial_filter_width=wavenet_params["initial_filter_width"],
        histograms=args.histograms,
        global_condition_channels=args.gc_channels,
        global_condition_cardinality=reader.gc_category_cardinality,
        residual_postproc=wavenet_params.get("residual_postproc", False))
    l2 = args.l2_regularization_strength or None
    optimizer = optimizer_factory[args.optimizer](
        learning_rate=args.learning_rate, momentum=args.momentum)

    try:
        saved_global_step = load(net, restore_from)
        if is_overwritten_training or saved_global_step is None:
            # the first training step will be saved_global_step + 1
            saved_global_step = -1
    except Exception:
        print("Something went wrong while restoring checkpoint. "
              "We will terminate training to avoid accidentally overwriting "
              "the previous model.")
        raise
    parallel.broadcast_parameters(net)
    # every loss() below is followed by optimizer.minimize(): the skip /
    # post-processing gradients' all-reduce MAY start inside the backward pass
    # (opt-in: the schedule has not been measured on an N-GPU RCCL node yet)
    net.dp_overlap_allreduce = bool(args.dp_overlap_allreduce) and world > 1

    threads = reader.start_threads()
    events = None
    if rank == 0:
        os.makedirs(logdir, exist_ok=True)
        events = open(os.path.join(logdir, 'events.jsonl'), 'a')

    step = None
    last_saved_step = saved_global_step
    pending = None            # (step, mean loss tensor, start time) not yet printed
    last_report = [None]

    def report(k, mean_loss, started):
        """Fetch step k's loss (waits for that step), check it, print / log the
        reference's line (train.py:310-311).  sec/step: from the previous line
        (the pipeline's cadence), or from the step's start for the first."""
        # (float(tensor) would wait for EVERYTHING queued on the stream, the
        # next step included: the loss went to a pinned scalar behind an event)
        host_scalar, done = mean_loss
        done.synchronize()
        loss_value = float(host_scalar)
        if not np.isfinite(loss_value):
            # every rank sees the same NaN mean: decide TOGETHER whether a
            # kernel reported an error, so that no rank is left waiting in
            # the next step's collectives
            dev_err = None
            try:
                net.check_device_errors()
            except Exception as e:
                dev_err = e
            if parallel.any_rank(dev_err is not None, net.device):
                raise dev_err or RuntimeError(
                    'rank %d: another rank reported an expired dependency '
                    'wait in a persistent stack launch at step %d'
                    % (rank, k))
        now = time.time()
        duration = now - (last_report[0] if last_report[0] is not None and
                          last_report[0] > started else started)
        last_report[0] = now
        if rank == 0:
            print('step {:d} - loss = {:.3f}, ({:.3f} sec/step)'
                  .format(k, loss_value, duration))
            events.write(json.dumps({'step': k, 'loss': loss_value,
                                     'sec_per_step': duration}) + '\n')
            events.flush()

    fetch_slots = {}

    def fetch_later(t, k):
        """(pinned host scalar, event): the scalar holds t once the event has
        completed; two slots used alternately (a slot is read before the step
        after next overwrites it)."""
        if not t.is_cuda:
            class _Done(object):
                def synchronize(self):
                    pass
            return t.detach().reshape(()).clone(), _Done()
        if (k & 1) not in fetch_slots:
            fetch_slots[k & 1] = (torch.empty((), dtype=torch.float32).pin_memory(),
                                  torch.cuda.Event())
        host_scalar, ev = fetch_slots[k & 1]
        host_scalar.copy_(t.detach().reshape(()).float(), non_blocking=True)
        ev.record()
        return host_scalar, ev

    copy_stream = [None]

    def stage_in(host, k):
        """host [B, n] float tensor -> device tensor, copied on a stream of its
        own: the (pageable, hence host-blocking) copy then does not queue
        behind the previous step's kernels, and the training stream only
        waits for the copy.  (Pinned staging buffers measured 33 instead of
        9.6 ms per step on this platform, tools/h2d_probe.py.)"""
        if copy_stream[0] is None:
            copy_stream[0] = torch.cuda.Stream(device=net.device)
        with torch.cuda.stream(copy_stream[0]):
            dev = host.contiguous().to(net.device)
        torch.cuda.current_stream().wait_stream(copy_stream[0])
        dev.record_stream(torch.cuda.current_stream())
        return dev

    try:
        for step in range(saved_global_step + 1, args.num_steps):
            start_time = time.time()
            # every rank takes the same decision for this step (skip / common
            # clip length / abort) BEFORE any collective of the step is issued
            err = None
            try:
                audio = reader.dequeue(args.batch_size)
                gc = reader.dequeue_gc(args.batch_size) if gc_enabled else None
            except Exception as e:        # e.g. a reader-thread failure
                err, audio, gc = e, None, None
            n_t, all_ok = parallel.agree_step(
                audio.shape[1] if err is None else 0, err is None, net.device)
            if not all_ok:
                raise RuntimeError('rank %d: a rank failed to produce a batch '
                                   'at step %d%s' % (rank, step, '' if err is
                                                     None else ': %r' % err))
            if n_t < 2:
                continue
            audio = audio[:, :n_t]
            if audio.device.type == 'cpu' and net.device.type == 'cuda':
                # pinned staging + asynchronous copy: a pageable host tensor
                # handed to net.loss is copied synchronously BEHIND the previous
                # step's kernels, i.e. the host would wait for the device every
                # step and prepare the next batch while it idles
                audio = stage_in(audio.reshape(audio.shape[0], -1), step)
            trace_step = args.store_metadata and step % 50 == 0
            trace = trace_step and rank == 0
            if trace:
                print('Storing metadata')
                prof = torch.profiler.profile(
                    activities=[torch.profiler.ProfilerActivity.CPU,
                                torch.profiler.ProfilerActivity.CUDA])
                prof.__enter__()
            loss = net.loss(input_batch=audio, global_condition_batch=gc,
                            l2_regularization_strength=l2)
            optimizer.minimize(loss)
            # The reference fetches the loss inside sess.run and so waits for
            # every step (train.py:300-311).  Here the step is queued on the
            # device and its loss is read ONE step later, while the next step
            # runs (the same lines, one step late; 12.0 -> 9.5 ms per step at
            # 8 x 16000: bench.py's step time) -- except where the step's own state is needed at
            # once: a checkpoint step, a traced step, the last step.
            mean_loss = fetch_later(parallel.allreduce_mean_scalar(loss), step)
            if pending is not None:
                report(*pending)
            pending = (step, mean_loss, start_time)
            if trace_step:
                report(*pending)
                pending = None
            if trace:
                prof.__exit__(None, None, None)
                prof.export_chrome_trace(os.path.join(logdir,
                                                      'timeline.trace'))
            if step % args.checkpoint_every == 0:
                # (every rank resolves the step here, so that the ranks'
                # collective sequences stay equal on the error path of report)
                if pending is not None:
                    report(*pending)
                    pending = None
            if rank == 0 and step % args.checkpoint_every == 0:
                save(net, logdir, step)
                last_saved_step = step
                if args.histograms:
                    # the reference's histogram summaries (model.py:314-325)
                    # as an .npz next to the checkpoint
                    hs = net.histogram_summaries()
                    np.savez(os.path.join(
                        logdir, 'histograms-%d.npz' % step),
                        **{k + '/counts': v[0] for k, v in hs.items()},
                        **{k + '/range': np.asarray(v[1:])
                           for k, v in hs.items()})
        if pending is not None:
            report(*pending)
            pending = None
    except KeyboardInterrupt:
        print()
    finally:
        if rank == 0 and step is not None and step > last_saved_step:
            save(net, logdir, step)
        coord.request_stop()
        coord.join(threads)
        if events:
            events.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
