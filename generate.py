#!/usr/bin/env python
"""Generation script: MI355X counterpart of the reference's generate.py.

Same flags and defaults (generate.py:38-116); the per-sample
`sess.run` loop of the reference (generate.py:213-241) is one persistent HIP
kernel (`WaveNetModel.generate`): priming with a --wav_seed (generate.py:195-210,
one step per seed sample) and temperature sampling happen on the device.
--fast_generation false uses the windowed naive path (`predict_proba`, host-side
np.random.choice) like the reference.  wav I/O is scipy (librosa is absent).
"""
from __future__ import division
from __future__ import print_function

import argparse
import json
import os
import sys
from datetime import datetime

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-wavenet_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from wavenet import tf_checkpoint  # noqa: E402

SAMPLES = 16000
TEMPERATURE = 1.0
LOGDIR = './logdir'
WINDOW = 8000
WAVENET_PARAMS = './wavenet_params.json'
SAVE_EVERY = None
SILENCE_THRESHOLD = 0.1


def _str_to_bool(s):
    if s.lower() not in ('true', 'false'):
        raise ValueError('Argument needs to be a boolean, got {}'.format(s))
    return s.lower() == 'true'


def _ensure_positive_float(f):
    if float(f) <= 0:
        raise argparse.ArgumentTypeError('Argument must be greater than zero')
    return float(f)


def get_arguments(argv=None):
    p = argparse.ArgumentParser(description='WaveNet generation script')
    p.add_argument('checkpoint', type=str,
                   help='Which model checkpoint to generate from')
    p.add_argument('--samples', type=int, default=SAMPLES)
    p.add_argument('--temperature', type=_ensure_positive_float,
                   default=TEMPERATURE)
    p.add_argument('--logdir', type=str, default=LOGDIR)
    p.add_argument('--window', type=int, default=WINDOW,
                   help='Past samples taken into account per step (naive '
                   'path) / kept of the seed')
    p.add_argument('--wavenet_params', type=str, default=WAVENET_PARAMS)
    p.add_argument('--wav_out_path', type=str, default=None)
    p.add_argument('--save_every', type=int, default=SAVE_EVERY)
    p.add_argument('--fast_generation', type=_str_to_bool, default=True)
    p.add_argument('--wav_seed', type=str, default=None)
    p.add_argument('--gc_channels', type=int, default=None)
    p.add_argument('--gc_cardinality', type=int, default=None)
    p.add_argument('--gc_id', type=int, default=None)
    p.add_argument('--seed', type=int, default=0, help='sampling RNG seed')
    a = p.parse_args(argv)
    if a.gc_channels is not None:
        if a.gc_cardinality is None:
            raise ValueError("Globally conditioning but gc_cardinality not "
                             "specified. Use --gc_cardinality=377 for full "
                             "VCTK corpus.")
        if a.gc_id is None:
            raise ValueError("Globally conditioning, but global condition was "
                             "not specified. Use --gc_id to specify global "
                             "condition.")
    return a


def write_wav(waveform, sample_rate, filename):
    from scipy.io import wavfile
    wavfile.write(filename, int(sample_rate),
                  np.asarray(waveform, dtype=np.float32))
    print('Updated wav file at {}'.format(filename))


def create_seed(filename, sample_rate, quantization_channels,
                window_size=WINDOW, silence_threshold=SILENCE_THRESHOLD):
    """mu-law codes of the (silence-trimmed) seed wav, cut to the window."""
    from wavenet import mu_law_encode
    from wavenet.audio_reader import load_wav, trim_silence
    audio = trim_silence(load_wav(filename, sample_rate), silence_threshold)
    quantized = mu_law_encode(audio, quantization_channels)
    return quantized[:min(int(quantized.numel()), window_size)]


def main(argv=None):
    args = get_arguments(argv)
    from wavenet import WaveNetModel, mu_law_decode
    started = "{0:%Y-%m-%dT%H-%M-%S}".format(datetime.now())
    logdir = os.path.join(args.logdir, 'generate', started)
    with open(args.wavenet_params, 'r') as f:
        wavenet_params = json.load(f)
    net = WaveNetModel(
        batch_size=1,
        dilations=wavenet_params['dilations'],
        filter_width=wavenet_params['filter_width'],
        residual_channels=wavenet_params['residual_channels'],
        dilation_channels=wavenet_params['dilation_channels'],
        quantization_channels=wavenet_params['quantization_channels'],
        skip_channels=wavenet_params['skip_channels'],
        use_biases=wavenet_params['use_biases'],
        scalar_input=wavenet_params['scalar_input'],
        initial_filter_width=wavenet_params['initial_filter_width'],
        global_condition_channels=args.gc_channels,
        global_condition_cardinality=args.gc_cardinality,
        residual_postproc=wavenet_params.get("residual_postproc", False))
    print('Restoring model from {}'.format(args.checkpoint))
    if tf_checkpoint.checkpoint_format(args.checkpoint):
        # a checkpoint written by the reference itself (tf.train.Saver)
        tf_checkpoint.load_into(net, args.checkpoint)
    else:
        net.load_state_dict(torch.load(args.checkpoint,
                                       map_location='cpu')['variables'])
    Q = wavenet_params['quantization_channels']
    rate = wavenet_params['sample_rate']
    gc = None if args.gc_id is None else [args.gc_id]
    if args.wav_seed:
        waveform = create_seed(args.wav_seed, rate, Q,
                               args.window).cpu().numpy().tolist()
    else:
        waveform = np.random.default_rng(args.seed).integers(
            Q, size=(1,)).tolist()

    def dump(codes):
        if args.wav_out_path:
            out = mu_law_decode(np.asarray(codes, np.int32), Q).cpu().numpy()
            write_wav(out, rate, args.wav_out_path)

    if args.fast_generation:
        if args.wav_seed:
            print('Priming generation with {} seed samples...'
                  .format(len(waveform)))
        chunk = args.save_every or args.samples
        done = 0
        # first call primes (teacher-forced steps) and starts drawing; later
        # chunks continue from the device-resident queues
        codes = net.generate(min(chunk, args.samples), seed_samples=waveform,
                             temperature=args.temperature,
                             global_condition=gc, seed=args.seed)
        waveform = codes.cpu().numpy().tolist()
        done += min(chunk, args.samples)
        if args.save_every and done < args.samples:
            dump(waveform)
        while done < args.samples:
            n = min(chunk, args.samples - done)
            more = net.continue_generation(n, waveform[-1], args.temperature,
                                           gc, args.seed)
            waveform.extend(more.cpu().numpy().tolist())
            done += n
            print('Sample {:3<d}/{:3<d}'.format(done, args.samples), end='\r')
            if args.save_every and done < args.samples:
                dump(waveform)
    else:
        rng = np.random.default_rng(args.seed)
        # one workspace for the whole window: the growing inputs of the first
        # `window` steps are views of it
        net.reserve(1, min(args.window, len(waveform) + args.samples))
        for step in range(args.samples):
            window = waveform[-args.window:] if len(waveform) > args.window \
                else waveform
            prediction = net.predict_proba(np.asarray(window), gc
                                           ).cpu().numpy().astype(np.float64)
            # temperature (generate.py:229-233)
            with np.errstate(divide='ignore'):
                scaled = np.log(prediction) / args.temperature
            scaled = np.exp(scaled - np.logaddexp.reduce(scaled))
            if args.temperature == 1.0:
                np.testing.assert_allclose(
                    prediction, scaled, atol=1e-5,
                    err_msg='Prediction scaling at temperature=1.0 is not '
                            'working as intended.')
            waveform.append(int(rng.choice(np.arange(Q), p=scaled / scaled.sum())))
            if (step + 1) % 100 == 0:
                print('Sample {:3<d}/{:3<d}'.format(step + 1, args.samples),
                      end='\r')
            if (args.wav_out_path and args.save_every and
                    (step + 1) % args.save_every == 0):
                dump(waveform)
    print()
    os.makedirs(logdir, exist_ok=True)
    np.save(os.path.join(logdir, 'generated_codes.npy'),
            np.asarray(waveform, np.int32))
    dump(waveform)
    print('Finished generating. Codes saved under {}.'.format(logdir))
    return 0


if __name__ == '__main__':
    sys.exit(main())
