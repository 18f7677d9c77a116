"""Background audio reader: counterpart of the reference's
wavenet/audio_reader.py (find *.wav recursively, load + resample to mono
float32, RMS silence trim, cut into `sample_size` pieces, speaker id from
`p<id>_<rec>.wav`), feeding the MI355X training loop instead of a
tf.PaddingFIFOQueue.

Same constructor arguments and method names as the reference
(`AudioReader(audio_dir, coord, sample_rate, gc_enabled, sample_size,
silence_threshold, queue_size)`, `dequeue`, `dequeue_gc`, `start_threads`,
`gc_category_cardinality`); new keyword arguments `rank` / `world` shard the
file list per data-parallel rank, `seed` makes the shuffle reproducible.

librosa is not available in this image: wav I/O is scipy.io.wavfile,
resampling is scipy.signal.resample_poly (librosa's default is a Kaiser-windowed
sinc; the two differ at the 1e-3 level, which only matters if one wants
sample-identical pieces), framing of the RMS energy follows librosa's defaults
(frame 2048, hop 512, centred with reflect padding).
"""
import fnmatch
import os
import queue
import random
import re
import threading
from fractions import Fraction

import numpy as np

_ID_RE = re.compile(r'p([0-9]+)_([0-9]+)\.wav')


class Coordinator(object):
    """Minimal stand-in for tf.train.Coordinator (should_stop/request_stop)."""

    def __init__(self):
        self._stop = threading.Event()

    def should_stop(self):
        return self._stop.is_set()

    def request_stop(self):
        self._stop.set()

    def join(self, threads=(), timeout=5.0):
        for t in threads:
            t.join(timeout)


def find_files(directory, pattern='*.wav'):
    '''Recursively finds all files matching the pattern (sorted).'''
    found = []
    for root, _, names in os.walk(directory):
        for name in fnmatch.filter(names, pattern):
            found.append(os.path.join(root, name))
    return sorted(found)


def category_id_of(filename):
    """Speaker id of a VCTK-style file name `p<id>_<rec>.wav`, else None."""
    m = _ID_RE.findall(os.path.basename(filename))
    return int(m[0][0]) if m else None


def get_category_cardinality(files):
    """(min id, max id) over the files that carry an id."""
    ids = [category_id_of(f) for f in files]
    ids = [i for i in ids if i is not None]
    if not ids:
        return None, None
    return min(ids), max(ids)


def not_all_have_id(files):
    return any(category_id_of(f) is None for f in files)


def load_wav(filename, sample_rate):
    """Mono float32 waveform in [-1, 1] at `sample_rate`."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    sr, data = wavfile.read(filename)
    if data.dtype == np.int16:
        audio = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        audio = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        audio = (data.astype(np.float32) - 128.0) / 128.0
    else:
        audio = data.astype(np.float32)
    if audio.ndim > 1:
        audio = audio.mean(axis=1)
    if sr != sample_rate:
        frac = Fraction(int(sample_rate), int(sr)).limit_denominator(1000)
        audio = resample_poly(audio, frac.numerator, frac.denominator)
    return np.ascontiguousarray(audio, dtype=np.float32)


def rms_energy(audio, frame_length=2048, hop_length=512):
    """Frame-wise RMS, centred frames with reflect padding."""
    audio = np.asarray(audio, dtype=np.float32)
    if audio.size == 0:
        return np.zeros(0, np.float32)
    pad = frame_length // 2
    mode = 'reflect' if audio.size > pad else 'edge'
    y = np.pad(audio, pad, mode=mode)
    n = 1 + (y.size - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    return np.sqrt(np.mean(y[idx] ** 2, axis=1))


def trim_silence(audio, threshold, frame_length=2048, hop_length=512):
    '''Removes silence at the beginning and end of a sample.'''
    energy = rms_energy(audio, frame_length, hop_length)
    frames = np.nonzero(energy > threshold)[0]
    if frames.size == 0:
        return audio[0:0]
    lo, hi = frames[0] * hop_length, frames[-1] * hop_length
    return audio[lo:hi]


def load_generic_audio(files, sample_rate, rng):
    '''Yields (audio [T,1], filename, category_id) in a shuffled order.'''
    order = list(files)
    rng.shuffle(order)
    for filename in order:
        audio = load_wav(filename, sample_rate)
        yield audio.reshape(-1, 1), filename, category_id_of(filename)


class AudioReader(object):
    '''Generic background audio reader that preprocesses audio files and
    queues fixed-size pieces for the training loop.'''

    def __init__(self,
                 audio_dir,
                 coord,
                 sample_rate,
                 gc_enabled,
                 sample_size=None,
                 silence_threshold=None,
                 queue_size=32,
                 rank=0,
                 world=1,
                 seed=None):
        self.audio_dir = audio_dir
        self.sample_rate = sample_rate
        self.coord = coord if coord is not None else Coordinator()
        self.sample_size = sample_size
        self.silence_threshold = silence_threshold
        self.gc_enabled = gc_enabled
        self.threads = []
        self.queue = queue.Queue(maxsize=queue_size)
        self.gc_queue = queue.Queue(maxsize=queue_size) if gc_enabled else None
        self._rng = random.Random(seed)

        files = find_files(audio_dir)
        if not files:
            raise ValueError("No audio files found in '{}'.".format(audio_dir))
        if self.gc_enabled and not_all_have_id(files):
            raise ValueError("Global conditioning is enabled, but file names "
                             "do not conform to pattern having id.")
        if self.gc_enabled:
            # zero-indexed embedding table: largest id + 1 categories
            _, max_id = get_category_cardinality(files)
            self.gc_category_cardinality = max_id + 1
            print("Detected --gc_cardinality={}".format(
                self.gc_category_cardinality))
        else:
            self.gc_category_cardinality = None
        # data-parallel sharding of the file list (new; the reference is
        # single-process)
        self.files = files[rank::world] if world > 1 else files
        if not self.files:
            raise ValueError('rank %d of %d has no audio files' % (rank, world))

    # ------------------------------------------------------------- consumer
    def _get(self, q):
        while True:
            try:
                return q.get(timeout=0.2)
            except queue.Empty:
                if self.coord.should_stop() and q.empty():
                    raise RuntimeError('AudioReader stopped')
                if self.threads and not any(t.is_alive() for t in self.threads):
                    raise RuntimeError('AudioReader thread died')

    def dequeue(self, num_elements):
        """float32 tensor [num_elements, T_max, 1], shorter pieces zero-padded
        at the end (tf.PaddingFIFOQueue.dequeue_many semantics)."""
        import torch
        pieces = [self._get(self.queue) for _ in range(num_elements)]
        tmax = max(p.shape[0] for p in pieces)
        out = np.zeros((num_elements, tmax, 1), np.float32)
        for i, p in enumerate(pieces):
            out[i, :p.shape[0], :] = p
        return torch.from_numpy(out)

    def dequeue_gc(self, num_elements):
        import torch
        ids = [self._get(self.gc_queue) for _ in range(num_elements)]
        return torch.tensor(ids, dtype=torch.int32)

    # ------------------------------------------------------------- producer
    def _put(self, q, item):
        while not self.coord.should_stop():
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                continue
        return False

    def thread_main(self, sess=None):
        buffer_ = np.zeros((0,), np.float32)
        while not self.coord.should_stop():      # many passes over the data
            for audio, filename, category_id in load_generic_audio(
                    self.files, self.sample_rate, self._rng):
                if self.coord.should_stop():
                    return
                if self.silence_threshold is not None:
                    audio = trim_silence(audio[:, 0], self.silence_threshold)
                    audio = audio.reshape(-1, 1)
                    if audio.size == 0:
                        print("Warning: {} was ignored as it contains only "
                              "silence. Consider decreasing trim_silence "
                              "threshold, or adjust volume of the audio."
                              .format(filename))
                if self.sample_size:
                    # cut into fixed-size pieces (the last piece of a file is
                    # short; consecutive files are concatenated like the
                    # reference's running buffer)
                    buffer_ = np.append(buffer_, audio)
                    while len(buffer_) > 0:
                        piece = buffer_[:self.sample_size].reshape(-1, 1)
                        if not self._put(self.queue, piece.copy()):
                            return
                        buffer_ = buffer_[self.sample_size:]
                        if self.gc_enabled and not self._put(self.gc_queue,
                                                             category_id):
                            return
                elif audio.size:
                    if not self._put(self.queue, audio):
                        return
                    if self.gc_enabled and not self._put(self.gc_queue,
                                                         category_id):
                        return

    def start_threads(self, sess=None, n_threads=1):
        for _ in range(n_threads):
            thread = threading.Thread(target=self.thread_main, args=(sess,))
            thread.daemon = True  # Thread will close when parent quits.
            thread.start()
            self.threads.append(thread)
        return self.threads
