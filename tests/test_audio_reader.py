"""AudioReader counterpart (reference: wavenet/audio_reader.py; the reference
has no test for it).  CPU only: wav files are synthesised with scipy."""
import os

import numpy as np
import pytest
from scipy.io import wavfile

from util import PKG  # noqa: F401  (sys.path)
from wavenet import audio_reader as ar


def make_wavs(root, rate=16000):
    os.makedirs(os.path.join(root, 'p225'))
    os.makedirs(os.path.join(root, 'p226'))
    t = np.arange(rate) / rate
    tone = 0.5 * np.sin(2 * np.pi * 220 * t)
    quiet = np.zeros(rate // 4)
    sig = np.concatenate([quiet, tone, quiet])
    files = {}
    for spk, rec, scale in ((225, 1, 1.0), (225, 2, 0.8), (226, 1, 0.6)):
        path = os.path.join(root, 'p%d' % spk, 'p%d_%03d.wav' % (spk, rec))
        wavfile.write(path, rate, (sig * scale * 32767).astype(np.int16))
        files[path] = scale
    return files


def test_find_files_and_ids(tmp_path):
    files = make_wavs(str(tmp_path))
    found = ar.find_files(str(tmp_path))
    assert sorted(files) == found
    assert [ar.category_id_of(f) for f in found] == [225, 225, 226]
    assert ar.get_category_cardinality(found) == (225, 226)
    assert not ar.not_all_have_id(found)
    assert ar.not_all_have_id(found + ['/x/other.wav'])
    assert ar.category_id_of('/x/other.wav') is None


def test_load_wav_resamples_and_normalises(tmp_path):
    files = make_wavs(str(tmp_path), rate=8000)
    path = sorted(files)[0]
    a8 = ar.load_wav(path, 8000)
    a16 = ar.load_wav(path, 16000)
    assert a8.dtype == np.float32 and abs(np.abs(a8).max() - 0.5) < 1e-3
    assert abs(len(a16) - 2 * len(a8)) <= 1
    # stereo -> mono
    st = np.stack([a8, -a8 * 0.5], 1)
    p2 = str(tmp_path / 'p300_001.wav')
    wavfile.write(p2, 8000, st.astype(np.float32))
    m = ar.load_wav(p2, 8000)
    assert m.ndim == 1 and np.allclose(m, a8 * 0.25, atol=1e-6)


def test_trim_silence():
    rate = 16000
    sig = np.concatenate([np.zeros(4000), 0.5 * np.sin(
        2 * np.pi * 220 * np.arange(8000) / rate), np.zeros(4000)]
    ).astype(np.float32)
    out = ar.trim_silence(sig, 0.1)
    assert 6000 < len(out) < 9500
    assert np.abs(out).max() > 0.4
    assert len(ar.trim_silence(np.zeros(5000, np.float32), 0.1)) == 0
    e = ar.rms_energy(sig)
    assert e.shape[0] == 1 + len(sig) // 512


def test_reader_pieces_padding_and_gc(tmp_path):
    make_wavs(str(tmp_path))
    reader = ar.AudioReader(str(tmp_path), None, sample_rate=16000,
                            gc_enabled=True, sample_size=5000,
                            silence_threshold=0.1, queue_size=8, seed=0)
    assert reader.gc_category_cardinality == 227      # max id + 1
    reader.start_threads()
    batch = reader.dequeue(4)
    ids = reader.dequeue_gc(4)
    assert tuple(batch.shape)[0] == 4 and batch.shape[2] == 1
    assert batch.shape[1] <= 5000 and batch.dtype.is_floating_point
    assert set(ids.tolist()) <= {225, 226}
    assert float(batch.abs().max()) <= 1.0
    reader.coord.request_stop()


def test_reader_sharding_and_errors(tmp_path):
    make_wavs(str(tmp_path))
    r0 = ar.AudioReader(str(tmp_path), None, 16000, False, 4000, rank=0,
                        world=2)
    r1 = ar.AudioReader(str(tmp_path), None, 16000, False, 4000, rank=1,
                        world=2)
    assert len(r0.files) == 2 and len(r1.files) == 1
    assert not set(r0.files) & set(r1.files)
    with pytest.raises(ValueError, match='No audio files'):
        ar.AudioReader(str(tmp_path / 'p225' / 'nothing'), None, 16000, False)
    wavfile.write(str(tmp_path / 'noid.wav'), 16000,
                  np.zeros(100, np.int16))
    with pytest.raises(ValueError, match='do not conform'):
        ar.AudioReader(str(tmp_path), None, 16000, True)
