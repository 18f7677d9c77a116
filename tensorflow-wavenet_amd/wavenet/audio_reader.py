"""AudioReader placeholder (reference: wavenet/audio_reader.py:1-193).

Input I/O is outside the hot path named by BASELINE.json (SURVEY.md section
8f-3, "next"); benchmarks and tests use synthetic clips already resident on
the device.  The name is exported so `from wavenet import AudioReader` keeps
working; constructing it says what is missing instead of failing obscurely.
"""


class AudioReader(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            'AudioReader (wav discovery / librosa resampling / silence trim, '
            'reference audio_reader.py) is not part of the MI355X hot path '
            'yet; feed float32 clips in [-1, 1] to WaveNetModel.loss directly')
