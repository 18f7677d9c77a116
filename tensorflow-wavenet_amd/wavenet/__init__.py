"""MI355X-native drop-in for the `wavenet` package of
jyegerlehner/tensorflow-wavenet (export list: wavenet/__init__.py:1-4)."""
from .model import WaveNetModel
from .audio_reader import AudioReader
from .ops import (mu_law_encode, mu_law_decode, time_to_batch,
                  batch_to_time, causal_conv, optimizer_factory)
