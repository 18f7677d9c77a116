"""MI355X-native drop-in for the `wavenet` package of
jyegerlehner/tensorflow-wavenet.

Public names are the reference's export list (wavenet/__init__.py:1-4) plus the
helpers that only exist here (`parallel` for the data-parallel launch,
`WaveNetHipError`).  Importing the package does not need a GPU; every compute
entry point raises `WaveNetHipError` when libwavenet_hip.so or the device is
missing (there is no CPU fallback).
"""
from . import parallel
from ._lib import WaveNetHipError
from .audio_reader import AudioReader
from .model import WaveNetModel
from .ops import batch_to_time
from .ops import causal_conv
from .ops import mu_law_decode
from .ops import mu_law_encode
from .ops import optimizer_factory
from .ops import time_to_batch

__all__ = [
    'WaveNetModel', 'AudioReader', 'mu_law_encode', 'mu_law_decode',
    'time_to_batch', 'batch_to_time', 'causal_conv', 'optimizer_factory',
    'parallel', 'WaveNetHipError',
]
