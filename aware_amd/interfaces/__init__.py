"""Abstract call shapes of the plug-in seams (reference: src/AWARE/interfaces/*.py).

They define the drop-in boundary: a DSP plug-in is a callable tensor -> tensor held in an
ordered list, an embedder has .embed, a detector has .detect, a loss has .forward, an attack
has .apply(audio, sr) and .name."""
from abc import ABC, abstractmethod


class BaseAudioProcessor(ABC):
    """interfaces/audio.py:6-9"""

    @abstractmethod
    def __call__(self, data, *args, **kwargs):
        ...


class BasePatternProcessor(ABC):
    """interfaces/watermark.py:5-8"""

    @abstractmethod
    def __call__(self, data, *args, **kwargs):
        ...


class BaseEmbedder(ABC):
    """interfaces/embedding.py:5-8"""

    @abstractmethod
    def embed(self, audio, sample_rate, watermark):
        ...


class BaseDetector(ABC):
    """interfaces/detection.py:11-14"""

    @abstractmethod
    def detect(self, audio, sampling_rate):
        ...


class BaseDetectorNet(ABC):
    """interfaces/detection.py:6-9 (an nn.Module in the reference; here the weights are frozen
    host arrays plus a device handle, there is nothing to train)."""

    @abstractmethod
    def forward(self, x):
        ...

    def __call__(self, x):
        return self.forward(x)


class Loss(ABC):
    """interfaces/loss.py:4-22"""

    @abstractmethod
    def forward(self, predicted, target_pattern):
        ...

    def __call__(self, predicted, target_pattern):
        return self.forward(predicted, target_pattern)


class BaseMetrics(ABC):
    """interfaces/metrics.py:4-7"""

    @abstractmethod
    def __call__(self, output, target):
        ...
