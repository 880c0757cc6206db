"""``task._name`` classes of the shipped YAMLs (reference msmctts/tasks/msmc_tts.py:10-151).
On the training path they are only the container; inference glue is out of scope."""
from .base_task import BaseTask


class MSMCTTS(BaseTask):
    """``task._name: MSMCTTS`` of the shipped YAMLs; on the training path it is only the container."""


class TTS(MSMCTTS):
    pass
