"""``EnvMode`` (reference ``fluidgym/types.py:15-20``)."""
from enum import Enum


class EnvMode(Enum):
    TRAIN = "train"
    VAL = "val"
    TEST = "test"
