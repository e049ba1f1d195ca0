"""Names train_ISPRS.py imports from `utils` that belong to the ResUnet-a training path (reference utils.py:7-24,466-491).
The U-Net / ResNet50 baselines and the numpy patch plumbing of the reference's utils.py are out of scope."""
import numpy as np  # noqa: F401

from resunet_a_mltsk_keras_amd.keras_api import (  # noqa: F401
    SGD, Adam, K, load_model, weighted_categorical_crossentropy)


def unet(*_a, **_k):
    raise NotImplementedError("the baseline U-Net (--resunet_a False) is not part of the accelerated path")
