"""Hyper-parameter holder with the reference's names and values (reference ResUnet_a/config.py:3-20)."""
import numpy as np


class UnetConfig(object):
    MEAN = np.array([82, 92, 88], dtype=float)
    CLASSES_NUM = 5
    IMAGE_W, IMAGE_H, IMAGE_C = 512, 512, 3
    EPOCHS = 5000
    batch_size = 8

    def displayConfiguration(self):
        print("\nConfigurations:")
        for name in sorted(n for n in dir(self) if not n.startswith("__") and not callable(getattr(self, n))):
            print(f"{name:30} {getattr(self, name)}")
        print("\n")
