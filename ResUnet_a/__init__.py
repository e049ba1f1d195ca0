"""Drop-in package name of the reference (`from ResUnet_a.model2 import Resunet_a`, train_ISPRS.py:4)."""
