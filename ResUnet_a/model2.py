"""`Resunet_a` with the reference constructor (reference ResUnet_a/model2.py:6-12), built on the MI355X HIP engine."""
from resunet_a_mltsk_keras_amd.keras_api import Resunet_a  # noqa: F401
