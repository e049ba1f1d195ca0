"""`Resunet_a` of the earlier graph variant (reference ResUnet_a/model.py:6-12: no skip term in the ResBlock sum, no BN on
1x1 convolutions, conv-then-upsample decoder), same constructor."""
from resunet_a_mltsk_keras_amd.keras_api import Resunet_a_v1 as Resunet_a  # noqa: F401
