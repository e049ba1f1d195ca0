"""Loss factory of the reference's multitask path (reference multitasking_utils.py:38-85), evaluated by HIP kernels."""
from resunet_a_mltsk_keras_amd.keras_api import Tanimoto_dual_loss, TanimotoDualLoss  # noqa: F401


from resunet_a_mltsk_keras_amd.keras_api import Tanimoto_loss  # noqa: E402,F401


# Label synthesis of the reference module (multitasking_utils.py:6-35), without cv2: see resunet_a_mltsk_keras_amd/labels.py
from resunet_a_mltsk_keras_amd.labels import get_boundary_label, get_distance_label  # noqa: E402,F401
