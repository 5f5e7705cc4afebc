"""MI355X-native UNet_Nested (UNet++) forward/backward path.

Drop-in for ``models.UNet_Nested`` of unanan/UNet_Nested4Tiny_Objects_Keypoints
(/root/reference/models/unet.py:204-300; resolved by name as trainer/trainer.py:337 does):

    import unet_nested4tiny_objects_keypoints_amd as models
    net = getattr(models, "UNet_Nested")().to("cuda")

All arithmetic of the path runs in the in-tree HIP library (csrc/ -> libunetpp_hip.so, C ABI in
include/unetpp_hip.h).  There is no CPU or eager-PyTorch fallback: using the model without the
library, or with CPU tensors, raises.
"""
from .unet import UNet, UNet_Nested, count_param  # noqa: F401
from .losses import FocalLoss_BCE_2d  # noqa: F401
from .step import train_step  # noqa: F401
from .targets import create_heatmap  # noqa: F401
from .keypoints import Heatmap  # noqa: F401
from .serving import GraphedForward  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401

__all__ = ["UNet_Nested", "UNet", "count_param", "FocalLoss_BCE_2d", "train_step", "create_heatmap", "Heatmap",
           "GraphedForward", "GraphedTrainStep"]
