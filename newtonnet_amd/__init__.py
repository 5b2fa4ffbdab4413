"""newtonnet_amd -- MI355X-native (gfx950) implementation of NewtonNet's per-edge message-passing hot path.

Drop-in surface (mirrors THGLab/NewtonNet v2.1.0):
    newtonnet_amd.models.NewtonNet          <-> newtonnet.models.NewtonNet
    newtonnet_amd.utils.MLAseCalculator     <-> newtonnet.utils.MLAseCalculator
    newtonnet_amd.layers.*                  <-> newtonnet.layers.* (factories / parameter holders)
All arithmetic runs in libnewtonnet_hip.so (newtonnet_amd/csrc, C ABI in include/newtonnet_hip.h).
"""
__version__ = '0.1.0'

from newtonnet_amd.models import NewtonNet  # noqa: E402,F401
