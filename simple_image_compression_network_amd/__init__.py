"""MI355X-native integer conv/deconv transform path of simple_image_compression_network.

`config` (layer descriptors) imports anywhere; `api` needs the built HIP library (libsicn.so)."""
from .config import LayerDesc, REFERENCE_DESCS, eight_layer_descs  # noqa: F401

__all__ = ["LayerDesc", "REFERENCE_DESCS", "eight_layer_descs"]
