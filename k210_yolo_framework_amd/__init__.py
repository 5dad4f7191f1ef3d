"""MI355X-native YOLOv3 detection hot path (conv backbone + region decode/NMS) and its training step.

Host-side mirror of the reference's plugin API; all arithmetic lives in the HIP
library `csrc/libyolo_hip.so` behind the C-ABI declared in `include/yolo_hip.h`.
"""
__version__ = "0.1.0"
