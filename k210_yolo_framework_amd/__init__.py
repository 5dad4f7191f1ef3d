"""MI355X-native YOLOv3 detection hot path (conv backbone + region decode/NMS) and its training step.

Host-side mirror of the reference's plugin API; all arithmetic lives in the HIP
library `csrc/libyolo_hip.so` behind the C-ABI declared in `include/yolo_hip.h`.
"""
__version__ = "0.1.0"

import os as _os

# Several batches in flight (engine.Pipeline) want one HARDWARE queue per HIP stream.  The ROCm runtime multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES queues (default 4, counted with the null stream and every pooled torch stream): with an unlucky creation order
# two of three pipeline streams share a queue and the third batch in flight buys nothing (measured: 51 k instead of 68 k images/s on the
# same box; 8 queues: always distinct, no effect on the one-batch rate).  Read by the runtime when it first touches the device, so it
# has to be in the environment before the first HIP call of the process; a value the user has set is left alone.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
