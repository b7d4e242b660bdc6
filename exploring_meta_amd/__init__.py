"""MI355X-native MAML / ANIL inner/outer-loop engine (host side of libmi_maml.so); see DESIGN.md."""
import os as _os

# Side-stream weight gradients + RCCL's own stream need more than HIP's default 4 hardware queues to stay concurrent (see
# bench.py / DESIGN.md); harmless if the process initialised HIP already (then the variable is simply not read).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
