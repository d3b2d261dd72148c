"""MI355X-native SuperDSM hot path."""
import os as _os

# A launch of the engine runs its solve classes on up to four streams (the caller's + three side streams).  The HIP runtime maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that share a queue run one after the other: give
# it room for the caller's own streams as well.  Only effective if this package is imported before the process first touches the GPU;
# an explicit setting of the variable wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
