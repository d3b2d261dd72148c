"""MI355X-native SuperDSM hot path."""
import logging as _logging
import os as _os

# A launch of the engine runs its solve classes on up to four streams (the caller's + three side streams).  The HIP runtime maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that share a queue run one after the other: give
# it room for the caller's own streams as well.  Only effective if this package is imported before the process first touches the GPU;
# an explicit setting of the variable wins, SDSM_SET_HW_QUEUES=0 leaves the environment alone.  The library checks at its first launch
# whether the streams really run side by side (sdsm_side_queues_distinct) and says so once on stderr if they do not.
if 'GPU_MAX_HW_QUEUES' not in _os.environ and _os.environ.get('SDSM_SET_HW_QUEUES', '1') != '0':
    _os.environ['GPU_MAX_HW_QUEUES'] = '8'
    _logging.getLogger(__name__).info('GPU_MAX_HW_QUEUES=8 set for this process (SDSM_SET_HW_QUEUES=0: leave the environment alone)')
