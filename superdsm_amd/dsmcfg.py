"""``dsm/*`` hyper-parameters (reference: superdsm/dsmcfg.py:6-21, 79-97): the operator's knob surface, kept
verbatim as API.  Keys that only concern the CPU implementation (cachesize, cachetest,
smooth_mat_max_allocations, smooth_mat_dtype, cp_timeout) are accepted and ignored by the GPU path."""
import numpy as np

from .pipeline import Stage

DSM_CONFIG_DEFAULTS = {
    'cachesize': 1,
    'cachetest': None,
    'sparsity_tol': 0,
    'init': 'elliptical',
    'smooth_amount': 10,
    'epsilon': 1.0,
    'alpha': 0.5,
    'scale': 1000,
    'smooth_subsample': 20,
    'gaussian_shape_multiplier': 2,
    'smooth_mat_dtype': 'float32',
    'smooth_mat_max_allocations': np.inf,
    'background_margin': 20,
    'cp_timeout': 300,
}


class DSM_Config(Stage):
    """Publishes the ``dsm`` namespace as the pipeline output ``dsm_cfg``."""

    ENABLED_BY_DEFAULT = True

    def __init__(self):
        super().__init__('dsm', inputs=[], outputs=['dsm_cfg'])

    def process(self, input_data, cfg, out, log_root_dir):
        return {'dsm_cfg': {key: cfg.get(key, default) for key, default in DSM_CONFIG_DEFAULTS.items()}}

    def configure_ex(self, scale, radius, diameter):
        return {
            'alpha': (scale ** 2, 0.0005),
            'smooth_amount': (scale, 0.2, dict(type=int, min=4)),
            'smooth_subsample': (scale, 0.4, dict(type=int, min=8)),
            'background_margin': (scale, 0.4, dict(type=int, min=8)),
        }
