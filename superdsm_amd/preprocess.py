"""``preprocess`` stage (reference: superdsm/preprocess.py:34-73): ``g_raw`` -> ``y`` on the GPU."""
import math

from . import engine
from .pipeline import Stage


class Preprocessing(Stage):

    ENABLED_BY_DEFAULT = True

    def __init__(self):
        super().__init__('preprocess', inputs=['g_raw'], outputs=['y'])

    def process(self, input_data, cfg, out, log_root_dir):
        y = engine.preprocess(input_data['g_raw'],
                              sigma1=cfg.get('sigma1', math.sqrt(2)),
                              sigma2=cfg.get('sigma2', 40),
                              offset_clip=cfg.get('offset_clip', 3),
                              lower_clip_mean=cfg.get('lower_clip_mean', False))
        return {'y': y}

    def configure_ex(self, scale, radius, diameter):
        return {'sigma2': (scale, 1.0)}
