"""Stage / Pipeline plugin API (reference: superdsm/pipeline.py:10-265), restated so that the GPU stages drop
into a SuperDSM-style pipeline: same constructor, ``process`` contract, ``configure_ex`` factors, callbacks,
``first_stage`` / ``last_stage`` re-entry and input/output driven ordering."""
import math
import os
import time

from .image import normalize_image
from .output import get_output


class Stage:
    """A pipeline stage: declares ``inputs`` / ``outputs`` and implements ``process``."""

    ENABLED_BY_DEFAULT = True

    def __init__(self, name, cfgns=None, inputs=(), outputs=()):
        self.name = name
        self.cfgns = name if cfgns is None else cfgns
        self.inputs = {key: key for key in inputs}
        self.outputs = {key: key for key in outputs}
        self._callbacks = {}

    def add_callback(self, name, cb):
        self._callbacks.setdefault(name, []).append(cb)

    def remove_callback(self, name, cb):
        if name in self._callbacks:
            self._callbacks[name].remove(cb)

    def _callback(self, name, *args, **kwargs):
        for cb in self._callbacks.get(name, ()):
            cb(name, *args, **kwargs)

    def __call__(self, data, cfg, out=None, log_root_dir=None):
        out = get_output(out)
        cfg = cfg.get(self.cfgns, {})
        if not cfg.get('enabled', self.ENABLED_BY_DEFAULT):
            out.write(f'Skipping disabled stage "{self.name}"')
            self._callback('skip', data)
            return 0
        out.intermediate(f'Starting stage "{self.name}"')
        self._callback('start', data)
        stage_input = {inner: data[outer] for outer, inner in self.inputs.items()}
        t0 = time.time()
        produced = self.process(stage_input, cfg=cfg, out=out, log_root_dir=log_root_dir)
        dt = time.time() - t0
        assert set(produced.keys()) == set(self.outputs), 'stage "%s" generated unexpected output' % self.name
        for inner, outer in self.outputs.items():
            data[outer] = produced[inner]
        self._callback('end', data)
        return dt

    def process(self, input_data, cfg, out, log_root_dir):
        raise NotImplementedError()

    def configure(self, scale):
        radius = scale * math.sqrt(2)
        return self.configure_ex(scale, radius, 2 * radius)

    def configure_ex(self, scale, radius, diameter):
        return {}


class Pipeline:

    def __init__(self):
        self.stages = []

    def find(self, stage_name, not_found_dummy=float('inf')):
        names = [stage.name for stage in self.stages]
        return names.index(stage_name) if stage_name in names else not_found_dummy

    def append(self, stage, after=None):
        if after is None:
            self.stages.append(stage)
        else:
            pos = self.find(after) if isinstance(after, str) else after
            self.stages.insert(pos + 1, stage)

    def init(self, g_raw, cfg):
        data = {}
        if cfg.get('histological', False):
            data['g_rgb'] = g_raw
            g_raw = g_raw.mean(axis=2)
            g_raw = g_raw.max() - g_raw
        data['g_raw'] = normalize_image(g_raw)
        return data

    def process_image(self, g_raw, cfg, first_stage=None, last_stage=None, data=None, out=None, log_root_dir=None):
        cfg = cfg.copy()
        if log_root_dir is not None:
            os.makedirs(log_root_dir, exist_ok=True)
        if first_stage == self.stages[0].name and data is None:
            first_stage = None
        if first_stage is not None and first_stage.endswith('+'):
            first_stage = self.stages[1 + self.find(first_stage[:-1])].name
        if first_stage is not None and last_stage is not None and self.find(first_stage) > self.find(last_stage):
            return data, cfg, {}
        out = get_output(out)
        running = first_stage is None
        if running:
            data = self.init(g_raw, cfg)
        else:
            assert data is not None, 'data argument must be provided if first_stage is used'
        timings = {}
        for stage in self.stages:
            if not running and stage.name == first_stage:
                running = True
            if running:
                timings[stage.name] = stage(data, cfg, out=out, log_root_dir=log_root_dir)
            if stage.name == last_stage:
                running = False
        return data, cfg, timings


def create_pipeline(stages):
    """Orders ``stages`` so that every stage's inputs are produced before it runs (``g_raw`` is given)."""
    available = {'g_raw'}
    pending = list(stages)
    pipeline = Pipeline()
    while pending:
        ready = next((s for s in pending if set(s.inputs) <= available), None)
        if ready is None:
            raise ValueError('failed to resolve total ordering')
        pending.remove(ready)
        pipeline.append(ready)
        available |= set(ready.outputs)
    return pipeline


def create_default_pipeline(extra_stages=()):
    """Preprocessing -> DSM_Config -> [extra stages, e.g. a region-analysis stage producing ``atoms`` and
    ``adjacencies``] -> GlobalEnergyMinimization.  The reference's C2F region analysis and post-processing stages
    stay on the host and are not part of this package."""
    from .dsmcfg import DSM_Config
    from .globalenergymin import GlobalEnergyMinimization
    from .preprocess import Preprocessing
    return create_pipeline([Preprocessing(), DSM_Config(), *extra_stages, GlobalEnergyMinimization()])
