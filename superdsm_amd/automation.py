"""AF_ expansion of scale-dependent hyper-parameters (reference: superdsm/automation.py:71-102).  Automatic scale
estimation (automation.py:41-68) is out of scope: ``AF_scale`` must be given."""


def _expand(cfg, key, factor, default_user_factor, type=None, min=None, max=None):
    *ns, leaf = key.split('/')
    af_key = '/'.join(ns + ['AF_' + leaf])
    cfg.set_default(key, factor * cfg.get(af_key, default_user_factor), True)
    if type is not None:
        cfg.update(key, func=type)
    if min is not None:
        cfg.update(key, func=lambda value: value if value >= min else min)
    if max is not None:
        cfg.update(key, func=lambda value: value if value <= max else max)


def create_config(pipeline, base_cfg, img=None):
    cfg = base_cfg.copy()
    scale = cfg.get('AF_scale', None)
    if scale is None:
        raise ValueError('automatic scale estimation is not part of this package: set AF_scale')
    for stage in pipeline.stages:
        for key, spec in stage.configure(scale).items():
            assert len(spec) in (2, 3), f'{type(stage).__name__}.configure returned tuple of unknown length ({len(spec)})'
            _expand(cfg, f'{stage.cfgns}/{key}', spec[0], spec[1], **(spec[2] if len(spec) == 3 else {}))
    return cfg, scale
