"""Hyper-parameter container with the reference's ``Config`` API (superdsm/config.py:9-187): nested dictionaries
addressed by ``a/b/c`` keys.  Behavioural details that callers rely on are kept: ``get`` INSERTS the default
(config.py:80-81), ``set_default(..., override_none=True)`` replaces ``None`` entries, ``Config(other_config)``
deep-copies while ``Config(dict)`` wraps without copying, ``md5`` hashes the JSON dump."""
import hashlib
import json


def _plain(value):
    return value.entries if isinstance(value, Config) else value


class Config:

    def __init__(self, other=None):
        if other is None:
            self.entries = {}
        elif isinstance(other, dict):
            self.entries = other
        elif isinstance(other, Config):
            self.entries = json.loads(json.dumps(other.entries))
        else:
            raise ValueError(f'Unknown argument: {other}')

    # -- path helpers ---------------------------------------------------------------------------------
    def _descend(self, key, create):
        """Returns (config holding the leaf, leaf key).  Intermediate levels are created if ``create``."""
        *parents, leaf = key.split('/')
        node = self
        for part in parents:
            node = node.get(part, {}) if create else node[part]
        return node, leaf

    def _wrap(self, value):
        return Config(value) if isinstance(value, dict) else value

    # -- access ---------------------------------------------------------------------------------------
    def get(self, key, default):
        if '/' in key:
            node, leaf = self._descend(key, create=True)
            return node.get(leaf, default)
        if key not in self.entries:
            self.entries[key] = _plain(default)
        return self._wrap(self.entries[key])

    def __getitem__(self, key):
        if '/' in key:
            node, leaf = self._descend(key, create=False)
            return node[leaf]
        return self._wrap(self.entries[key])

    def __contains__(self, key):
        try:
            self[key]
        except KeyError:
            return False
        return True

    def pop(self, key, default):
        if '/' in key:
            node, leaf = self._descend(key, create=True)
            return node.pop(leaf, default)
        return self.entries.pop(key, default)

    def set_default(self, key, default, override_none=False):
        if '/' in key:
            *parents, leaf = key.split('/')
            node = self
            for part in parents:
                node = node.set_default(part, {}, override_none)
            return node.set_default(leaf, default, override_none)
        if key not in self.entries or (override_none and self.entries[key] is None):
            self.entries[key] = _plain(default)
        return self[key]

    def update(self, key, func):
        if '/' in key:
            node, leaf = self._descend(key, create=True)
            return node.update(leaf, func)
        self.entries[key] = _plain(func(self.entries.get(key, None)))
        return self.entries[key]

    def __setitem__(self, key, value):
        self.update(key, lambda *_: value)

    # -- whole-config operations ----------------------------------------------------------------------
    def merge(self, config_override):
        for key, value in _plain(config_override).items():
            if isinstance(value, dict):
                self.get(key, {}).merge(value)
            else:
                self.entries[key] = value
        return self

    def copy(self):
        return Config(self)

    def derive(self, config_override):
        return self.copy().merge(config_override)

    def dump_json(self, fp):
        json.dump(self.entries, fp)

    @property
    def md5(self):
        return hashlib.md5(json.dumps(self.entries).encode('utf8'))

    def __str__(self):
        return json.dumps(self.entries, indent=2)
