"""Tiny attribute-dict (the reference uses easydict / fvcore CfgNode; neither is a dependency here)."""
import ast
import copy

import yaml


class AttrDict(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge(self, other):
        """Deep-merge `other` (yacs semantics: "(3, 7, 7)" strings and tuples become lists)."""
        for k, v in other.items():
            if isinstance(v, dict):
                if not isinstance(self.get(k), AttrDict):
                    self[k] = AttrDict()
                self[k].merge(v)
            else:
                if isinstance(v, str):
                    try:
                        lit = ast.literal_eval(v)
                        if isinstance(lit, (tuple, list)):
                            v = list(lit)
                    except (ValueError, SyntaxError):
                        pass
                elif isinstance(v, tuple):
                    v = list(v)
                self[k] = v
        return self

    def merge_from_file(self, path):
        with open(path) as f:
            return self.merge(yaml.safe_load(f) or {})
