"""Minimal ConfigDict (ml_collections is not installed on this platform): nested attribute access, item access,
unknown field -> AttributeError (what the reference's trainers rely on), update from dicts."""


class ConfigDict:
    def __init__(self, initial=None):
        object.__setattr__(self, "_fields", {})
        for k, v in (initial or {}).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        return ConfigDict(v) if isinstance(v, dict) else v

    def __getattr__(self, name):
        try:
            return object.__getattribute__(self, "_fields")[name]
        except KeyError:
            raise AttributeError(f"config has no field '{name}'") from None

    def __setattr__(self, name, value):
        self._fields[name] = self._wrap(value)

    __getitem__ = __getattr__
    __setitem__ = __setattr__

    def __contains__(self, name):
        return name in self._fields

    def get(self, name, default=None):
        return self._fields.get(name, default)

    def keys(self):
        return self._fields.keys()

    def items(self):
        return self._fields.items()

    # dict-valued fields that are assigned as a whole (`config.reward_fn = {...}` upstream), never merged
    ATOMIC = ("reward_fn", "eval_reward_fn", "prompt_fn_kwargs")

    def update(self, other):
        for k, v in (other.items() if hasattr(other, "items") else other):
            if k not in self.ATOMIC and isinstance(v, (dict, ConfigDict)) and isinstance(self._fields.get(k), ConfigDict):
                self._fields[k].update(v)
            else:
                self[k] = v
        return self

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self._fields.items()}

    def __repr__(self):
        return f"ConfigDict({self.to_dict()!r})"
