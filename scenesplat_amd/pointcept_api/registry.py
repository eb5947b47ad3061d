"""Name -> class tables with the reference's build contract: ``REG.build(dict(type=name, **kwargs))``.

Boundary b1 (pointcept/utils/registry.py:9-56,212 for the contract, not the code): configs name a class by
``type`` (a registered string or a class object), remaining keys are constructor kwargs, ``default_args`` fill
keys the config leaves out, and a constructor failure is re-raised as the same exception type with the class name
in front.  Classes enter a table through ``@REG.register_module()``, ``@REG.register_module("alias")`` or
``REG.register_module(name=..., module=cls)``.
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self._table = {}

    # ---- lookup ---------------------------------------------------------------------------------------------
    @property
    def module_dict(self):
        return self._table

    def get(self, key):
        return self._table.get(key)

    def __contains__(self, key):
        return key in self._table

    def __len__(self):
        return len(self._table)

    def __repr__(self):
        return "Registry(%s: %s)" % (self.name, ", ".join(sorted(self._table)))

    # ---- registration ---------------------------------------------------------------------------------------
    def _add(self, cls, names, force):
        if not isinstance(cls, type):
            raise TypeError("only classes can be registered in %r, got %r" % (self.name, cls))
        for nm in names:
            if nm in self._table and not force:
                raise KeyError("%r already names a class in the %s registry" % (nm, self.name))
            self._table[nm] = cls
        return cls

    def register_module(self, name=None, force=False, module=None):
        if isinstance(name, type):                          # bare @REG.register_module
            return self._add(name, [name.__name__], force)
        aliases = None if name is None else ([name] if isinstance(name, str) else list(name))
        if module is not None:                              # call form
            return self._add(module, aliases or [module.__name__], force)
        return lambda cls: self._add(cls, aliases or [cls.__name__], force)

    # ---- construction ---------------------------------------------------------------------------------------
    def resolve(self, spec):
        """A ``type`` entry -> class."""
        if isinstance(spec, type):
            return spec
        if isinstance(spec, str):
            try:
                return self._table[spec]
            except KeyError:
                raise KeyError("%s is not in the %s registry" % (spec, self.name)) from None
        raise TypeError("'type' must be a registered name or a class, got %r" % type(spec).__name__)

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict):
            raise TypeError("config must be a dict, got %r" % type(cfg).__name__)
        kwargs = {**(default_args or {}), **cfg}            # config keys win over defaults
        if "type" not in kwargs:
            raise KeyError("config (or default_args) needs a 'type' key: %r" % (cfg,))
        cls = self.resolve(kwargs.pop("type"))
        try:
            return cls(**kwargs)
        except Exception as err:                            # plain constructor errors do not say which class failed
            raise type(err)("%s: %s" % (cls.__name__, err)) from err


MODELS = Registry("models")
MODULES = Registry("modules")
LOSSES = Registry("losses")
HOOKS = Registry("hooks")
TRAINERS = Registry("trainers")


def build_model(cfg):
    """pointcept/models/builder.py:14-16"""
    return MODELS.build(cfg)
