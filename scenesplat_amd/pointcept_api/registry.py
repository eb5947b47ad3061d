"""Registry with the reference's build contract: ``REG.build(dict(type=name, **kwargs))``.

Mirrors pointcept/utils/registry.py:9-56,59-316 (mmcv-style) as far as the hot path uses it:
``register_module(name=None)`` decorator / call form, ``get``, ``build`` with the class-name
prefixed re-raise of constructor errors (registry.py:52-56)."""
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if "type" not in cfg:
        if default_args is None or "type" not in default_args:
            raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)(f"{obj_cls.__name__}: {e}")


class Registry:
    def __init__(self, name, build_func=None):
        self._name = name
        self._module_dict = {}
        self.build_func = build_func or build_from_cfg

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f"{self.__class__.__name__}(name={self._name}, items={list(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register_module(self, module_class, module_name=None, force=False):
        if not inspect.isclass(module_class):
            raise TypeError(f"module must be a class, but got {type(module_class)}")
        names = [module_name] if isinstance(module_name, str) else (module_name or [module_class.__name__])
        for name in names:
            if not force and name in self._module_dict:
                raise KeyError(f"{name} is already registered in {self.name}")
            self._module_dict[name] = module_class

    def register_module(self, name=None, force=False, module=None):
        if isinstance(name, type):  # @REG.register_module without call
            self._register_module(name)
            return name
        if module is not None:
            self._register_module(module, name, force)
            return module

        def _register(cls):
            self._register_module(cls, name, force)
            return cls

        return _register


MODELS = Registry("models")
MODULES = Registry("modules")
LOSSES = Registry("losses")
HOOKS = Registry("hooks")
TRAINERS = Registry("trainers")


def build_model(cfg):
    """pointcept/models/builder.py:14-16"""
    return MODELS.build(cfg)
