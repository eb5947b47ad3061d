"""Host-side mirror of the pointcept interfaces the hot path sits behind (registries,
``Point``, ``PT-v3m1``, ``LangPretrainer`` + criteria, trainer/hook API)."""
from .registry import HOOKS, LOSSES, MODELS, MODULES, TRAINERS, Registry, build_model  # noqa: F401
from .structure import Point  # noqa: F401
from . import ptv3  # noqa: F401  (registers PT-v3m1)
from . import lang  # noqa: F401  (registers LangPretrainer and the criteria)
from .ptv3 import PointTransformerV3, RUNTIME, bench_runtime  # noqa: F401
from .lang import LangPretrainer, build_criteria  # noqa: F401
from . import engine  # noqa: F401  (registers DefaultTrainer and the hooks)
from .engine import HookBase, Trainer, TrainerBase, create_ddp_model  # noqa: F401
