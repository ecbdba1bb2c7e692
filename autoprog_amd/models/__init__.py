from .registry import create_model, register_model, list_models, is_model  # noqa: F401
from .volo import *  # noqa: F401,F403
from . import volo, submodels, deit  # noqa: F401
from .submodels import model_variant  # noqa: F401
