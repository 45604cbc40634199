"""MI355X-native STLT forward hot path (drop-in for the reference's ``StltBackbone`` / ``Stlt``).

The directory name carries a hyphen, so import it with
``importlib.import_module("revisiting-spatial-temporal-layouts_amd")``.
"""
from . import _lib, collate, dist, infer, ops, synth, train  # noqa: F401
from ._lib import StltHipError  # noqa: F401
from .modelling.configs import MultimodalModelConfig, StltModelConfig, model_configs_factory  # noqa: F401
from .modelling.fusion import CrossAttentionCentralNetFusion, CrossAttentionFusion, LateConcatenationFusion  # noqa: F401
from .modelling.models import (  # noqa: F401
    CategoryBoxEmbeddings,
    ClassificationHead,
    FramesEmbeddings,
    SpatialTransformer,
    Stlt,
    StltBackbone,
    models_factory,
)
from .utils.evaluation import evaluators_factory  # noqa: F401
from .utils.model_utils import generate_square_subsequent_mask  # noqa: F401

__all__ = ["Stlt", "StltBackbone", "StltModelConfig", "models_factory", "model_configs_factory", "StltHipError",
           "ops", "synth", "dist", "infer", "train"]
