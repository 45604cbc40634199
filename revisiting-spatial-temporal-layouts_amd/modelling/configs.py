"""Model configuration for the STLT path — same attribute surface as the reference's
``GeneralModelConfig`` / ``StltModelConfig`` (src/modelling/configs.py:92-111) so train.py / inference.py
constructor calls work unchanged."""


class StltModelConfig:
    _DEFAULTS = (
        ("hidden_size", 768),
        ("hidden_dropout_prob", 0.1),
        ("layer_norm_eps", 1e-12),
        ("num_attention_heads", 12),
        ("num_spatial_layers", 4),
        ("num_temporal_layers", 8),
        ("layout_num_frames", 256),
        ("load_backbone_path", None),
        ("freeze_backbone", False),
    )

    def __init__(self, **kwargs):
        self.num_classes = kwargs.pop("num_classes", None)
        assert self.num_classes, "num_classes must not be None!"
        self.unique_categories = kwargs.pop("unique_categories", None)
        assert self.unique_categories, "unique_categories must not be None!"
        for name, default in self._DEFAULTS:
            setattr(self, name, kwargs.pop(name, default))

    def __repr__(self):
        rows = [("Unique categories", self.unique_categories), ("Number of classes", self.num_classes),
                ("Hidden size", self.hidden_size), ("Hidden dropout probability", self.hidden_dropout_prob),
                ("Layer normalization epsilon", self.layer_norm_eps),
                ("Number of attention heads", self.num_attention_heads),
                ("Number of spatial layers", self.num_spatial_layers),
                ("Number of temporal layers", self.num_temporal_layers),
                ("Max number of layout frames", self.layout_num_frames),
                ("The backbone path is", self.load_backbone_path), ("Freezing the backbone", self.freeze_backbone)]
        return "\n".join(f"- {k}: {v}" for k, v in rows)


class MultimodalModelConfig:
    """Attribute surface of the reference's ``MultimodalModelConfig`` (src/modelling/configs.py:128-175) for CAF / CACNF
    on precomputed appearance features: ``stlt_config`` for the layout branch plus the appearance / fusion sizes.
    ``resnet_model_path`` is accepted and ignored (the R3D-50 trunk does not run here)."""

    def __init__(self, **kwargs):
        self.stlt_config = StltModelConfig(**dict(kwargs))
        self.num_classes = self.stlt_config.num_classes
        self.hidden_size = self.stlt_config.hidden_size
        self.hidden_dropout_prob = self.stlt_config.hidden_dropout_prob
        self.layer_norm_eps = self.stlt_config.layer_norm_eps
        self.num_attention_heads = self.stlt_config.num_attention_heads
        self.appearance_num_frames = kwargs.pop("appearance_num_frames", None)
        assert self.appearance_num_frames, "appearance_num_frames must not be None!"
        self.resnet_model_path = kwargs.pop("resnet_model_path", None)
        self.num_appearance_layers = kwargs.pop("num_appearance_layers", 4)
        self.num_fusion_layers = kwargs.pop("num_fusion_layers", 4)
        self.load_backbone_path = kwargs.pop("load_backbone_path", None)
        self.freeze_backbone = kwargs.pop("freeze_backbone", False)
        self.appearance_config = self
        self.stlt_config.load_backbone_path = None  # the fusion models build a fresh layout branch (models.py:439)


model_configs_factory = {"stlt": StltModelConfig, "caf": MultimodalModelConfig, "cacnf": MultimodalModelConfig}
