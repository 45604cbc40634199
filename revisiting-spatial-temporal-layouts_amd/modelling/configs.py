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


model_configs_factory = {"stlt": StltModelConfig}
