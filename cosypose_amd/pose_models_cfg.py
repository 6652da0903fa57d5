"""Model factories with the interface of the reference's cosypose/training/pose_models_cfg.py:13-53:
check_update_config(cfg), create_model_pose / create_model_refiner / create_model_coarse(cfg, renderer, mesh_db).

Only the backbone every released CosyPose model uses is built for the MI355X path; the reference's ablation
backbones are recognised (so that a config naming them fails with a clear message) but not provided.
"""
from .efficientnet import EfficientNet
from .pose import PosePredictor

# network input = observed crop (3 channels) + rendered crop (3 channels); crop size the reference trains and runs with
N_INPUT_CHANNELS = 6
RENDER_SIZE = (240, 320)

# backbone name -> (constructor, feature width); None = known to the reference, absent here
_BACKBONES = {
    'efficientnet-b3': (lambda: EfficientNet.from_name('efficientnet-b3', in_channels=N_INPUT_CHANNELS), 1536),
    'flownet': None,
}


def _lookup_backbone(name):
    if name in _BACKBONES:
        return _BACKBONES[name]
    if 'resnet34' in name or 'resnet18' in name:       # the reference matches these by substring
        return None
    raise ValueError('Unknown backbone', name)


def check_update_config(config):
    """Fill in fields that older configs lack (the reference's only such field: init_method)."""
    defaults = {'init_method': 'v0'}
    for key, value in defaults.items():
        if not hasattr(config, key):
            setattr(config, key, value)
    return config


def create_model_pose(cfg, renderer, mesh_db):
    entry = _lookup_backbone(cfg.backbone_str)
    if entry is None:
        raise ValueError('Backbone not available in the MI355X build (ablation-only in the reference)', cfg.backbone_str)
    make, n_features = entry
    backbone = make()
    backbone.n_features = n_features
    backbone.n_inputs = N_INPUT_CHANNELS
    return PosePredictor(backbone=backbone, renderer=renderer, mesh_db=mesh_db, render_size=RENDER_SIZE, pose_dim=cfg.n_pose_dims)


# the reference builds coarse and refiner networks with the same factory
create_model_refiner = create_model_pose
create_model_coarse = create_model_pose
