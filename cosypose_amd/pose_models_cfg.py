"""Model factories, same surface as the reference's cosypose/training/pose_models_cfg.py:13-53."""
from .efficientnet import EfficientNet
from .pose import PosePredictor


def check_update_config(config):
    if not hasattr(config, 'init_method'):
        config.init_method = 'v0'
    return config


def create_model_pose(cfg, renderer, mesh_db):
    n_inputs = 6
    backbone_str = cfg.backbone_str
    if backbone_str == 'efficientnet-b3':
        backbone = EfficientNet.from_name('efficientnet-b3', in_channels=n_inputs)
        backbone.n_features = 1536
    elif backbone_str == 'flownet' or 'resnet34' in backbone_str or 'resnet18' in backbone_str:
        raise ValueError('Backbone not available in the MI355X build (ablation-only in the reference)', backbone_str)
    else:
        raise ValueError('Unknown backbone', backbone_str)
    backbone.n_inputs = n_inputs
    render_size = (240, 320)
    return PosePredictor(backbone=backbone, renderer=renderer, mesh_db=mesh_db,
                         render_size=render_size, pose_dim=cfg.n_pose_dims)


def create_model_refiner(cfg, renderer, mesh_db):
    return create_model_pose(cfg, renderer, mesh_db)


def create_model_coarse(cfg, renderer, mesh_db):
    return create_model_pose(cfg, renderer, mesh_db)
