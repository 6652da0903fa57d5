"""cosypose_amd: MI355X-native implementation of CosyPose's render-and-compare pose-refinement
hot path (coarse + refiner loop) behind the reference's Python API.  See DESIGN.md."""
from .tensor_collection import TensorCollection, PandasTensorCollection, concatenate  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require torch.cuda or the built library
    if name in ('PosePredictor',):
        from .pose import PosePredictor
        return PosePredictor
    if name in ('CoarseRefinePosePredictor',):
        from .pose_predictor import CoarseRefinePosePredictor
        return CoarseRefinePosePredictor
    if name in ('create_model_pose', 'create_model_refiner', 'create_model_coarse', 'check_update_config'):
        from . import pose_models_cfg
        return getattr(pose_models_cfg, name)
    if name == 'BatchedMeshes':
        from .mesh_db import BatchedMeshes
        return BatchedMeshes
    if name in ('HipBatchRenderer', 'RenderMeshes'):
        from . import rasterizer
        return getattr(rasterizer, name)
    if name == 'h_pose':
        from .pose_forward_loss import h_pose
        return h_pose
    raise AttributeError(name)
