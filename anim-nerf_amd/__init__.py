"""anim_nerf_amd — MI355X-native per-ray rendering path of Anim-NeRF.

Python host classes keep the reference's names and signatures (AnimNeRF, VolumeRenderer, NeRF,
Embedding, SMPL/create, gen_rays, batched_inference); all per-ray and per-point work runs in
libanimnerf_hip.so (hand-written HIP for gfx950, C ABI in include/animnerf_hip.h).
"""
from . import _lib, data, mesh, mlp, ops, synthetic                  # noqa: F401
from .anim_nerf import AnimNeRF, batch_transform                      # noqa: F401
from .body_model import SMPL, create                                  # noqa: F401
from .nerf import Embedding, NeRF                                     # noqa: F401
from .rays import gen_ray_directions, gen_rays, get_ray_directions, get_rays   # noqa: F401
from .render import (batched_inference, gather_ray_shards, max_over_ranks, render_prepared,   # noqa: F401
                     shard_range, sigma_grid, sigma_grid_inference, system_forward)
from .training import (BodyModelParams, FlatAdam, GradientReducer, TrainHParams, Trainer, allreduce_gradients,   # noqa: F401
                       compute_loss)
from .volume_rendering import VolumeRenderer                          # noqa: F401

__version__ = "0.2.0"


def __getattr__(name):              # `python -m anim_nerf_amd.drivers` must find the module un-imported: load it lazily
    if name == "drivers":
        import importlib
        return importlib.import_module(".drivers", __name__)
    raise AttributeError(name)
