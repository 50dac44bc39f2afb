"""MI355X-native forward Gaussian-splat rasteriser behind the Python surface of
dcaustin33/intro_to_gaussian_splatting (``splat.GaussianScene``): PyTorch-ROCm tensors hold the
Gaussians, hand-written HIP kernels (libgsx.so, C ABI in include/gsx.h) do the work."""
from .gaussians import Gaussians
from .gaussian_scene import GaussianScene, NativeExtension, render_preprocessed
from .image import GaussianImage
from .schema import PreprocessedScene

__all__ = ["Gaussians", "GaussianScene", "GaussianImage", "PreprocessedScene", "render_preprocessed",
           "NativeExtension"]
