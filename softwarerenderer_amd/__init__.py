"""softwarerenderer_amd -- MI355X-native (gfx950) backend for the raster hot path of
OCSYT/SoftwareRenderer (Rasterizer.cs / Shaders.cs / Texture.cs / MainWindow buffers).

Layout:
  csrc/           hand-written HIP kernels + the C ABI of include/swr.h  -> libswr_hip.so
  _native.py      ctypes binding of that ABI (no CPU fallback: raises if the library is missing)
  rasterizer.py   host-side mirror of the reference API (Rasterizer, Shaders, Texture, MainWindow)
  scenes.py       synthetic scenes of BASELINE.json's configurations
  multigpu.py     tile-row bands across ranks + colour gather (torch.distributed / RCCL)
  hostmath.py     Matrix4x4 factories used to build inputs

Importing the package does not touch the GPU; creating a `Device` does.
"""
from .rasterizer import (BlendMode, CullMode, DebugMode, DepthTest, Device, FrustumCuller, MainWindow, Mesh, Program,  # noqa: F401
                         Rasterizer, ShaderProgram, Shaders, Texture, VERTEX_DTYPE, default_uniforms)

__all__ = ["BlendMode", "CullMode", "DebugMode", "DepthTest", "Device", "FrustumCuller", "MainWindow", "Mesh", "Program",
           "Rasterizer", "ShaderProgram", "Shaders", "Texture", "VERTEX_DTYPE", "default_uniforms"]
