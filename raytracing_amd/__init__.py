"""MI355X-native path-tracing pass (drop-in for lxtyin/Raytracing's Renderer).

Sub-modules
  build     compiles libraytracing_amd.so in-tree (hipcc, gfx950)
  _capi     ctypes view of include/raytracing_amd.h   (import fails if the .so is missing)
  scene     host-side scene assembly in the reference's buffer layouts
  device    thin Python objects over the C ABI: Context, Buffer, Texture, PathTracingPass
"""
__all__ = ["build", "scene", "device"]
