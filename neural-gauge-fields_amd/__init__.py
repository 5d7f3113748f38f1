"""MI355X-native TriPlane / InfoInv ray-march renderer (drop-in for the reference's
``field(rays)`` / ``renderer(rays, field, ...)`` boundary).  See DESIGN.md.

Sub-modules: ``synth`` (seeded inputs), ``geometry`` (init_para scalars), ``field`` (TriPlane /
InfoInv modules + renderer over the HIP C-ABI), ``opt`` (config_parser), ``dist`` (ray-sharded
multi-GPU render), ``_lib`` (ctypes binding of libngf_hip.so).
"""
__all__ = ["synth", "geometry", "field", "opt", "dist"]
