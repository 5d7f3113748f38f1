"""MI355X-native TriPlane / InfoInv ray-march renderer (drop-in for the reference's
``field(rays)`` / ``renderer(rays, field, ...)`` boundary).  See DESIGN.md.

Sub-modules: ``triplane`` / ``infoinv`` / ``uvmapping`` (drop-in field modules over the HIP C-ABI), ``fieldbase``
(shared Base + renderer), ``geometry`` (init_para scalars), ``opt`` (config_parser), ``rays`` (on-device ray
generation), ``dist`` (ray-sharded multi-GPU render), ``synth`` (seeded inputs), ``_lib`` (ctypes binding of
libngf_hip.so).
"""
__all__ = ["triplane", "infoinv", "uvmapping", "fieldbase", "geometry", "opt", "rays", "dist", "synth", "cases"]
