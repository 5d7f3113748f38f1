"""``config_parser`` -- the flag surface of TriPlane/opt.py:3-120 and InfoInv/opt.py (same names, types
and defaults), without the ``configargparse`` dependency (absent in this image).

Config files use configargparse's simple syntax, as in TriPlane/configs/lego.txt: ``key = value`` lines,
``#`` comments (also trailing), ``[a, b]`` lists for ``action="append"`` flags.  Command-line values
override the file.  Only the flags marked HOT reach the render path.
"""
from __future__ import annotations

import argparse

# (name, type, default, help) -- scalar options
_SCALARS = [
    ("expname", str, None, "experiment name"),
    ("basedir", str, "./log", "where to store ckpts and logs"),
    ("add_timestamp", int, 0, "add timestamp to dir"),
    ("datadir", str, "./data/llff/fern", "input data directory"),
    ("progress_refresh_rate", int, 10, "iterations between progress lines"),
    ("downsample_train", float, 1.0, "HOT: image size"),
    ("downsample_test", float, 1.0, "(unused by the reference)"),
    ("model_name", str, "TensorVMSplit", "HOT: class name resolved by the drivers (configs say TriPlane)"),
    ("batch_size", int, 4096, "HOT: rays per training step"),
    ("n_iters", int, 30000, ""),
    ("lr_init", float, 0.02, "learning rate of the planes"),
    ("lr_basis", float, 1e-3, "learning rate of the networks"),
    ("lr_decay_iters", int, -1, "-1 = n_iters"),
    ("lr_decay_target_ratio", float, 0.1, ""),
    ("lr_upsample_reset", int, 1, ""),
    ("L1_weight_initial", float, 0.0, ""),
    ("L1_weight_rest", float, 0, ""),
    ("Ortho_weight", float, 0.0, ""),
    ("TV_weight_density", float, 0.0, ""),
    ("TV_weight_app", float, 0.0, ""),
    ("rm_weight_mask_thre", float, 0.0001, "(parsed, never used by the reference: the model keeps 1e-4)"),
    ("alpha_mask_thre", float, 0.0001, "HOT: alpha-mask threshold"),
    ("distance_scale", float, 25, "HOT: scale of the sample distance in raw2alpha"),
    ("density_shift", float, -10, "(parsed, never used: -10 is hard-coded in feature2density)"),
    ("ckpt", str, None, "HOT: checkpoint to load"),
    ("render_only", int, 0, ""),
    ("render_test", int, 0, ""),
    ("render_train", int, 0, ""),
    ("render_path", int, 0, ""),
    ("export_mesh", int, 0, ""),
    ("perturb", float, 1.0, "(unused)"),
    ("accumulate_decay", float, 0.998, "(unused)"),
    ("ndc_ray", int, 0, "(unused)"),
    ("nSamples", int, 1e6, "HOT: cap on samples per ray (min with the automatic count)"),
    ("step_ratio", float, 0.5, "HOT: step = mean voxel size * step_ratio"),
    ("N_voxel_init", int, 100 ** 3, ""),
    ("N_voxel_final", int, 300 ** 3, ""),
    ("idx_view", int, 0, ""),
    ("N_vis", int, 5, ""),
    ("vis_every", int, 10000, ""),
    ("transform_type", str, "continuous", "(unused)"),
]
_FLAGS = ["with_depth", "lindisp", "white_bkgd"]
_APPEND_INT = ["upsamp_list", "update_AlphaMask_list"]
_DATASETS = ["blender", "llff", "nsvf", "dtu", "tankstemple", "own_data"]


def _build(infoinv: bool) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--config", type=str, default=None, help="config file path")
    for name, typ, default, hlp in _SCALARS:
        p.add_argument("--" + name, type=typ, default=default, help=hlp)
    p.add_argument("--dataset_name", type=str, default="blender", choices=_DATASETS)
    for name in _FLAGS:
        p.add_argument("--" + name, action="store_true", default=False)
    for name in _APPEND_INT:
        p.add_argument("--" + name, type=int, action="append")
    if infoinv:
        p.add_argument("--infoinv", action="store_true", default=False, help="HOT: sinusoidal feature modulation")
    else:
        p.add_argument("--gauge_start", type=int, default=0, help="HOT: gauge is applied from this iteration on")
    return p


def _config_file_args(path: str, parser: argparse.ArgumentParser) -> list:
    known = {a.dest: a for a in parser._actions}
    out = []
    with open(path) as fh:
        for raw in fh:
            line = raw.split("#", 1)[0].strip()
            if not line:
                continue
            if "=" in line:
                key, val = (s.strip() for s in line.split("=", 1))
            else:
                key, val = line, "true"
            key = key.lstrip("-")
            if key not in known:
                raise SystemExit(f"{path}: unknown option '{key}'")
            act = known[key]
            if isinstance(act, argparse._StoreTrueAction):
                if val.lower() in ("true", "1", "yes"):
                    out.append("--" + key)
            elif val.startswith("[") and val.endswith("]"):
                for item in val[1:-1].split(","):
                    if item.strip():
                        out += ["--" + key, item.strip()]
            else:
                out += ["--" + key, val]
    return out


def config_parser(cmd=None, infoinv: bool = False):
    """Parse ``cmd`` (list of argv strings) or sys.argv; ``--config file`` is expanded first."""
    import sys
    argv = list(sys.argv[1:] if cmd is None else cmd)
    parser = _build(infoinv)
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--config", type=str, default=None)
    cfg, _ = pre.parse_known_args(argv)
    if cfg.config:
        argv = _config_file_args(cfg.config, parser) + argv
    return parser.parse_args(argv)
