#!/usr/bin/env python
"""Command line of the reference's tools/extract_mesh.py (tools/extract_mesh.py:12-40) on the MI355X path:
    python tools/extract_mesh.py --config-file projects/ngp/configs/ngp_fox.py [--resolution 512] [--mcube_smooth True]
loads the run's params.pkl, writes mesh-origin.ply and mesh-color.ply next to it (jnerf_amd/mesh.py)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    parser = argparse.ArgumentParser(description="mesh out of a trained Instant-NGP field (libngp_hip, MI355X)")
    parser.add_argument("--config-file", default="", metavar="FILE", help="path to config file", type=str)
    parser.add_argument("--resolution", type=int, default=512, help="resolution of space division")
    # the reference declares type=bool (any non-empty value switches it on); "false" / "0" / "no" switch it off here
    parser.add_argument("--mcube_smooth", nargs="?", const="true", default="", help="smooth the occupancy before the iso-surface (mcubes.smooth in the reference)")
    args = parser.parse_args()
    print(args)
    from jnerf_amd.utils.config import init_cfg
    from jnerf_amd.runner import Runner
    from jnerf_amd.mesh import extract_mesh
    if args.config_file:
        init_cfg(args.config_file)
    runner = Runner()
    runner.load_ckpt(runner.ckpt_path)
    smooth = args.mcube_smooth.lower() not in ("", "false", "0", "no")
    extract_mesh(runner, resolution=args.resolution, smooth=smooth)


if __name__ == "__main__":
    main()
