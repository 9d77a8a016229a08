#!/usr/bin/env python
"""Command line of the reference's tools/run_net.py (tools/run_net.py:14-72) on the MI355X path:
    python tools/run_net.py --config-file projects/ngp/configs/ngp_fox.py --task {train,test,render}
    python tools/run_net.py --config-file projects/neus/configs/neus_womask.py --type mesh --task {train,validate_mesh}      (NeuS: jnerf_amd/neus_runner.py)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    parser = argparse.ArgumentParser(description="JNeRF-compatible runner on libngp_hip (MI355X)")
    parser.add_argument("--config-file", default="", metavar="FILE", help="path to config file", type=str)
    parser.add_argument("--task", default="train", help="train,test,render", type=str)
    parser.add_argument("--save_dir", default="", type=str)
    parser.add_argument("--type", default="novel_view", type=str)
    parser.add_argument("--mcube_threshold", default=0.0, type=float)
    args = parser.parse_args()
    assert args.type in ["novel_view", "mesh"], f"{args.type} not support, please choose [novel_view, mesh]"
    assert args.task in ["train", "test", "render", "validate_mesh"], f"{args.task} not support, please choose [train, test, render, validate_mesh]"
    from jnerf_amd.utils.config import init_cfg
    if args.config_file:
        init_cfg(args.config_file)
    if args.type == "mesh":
        from jnerf_amd.neus_runner import NeuSRunner
        runner = NeuSRunner(is_continue=args.task == "validate_mesh")
    else:
        from jnerf_amd.runner import Runner
        runner = Runner()
    if args.task == "train":
        runner.train()
    elif args.task == "test":
        runner.test(True)
    elif args.task == "render":
        runner.render(True, args.save_dir)                   # demo.mp4 with cv2, a PNG sequence next to it without (jnerf_amd/runner.py)
    elif args.task == "validate_mesh":
        runner.validate_mesh(world_space=False, resolution=512, threshold=args.mcube_threshold)

if __name__ == "__main__":
    main()
