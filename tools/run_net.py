#!/usr/bin/env python
"""Command line of the reference's tools/run_net.py (tools/run_net.py:14-72) on the MI355X path:
    python tools/run_net.py --config-file projects/ngp/configs/ngp_fox.py --task {train,test,render}
--type mesh / --task validate_mesh (NeuS) are out of scope."""
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
    if args.type == "mesh" or args.task == "validate_mesh":
        raise SystemExit("NeuS / mesh extraction is outside the Instant-NGP hot path this build covers (DESIGN.md §8)")
    from jnerf_amd.utils.config import init_cfg
    from jnerf_amd.runner import Runner
    if args.config_file:
        init_cfg(args.config_file)
    runner = Runner()
    if args.task == "train":
        runner.train()
    elif args.task == "test":
        runner.test(True)
    elif args.task == "render":
        runner.render(True, args.save_dir)                   # demo.mp4 with cv2, a PNG sequence next to it without (jnerf_amd/runner.py)

if __name__ == "__main__":
    main()
