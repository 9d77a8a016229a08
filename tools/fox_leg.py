"""bench.py's real-fox leg on its own (projects/ngp/configs/ngp_fox.py on data/fox: fp16 fused MLP, aabb_scale 4, cone stepping): iterations/s in steady state under the
environment's switches - for A/B pairs inside ONE gpurun call (boxes differ by +-3 %).   usage: [VAR=..] python tools/fox_leg.py [timed steps] [psnr 0|1]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

timed = int(sys.argv[1]) if len(sys.argv) > 1 else 200
psnr = len(sys.argv) > 2 and sys.argv[2] == "1"
out = bench.fox_leg(burn_in=1024, timed=timed, total=3000 if psnr else 1024 + timed, psnr=psnr)
env = {k: v for k, v in os.environ.items() if k.startswith("NGP_")}
print("fox_leg", json.dumps(env), json.dumps(out), flush=True)
