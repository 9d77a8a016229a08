set -u
mkdir -p gpurun_out
bash tools/repro_two_process.sh 3 > gpurun_out/r4k_repro_two_process.txt 2>&1; cat gpurun_out/r4k_repro_two_process.txt
python tools/probe_scatter.py 400 spheres fox > gpurun_out/r4k_probe_scatter_fox.txt 2>&1; tail -5 gpurun_out/r4k_probe_scatter_fox.txt
bash tools/gpu.sh tests > /dev/null 2>&1; tail -30 gpurun_out/tests_gpu.log
bash tools/gpu.sh ab fox "NGP_HASH_BWD_PAIRS=0"
