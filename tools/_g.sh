set -u
mkdir -p gpurun_out
python tools/probe_scatter_lego.py 300 spheres > gpurun_out/r4a_probe_scatter.txt 2>&1
tail -25 gpurun_out/r4a_probe_scatter.txt
timeout 1200 python -m pytest tests/test_hip_parity.py -q -x -k "hash_bwd or field32_split" 2>&1 | tail -15 | tee gpurun_out/r4a_tests1.log
timeout 900 python -m pytest tests/test_zz_refrun_gpu.py tests/test_zz_mesh_gpu.py -q -x -s 2>&1 | tail -40 | tee gpurun_out/r4a_tests2.log
bash tools/gpu.sh ab lego "NGP_HASH_BWD_PAIRS=0"
