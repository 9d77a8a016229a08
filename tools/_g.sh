set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_zz_refrun_gpu.py -q -x -k "field or refrun or replays" 2>&1 | tail -12 | tee gpurun_out/r4l_tests.log
bash tools/probe_two_rank_repro.sh 4 2>&1 | tee gpurun_out/r4l_two_rank.txt
bash tools/gpu.sh ab lego
bash tools/gpu.sh ab fox
cd jnerf_amd/csrc && touch field_split.hip field_mlp.hip && EXTRA=-DNGP_FIELD_STAGE16 bash build.sh > /dev/null 2>&1; cd ../..
echo "--- rebuilt with -DNGP_FIELD_STAGE16 (2-byte staging stores)"
bash tools/gpu.sh ab lego
bash tools/gpu.sh ab fox
