set -u
mkdir -p gpurun_out
bash tools/round_end.sh tests
bash tools/gpu.sh bench r04_bench_lego
bash tools/gpu.sh bench r04_bench_fox --config fox --no-fox --no-neus
PARTS=neus bash tools/round_end.sh parts
bash tools/round_end.sh profiles
bash tools/round_end.sh curve
