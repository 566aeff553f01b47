O=$PWD/gpurun_out/c12; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "attention or g3_full or g4 or full_size_properties" 2>&1 | tail -3 | tee $O/t.log
TAG=c12 AB='s16:' bash tools/gpu_round.sh ab
