mkdir -p gpurun_out/r02c
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r02c/pytest.log 2>&1; tail -4 gpurun_out/r02c/pytest.log
bash tools/profile_round.sh r02c bf16x3 > gpurun_out/r02c/profile.log 2>&1
tail -70 gpurun_out/r02c/profile.log | cut -c1-1500
