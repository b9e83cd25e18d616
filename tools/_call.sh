mkdir -p gpurun_out/r02b
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r02b/pytest.log 2>&1; tail -5 gpurun_out/r02b/pytest.log
bash tools/profile_round.sh r02b bf16x3 > gpurun_out/r02b/profile.log 2>&1
tail -75 gpurun_out/r02b/profile.log
