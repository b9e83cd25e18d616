mkdir -p gpurun_out/r02d
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r02d/pytest.log 2>&1; tail -3 gpurun_out/r02d/pytest.log
python __graft_entry__.py smoke 2>&1 | grep smoke | tail -14
bash tools/profile_round.sh r02d bf16x3 > gpurun_out/r02d/profile.log 2>&1
python - <<'PY'
import json
for t in ("bench","bench_serial","bench_serial_under_rocprof"):
    d=json.load(open('gpurun_out/r02d/%s.json'%t)); print(t, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['resnet_pass_ms'], d['roofline']['avg_launch_us'], d.get('pcie_inclusive_fps'), d.get('pcie_inclusive_fps_1024_frame_video'), d.get('bf16_fps'), d.get('fp32_fps'), d.get('e2e_verts_max_abs_err'))
PY
cat gpurun_out/r02d/pmc_summary.log
bash tools/profile_round.sh r02d_bf16 bf16 > gpurun_out/r02d/profile_bf16.log 2>&1
python - <<'PY'
import json
for t in ("bench","bench_serial"):
    d=json.load(open('gpurun_out/r02d_bf16/%s.json'%t)); print("bf16", t, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['resnet_pass_ms'])
PY
cat gpurun_out/r02d_bf16/pmc_summary.log
