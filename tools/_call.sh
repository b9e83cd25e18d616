python tools/pcie_probe.py bf16x3 2>&1 | grep -v amdgpu.ids | grep "N="
python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "stream or host" 2>&1 | tail -3
