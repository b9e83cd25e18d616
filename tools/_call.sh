python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "stream or host" 2>&1 | tail -2
python tools/pcie_probe.py bf16x3 2>&1 | grep "N=" | grep -v one-shot
python tools/pcie_probe.py bf16 2>&1 | grep "N=1024" | grep -v one-shot
