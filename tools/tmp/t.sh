python -m pytest tests/test_gpu_grad.py -q -x -k energy_gradient 2>&1 | grep -E "Error|^E " | head -20
