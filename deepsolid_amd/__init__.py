"""deepsolid_amd: MI355X-native VMC inner loop behind DeepSolid's Python API.

Host orchestration is Python (walkers are torch-ROCm tensors); every numeric
hot-path call goes through the C-ABI HIP library ``libdeepsolid_hip.so``
(``include/deepsolid_hip.h``).  There is no CPU fallback: calling a kernel
entry point without the library / a GPU raises.
"""
__version__ = "0.1.0"
