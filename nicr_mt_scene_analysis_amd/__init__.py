"""MI355X-native dense-prediction hot path of nicr-mt-scene-analysis.

Drop-in for the reference's `model.postprocessing`, `utils.panoptic_merge`,
`loss`, `metric` and `task_helper` surface on this path; the arithmetic runs in
hand-written HIP kernels for gfx950 behind the C-ABI declared in
`include/nmsa.h` (library: `csrc/libnmsa_hip.so`).
"""
__version__ = '0.1.0'
