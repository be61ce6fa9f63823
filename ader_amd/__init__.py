"""ader_amd: MI355X-native (gfx950) implementation of the ADER / SASRec training hot path.

Package layout (only what the hot path needs):
  csrc/      hand-written HIP kernels + the C-ABI (libader_hip.so, declared in include/ader_hip.h)
  _lib.py    ctypes binding of the C-ABI (fails loudly when the HIP library is missing)
  engine.py  device state (parameters, Adam slots, workspaces) and step orchestration
  model.py   `Ader` with the reference's call surface (ADER.py:13-150) + Session/Saver shims
  data.py    DataLoader / Sampler / Evaluator counterparts (reference util.py)
  exemplar.py ExemplarGenerator counterpart (herding on the GPU)
  dist.py    one-process-per-GPU data parallelism over RCCL
"""
__version__ = "0.1.0"
