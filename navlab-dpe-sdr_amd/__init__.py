"""navlab-dpe-sdr_amd -- MI355X-native DPE correlator engine (sampleblock -> BCS -> BCM hot path).

Python side = thin ctypes host layer over the C-ABI in csrc/ (include/dpe_hip.h), plus
synthetic-input and file-format utilities.  Imported as `navlab_dpe_sdr_amd` through the
root-level shim navlab_dpe_sdr_amd.py (the directory name carries a hyphen).
"""
from . import engine, handoff, pipeline, rinex, sharding, synth, workload  # noqa: F401
from .engine import Acquisition, BatchCorrManifold, BatchCorrScores, ChanMgr, DpeError, Pipe, cuEKF  # noqa: F401

__all__ = ["engine", "handoff", "pipeline", "rinex", "sharding", "synth", "workload", "BatchCorrScores", "BatchCorrManifold", "ChanMgr", "Acquisition", "DpeError", "cuEKF"]
