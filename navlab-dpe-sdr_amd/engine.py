"""ctypes host layer over the C-ABI (include/dpe_hip.h).

Mirrors the reference's operator interface for the hot path -- BatchCorrScores.Start/Update/
Stop and BatchCorrManifold.Start/Update/Stop (cudarecv/modules/src/batchcorrscores.cu:710-1208,
batchcorrmanifold.cu:2315-2635) -- with the reference's port names as keyword arguments.
There is NO CPU fallback: if libdpe_hip.so is missing or no GPU is present, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPE_LIB_PATH") or os.path.join(_HERE, "libdpe_hip.so")   # DPE_LIB_PATH: A/B builds of the library (experiments)
_lib = None

EXPORTS = [
    "dpe_abi_version", "dpe_last_error", "dpe_device_info", "dpe_gen_ca_code", "dpe_sampleblock_upload",
    "dpe_host_alloc_pinned", "dpe_host_free_pinned", "dpe_device_alloc", "dpe_device_free", "dpe_memcpy_h2d",
    "dpe_memcpy_d2h", "dpe_stream_create", "dpe_stream_destroy", "dpe_stream_synchronize",
    "dpe_bcs_create", "dpe_bcs_destroy", "dpe_bcs_update", "dpe_bcs_outputs", "dpe_bcs_read_info",
    "dpe_bcs_export_dense", "dpe_bcm_create", "dpe_bcm_destroy", "dpe_bcm_update", "dpe_bcm_results",
    "dpe_bcm_scores", "dpe_bcm_scores_pitch", "dpe_bcm_keys", "dpe_bcm_results_from_keys", "dpe_event_create", "dpe_event_record",
    "dpe_event_elapsed_ms", "dpe_event_destroy", "dpe_chm_create", "dpe_chm_destroy", "dpe_chm_start",
    "dpe_chm_update", "dpe_chm_outputs", "dpe_bcs_profile", "dpe_bcm_profile", "dpe_acq_create", "dpe_acq_destroy",
    "dpe_acq_search", "dpe_acq_results", "dpe_acq_surface", "dpe_bcs_set_graph", "dpe_bcm_set_graph",
    "dpe_acq_fine", "dpe_acq_scalar_acquisition",
    "dpe_ekf_create", "dpe_ekf_destroy", "dpe_ekf_step_update", "dpe_ekf_step_predict", "dpe_ekf_state",
    "dpe_hbm_ceiling", "dpe_bcs_stage1_kernel",
    "dpe_set_device", "dpe_comm_create", "dpe_comm_wrap_nccl", "dpe_comm_destroy", "dpe_comm_rank", "dpe_comm_allreduce_max_u64",
    "dpe_comm_allgather", "dpe_bcm_exchange_keys", "dpe_bcs_allgather_banks",
    "dpe_bcs_update_dev", "dpe_bcs_dev_status", "dpe_bcm_update_dev", "dpe_bcm_export_scores_f64",
    "dpe_chm_dev_create", "dpe_chm_dev_destroy", "dpe_chm_dev_attach", "dpe_chm_dev_ports", "dpe_chm_dev_start", "dpe_chm_dev_update",
    "dpe_chm_dev_step", "dpe_chm_dev_fix", "dpe_chm_dev_read", "dpe_bcs_update_prepared", "dpe_bcm_update_prepared", "dpe_bcs_set_dev_hint",
    "dpe_chm_dev_set_shard", "dpe_chm_dev_set_ekf",
    "dpe_pipe_create", "dpe_pipe_destroy", "dpe_pipe_in_flight", "dpe_pipe_submit", "dpe_pipe_acquire", "dpe_pipe_mark_stage1",
    "dpe_pipe_commit", "dpe_pipe_lane", "dpe_pipe_set_in_flight", "dpe_pipe_lane_at", "dpe_pipe_results", "dpe_pipe_samples_consumed", "dpe_pipe_join", "dpe_pipe_synchronize",
]


class DpeError(RuntimeError):
    pass


class BcsConfig(C.Structure):
    _fields_ = [("samplesPerWindow", C.c_int32), ("lagHalfWidth", C.c_int32), ("binHalfWidth", C.c_int32),
                ("maxWindows", C.c_int32), ("maxChannels", C.c_int32), ("reserved", C.c_int32),
                ("samplingFrequency", C.c_double)]


class BcsPortsDev(C.Structure):     # dpe_bcs_ports_dev: device pointers to the reference's port arrays
    _fields_ = [(n, C.c_void_p) for n in ("codePhaseStart", "carrierPhaseStart", "codeFrequency", "carrierFrequency",
                                          "cpElapsedStart", "cpReference", "validPRNs")]


class BcmPortsDev(C.Structure):     # dpe_bcm_ports_dev
    _fields_ = [(n, C.c_void_p) for n in ("xCurrkk1", "enu2ecef", "satStates", "codePhaseEnd", "codeFrequency",
                                          "carrierFrequency", "cpRefTOW", "cpElapsedEnd", "cpRef", "dopplerSign")] + \
               [("dimT", C.c_int32), ("reserved", C.c_int32)]


class ChanStart(C.Structure):
    _fields_ = [("codePhaseStart", C.c_double), ("carrierPhaseStart", C.c_double), ("codeFrequency", C.c_double),
                ("carrierFrequency", C.c_double), ("cpElapsedStart", C.c_int32), ("cpReference", C.c_int32),
                ("prn", C.c_int32), ("reserved", C.c_int32)]


class BcmConfig(C.Structure):
    _fields_ = [("samplesPerWindow", C.c_int32), ("lagHalfWidth", C.c_int32), ("binHalfWidth", C.c_int32),
                ("lPower", C.c_int32), ("maxWindows", C.c_int32), ("maxChannels", C.c_int32),
                ("numFFTPoints", C.c_int64), ("samplingFrequency", C.c_double),
                ("posGrid", C.POINTER(C.c_double)), ("velGrid", C.POINTER(C.c_double)),
                ("posGridSize", C.c_int64), ("velGridSize", C.c_int64),
                ("posGridIndexOffset", C.c_int64), ("velGridIndexOffset", C.c_int64),
                ("writeScores", C.c_int32), ("weightedMean", C.c_int32), ("referencePair", C.c_int32), ("reserved", C.c_int32)]


class BcmWindow(C.Structure):
    _fields_ = [("xCurrkk1", C.c_double * 8), ("enu2ecef", C.c_double * 9), ("rxTime", C.c_double),
                ("dopplerSign", C.c_int32), ("reserved", C.c_int32)]


class ChanEnd(C.Structure):
    _fields_ = [("satState", C.c_double * 8), ("codePhaseEnd", C.c_double), ("codeFrequency", C.c_double),
                ("carrierFrequency", C.c_double), ("cpRefTOW", C.c_int32), ("cpElapsedEnd", C.c_int32),
                ("cpRef", C.c_int32), ("reserved", C.c_int32)]


class BcmResult(C.Structure):
    _fields_ = [("zVal", C.c_double * 8), ("posIndex", C.c_int64), ("velIndex", C.c_int64),
                ("posScore", C.c_float), ("velScore", C.c_float), ("posOutOfWindow", C.c_int64),
                ("velOutOfWindow", C.c_int64), ("zValMean", C.c_double * 8), ("weightedSums", (C.c_double * 5) * 2)]


CHAN_START_DTYPE = np.dtype([("codePhaseStart", "<f8"), ("carrierPhaseStart", "<f8"), ("codeFrequency", "<f8"),
                             ("carrierFrequency", "<f8"), ("cpElapsedStart", "<i4"), ("cpReference", "<i4"),
                             ("prn", "<i4"), ("reserved", "<i4")])
CHAN_END_DTYPE = np.dtype([("satState", "<f8", (8,)), ("codePhaseEnd", "<f8"), ("codeFrequency", "<f8"),
                           ("carrierFrequency", "<f8"), ("cpRefTOW", "<i4"), ("cpElapsedEnd", "<i4"),
                           ("cpRef", "<i4"), ("reserved", "<i4")])
BCM_WINDOW_DTYPE = np.dtype([("xCurrkk1", "<f8", (8,)), ("enu2ecef", "<f8", (9,)), ("rxTime", "<f8"),
                             ("dopplerSign", "<i4"), ("reserved", "<i4")])
assert CHAN_START_DTYPE.itemsize == C.sizeof(ChanStart)
assert CHAN_END_DTYPE.itemsize == C.sizeof(ChanEnd)
assert BCM_WINDOW_DTYPE.itemsize == C.sizeof(BcmWindow)


def lib():
    """Load libdpe_hip.so (built in-tree by __graft_entry__.build()).  No fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DpeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                           % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 with the same SONAME as the system one
        # this library links against.  Whichever is loaded first serves both; loading the system runtime first leaves
        # torch with "No HIP GPUs are available".  So torch (when present) goes first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.dpe_last_error.restype = C.c_char_p
        _lib.dpe_bcs_stage1_kernel.restype = C.c_char_p
        for name in EXPORTS:
            getattr(_lib, name)  # AttributeError if the ABI is incomplete
    return _lib


def _check(rc):
    if rc != 0:
        raise DpeError(lib().dpe_last_error().decode("utf-8", "replace"))


def _ptr(x):
    """torch tensor / int -> raw device pointer."""
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


STREAM_NONE = "none"      # Pipe.submit / acquire: the samples are already resident (DPE_STREAM_NONE: no cross-stream wait)


def _stream(stream):
    if isinstance(stream, str) and stream == STREAM_NONE:
        return C.c_void_p(-1)
    if stream is None:
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


class Stream:
    """A HIP stream owned by the library (dpe_stream_create); pass it as ``stream=`` to the modules."""

    def __init__(self):
        self._s = C.c_void_p(None)
        _check(lib().dpe_stream_create(C.byref(self._s)))
        self.cuda_stream = self._s.value

    def synchronize(self):
        _check(lib().dpe_stream_synchronize(self._s))

    def close(self):
        if self._s:
            lib().dpe_stream_destroy(self._s)
            self._s = C.c_void_p(None)
            self.cuda_stream = 0


def carr_fft_len(S):
    p = 1
    while p < S:
        p <<= 1
    return 8 * p


def gen_ca_code():
    out = np.zeros((37, 1023), dtype=np.int8)
    _check(lib().dpe_gen_ca_code(out.ctypes.data_as(C.POINTER(C.c_int8))))
    return out


def device_info():
    name = C.create_string_buffer(128)
    cu, mem = C.c_int(0), C.c_int64(0)
    _check(lib().dpe_device_info(name, 128, C.byref(cu), C.byref(mem)))
    return name.value.decode(), cu.value, mem.value


def hbm_ceiling(bytes_per_array=1 << 30, iters=10, stream=None):
    """Measured stream-copy and triad bandwidth of this device in GB/s (diagnostic; see dpe_hbm_ceiling)."""
    cp, tr = C.c_double(0), C.c_double(0)
    _check(lib().dpe_hbm_ceiling(C.c_int64(bytes_per_array), C.c_int(iters), _stream(stream), C.byref(cp), C.byref(tr)))
    return cp.value, tr.value


def d2h(ptr, nbytes, dtype, stream=None):
    out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
    _check(lib().dpe_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_int64(nbytes), _stream(stream)))
    return out


def chan_start_array(prn, rc, ri, fc, fi, cp_ela, cp_ref):
    """Pack start-referenced channel params (arrays of shape [W,K] or [K]) for BatchCorrScores."""
    rc = np.asarray(rc, dtype=np.float64)
    a = np.zeros(rc.shape, dtype=CHAN_START_DTYPE)
    a["codePhaseStart"], a["carrierPhaseStart"] = rc, ri
    a["codeFrequency"], a["carrierFrequency"] = fc, fi
    a["cpElapsedStart"], a["cpReference"], a["prn"] = cp_ela, cp_ref, prn
    return a


def chan_end_array(sat, rc_end, fc, fi, cp_ref_tow, cp_ela_end, cp_ref):
    rc_end = np.asarray(rc_end, dtype=np.float64)
    a = np.zeros(rc_end.shape, dtype=CHAN_END_DTYPE)
    a["satState"] = sat
    a["codePhaseEnd"], a["codeFrequency"], a["carrierFrequency"] = rc_end, fc, fi
    a["cpRefTOW"], a["cpElapsedEnd"], a["cpRef"] = cp_ref_tow, cp_ela_end, cp_ref
    return a


def bcm_window_array(x_kk1, enu2ecef, rx_time, doppler_sign=1):
    rx_time = np.atleast_1d(np.asarray(rx_time, dtype=np.float64))
    a = np.zeros(rx_time.shape, dtype=BCM_WINDOW_DTYPE)
    a["xCurrkk1"] = x_kk1
    a["enu2ecef"] = np.asarray(enu2ecef).reshape(rx_time.shape + (9,))
    a["rxTime"] = rx_time
    a["dopplerSign"] = doppler_sign
    return a


class BatchCorrScores:
    """Module "BatchCorrScores" (batchcorrscores.cu:674-700): Start/Update/Stop + output ports."""

    def __init__(self, SamplingFrequency, SampleLength=None, samples_per_window=None, lag_half_width=8,
                 bin_half_width=48, max_windows=1, max_channels=8):
        S = int(samples_per_window) if samples_per_window is not None else int(round(SamplingFrequency * SampleLength))
        self.S, self.fs = S, float(SamplingFrequency)
        self.L, self.B = int(lag_half_width), int(bin_half_width)
        self.max_windows, self.max_channels = int(max_windows), int(max_channels)
        self._h = C.c_void_p(None)
        self.Started = False

    def Start(self):
        if self.Started:
            return 0
        cfg = BcsConfig(self.S, self.L, self.B, self.max_windows, self.max_channels, 0, self.fs)
        _check(lib().dpe_bcs_create(C.byref(cfg), C.byref(self._h)))
        code, carr = C.c_void_p(), C.c_void_p()
        nlag, nbin, nfft = C.c_int32(), C.c_int32(), C.c_int64()
        _check(lib().dpe_bcs_outputs(self._h, C.byref(code), C.byref(carr), C.byref(nlag), C.byref(nbin), C.byref(nfft)))
        self.CodeScores, self.CarrScores = code.value, carr.value     # device pointers (float2 banks)
        self.nLag, self.nBin, self.NumFFTPoints = nlag.value, nbin.value, nfft.value
        self.Started = True
        self._W = self._K = 0
        return 0

    def Update(self, Samples, chan, window_stride=None, stream=None):
        """Samples: device int16 [W, 2*S] (torch tensor or raw pointer); chan: CHAN_START_DTYPE [W,K]."""
        if not self.Started:
            raise DpeError("[BatchCorrScores] Error: Update() Failed due to batch correlator not initialized")
        chan = np.ascontiguousarray(chan)
        if chan.ndim == 1:
            chan = chan[None, :]
        W, K = chan.shape
        stride = self.S if window_stride is None else int(window_stride)
        _check(lib().dpe_bcs_update(self._h, _ptr(Samples), C.c_int64(stride), C.c_int32(W), C.c_int32(K),
                                    chan.ctypes.data_as(C.POINTER(ChanStart)), _stream(stream)))
        self._W, self._K = W, K
        return 0

    def UpdateDev(self, Samples, n_chan, ports, stream=None):
        """One window with the channel parameters in DEVICE arrays (dpe_bcs_update_dev): ports = {field: device pointer}
        with the fields of dpe_bcs_ports_dev."""
        if not self.Started:
            raise DpeError("[BatchCorrScores] Error: Update() Failed due to batch correlator not initialized")
        p = BcsPortsDev(**{k: _ptr(v).value for k, v in ports.items()})
        _check(lib().dpe_bcs_update_dev(self._h, _ptr(Samples), C.c_int32(n_chan), C.byref(p), _stream(stream)))
        self._W, self._K = 1, int(n_chan)
        return 0

    def UpdatePrepared(self, Samples, n_chan, stream=None):
        """One window whose channel block an attached ChanMgrDev has already written on the device."""
        _check(lib().dpe_bcs_update_prepared(self._h, _ptr(Samples), C.c_int32(n_chan), _stream(stream)))
        self._W, self._K = 1, int(n_chan)
        return 0

    def allgather_banks(self, comm, code_all, carr_all, stream=None):
        """Stage 1 sharded by window: this rank's banks of the last Update into code_all / carr_all (device pointers,
        [nRanks * W_local][maxChannels][2L+1 | 2B+1] float2), rank-major (dpe_bcs_allgather_banks)."""
        _check(lib().dpe_bcs_allgather_banks(self._h, comm._h, _ptr(code_all), _ptr(carr_all), _stream(stream)))

    def set_dev_hint(self, flags=1):
        """dpe_bcs_set_dev_hint: flags bit 0 = the chip kernels' conditions hold for every channel (no readback in UpdateDev)."""
        _check(lib().dpe_bcs_set_dev_hint(self._h, C.c_int32(flags)))

    def dev_status(self, stream=None):
        st = C.c_int32()
        _check(lib().dpe_bcs_dev_status(self._h, C.byref(st), _stream(stream)))
        return st.value

    def set_graph(self, enable=True):
        """Replay repeated Updates as one hipGraph launch (needs a created stream, see dpe_hip.h)."""
        _check(lib().dpe_bcs_set_graph(self._h, C.c_int32(1 if enable else 0)))

    PROFILE_SLOTS = ("bcs_sum", "bcs_bank", "bcs_finalize")

    def profile(self, enable=True):
        """-> {kernel: (total_ms, launches)} since the previous call; sets the enable flag.
        enable: False / True (every kernel) / one of PROFILE_SLOTS (events around that kernel only)."""
        ms, cnt = (C.c_float * 3)(), (C.c_int32 * 3)()
        code = 2 << self.PROFILE_SLOTS.index(enable) if isinstance(enable, str) else (1 if enable else 0)
        _check(lib().dpe_bcs_profile(self._h, C.c_int32(code), ms, cnt))
        return {n: (ms[i], cnt[i]) for i, n in enumerate(self.PROFILE_SLOTS)}

    @property
    def stage1_kernel(self):
        return lib().dpe_bcs_stage1_kernel(self._h).decode()

    def Stop(self):
        if self.Started:
            if getattr(self, "_owned", True):      # (a lane of a Pipe is destroyed by the pipe)
                _check(lib().dpe_bcs_destroy(self._h))
            self._h = C.c_void_p(None)
            self.Started = False
        return 0

    @classmethod
    def _adopt(cls, handle, fs, S, L, B, max_windows, max_channels):
        """Python face of a handle that a dpe_pipe owns."""
        self = cls(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=max_windows, max_channels=max_channels)
        self._h, self._owned = C.c_void_p(handle), False
        code, carr = C.c_void_p(), C.c_void_p()
        nlag, nbin, nfft = C.c_int32(), C.c_int32(), C.c_int64()
        _check(lib().dpe_bcs_outputs(self._h, C.byref(code), C.byref(carr), C.byref(nlag), C.byref(nbin), C.byref(nfft)))
        self.CodeScores, self.CarrScores = code.value, carr.value
        self.nLag, self.nBin, self.NumFFTPoints = nlag.value, nbin.value, nfft.value
        self.Started = True
        self._W = self._K = 0
        return self

    def __del__(self):
        try:            # at interpreter shutdown module globals may already be gone
            self.Stop()
        except Exception:
            pass

    # ---- host-side readers (tests / diagnostics)
    def read_banks(self, stream=None):
        W, K = self._W, self._K
        code = d2h(self.CodeScores, self.max_windows * self.max_channels * self.nLag * 8, np.complex64, stream)
        carr = d2h(self.CarrScores, self.max_windows * self.max_channels * self.nBin * 8, np.complex64, stream)
        code = code.reshape(self.max_windows, self.max_channels, self.nLag)[:W, :K]
        carr = carr.reshape(self.max_windows, self.max_channels, self.nBin)[:W, :K]
        return code, carr

    def read_info(self, stream=None):
        n = self._W * self._K
        idx = np.zeros(n, dtype=np.int32)
        nfl = np.zeros(n, dtype=np.int32)
        mean = np.zeros(2 * self._W)
        _check(lib().dpe_bcs_read_info(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                       nfl.ctypes.data_as(C.POINTER(C.c_int32)),
                                       mean.ctypes.data_as(C.POINTER(C.c_double)), _stream(stream)))
        return (idx.reshape(self._W, self._K), nfl.reshape(self._W, self._K).astype(bool),
                mean.reshape(self._W, 2).view(np.complex128).ravel())

    def export_dense(self, window, code_dev, carr_dev, stream=None):
        _check(lib().dpe_bcs_export_dense(self._h, C.c_int32(window), _ptr(code_dev) if code_dev is not None else None,
                                          _ptr(carr_dev) if carr_dev is not None else None, _stream(stream)))


class Comm:
    """dpe_comm: the multi-GPU exchange behind the C-ABI (RCCL, or host files for one-GPU functional tests)."""
    RCCL, HOSTFILES = 0, 1

    def __init__(self, rank, n_ranks, rendezvous="", backend=0):
        self._h = C.c_void_p(None)
        _check(lib().dpe_comm_create(C.c_int32(rank), C.c_int32(n_ranks), rendezvous.encode(), C.c_int32(backend), C.byref(self._h)))
        self.rank, self.n_ranks = rank, n_ranks

    def allreduce_max_u64(self, dev_ptr, count, stream=None):
        _check(lib().dpe_comm_allreduce_max_u64(self._h, C.c_void_p(dev_ptr), C.c_int64(count), _stream(stream)))

    def close(self):
        if self._h:
            lib().dpe_comm_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchCorrManifold:
    """Module "BatchCorrManifold" (batchcorrmanifold.cu:2247-2303): params PosGrid/VelGrid (host
    [G,4] ENU offsets, i.e. the LoadPosGrid path :2422-2448 generalised to both manifolds), LPower."""

    def __init__(self, SamplingFrequency, samples_per_window, NumFFTPoints, pos_grid, vel_grid, LPower=1,
                 lag_half_width=8, bin_half_width=48, max_windows=1, max_channels=8, write_scores=True,
                 pos_index_offset=0, vel_index_offset=0, weighted_mean=False, reference_pair=False):
        self.fs, self.S, self.C = float(SamplingFrequency), int(samples_per_window), int(NumFFTPoints)
        self.pos_grid = np.ascontiguousarray(pos_grid, dtype=np.float64)
        self.vel_grid = np.ascontiguousarray(vel_grid, dtype=np.float64)
        self.LPower, self.L, self.B = int(LPower), int(lag_half_width), int(bin_half_width)
        self.max_windows, self.max_channels = int(max_windows), int(max_channels)
        self.write_scores = bool(write_scores)
        self.weighted_mean = bool(weighted_mean)
        self.reference_pair = bool(reference_pair)
        self.pos_off, self.vel_off = int(pos_index_offset), int(vel_index_offset)
        self._h = C.c_void_p(None)
        self.Started = False

    def Start(self):
        if self.Started:
            return 0
        cfg = BcmConfig(self.S, self.L, self.B, self.LPower, self.max_windows, self.max_channels, self.C, self.fs,
                        self.pos_grid.ctypes.data_as(C.POINTER(C.c_double)),
                        self.vel_grid.ctypes.data_as(C.POINTER(C.c_double)),
                        self.pos_grid.shape[0], self.vel_grid.shape[0], self.pos_off, self.vel_off,
                        1 if self.write_scores else 0, 1 if self.weighted_mean else 0, 1 if self.reference_pair else 0, 0)
        _check(lib().dpe_bcm_create(C.byref(cfg), C.byref(self._h)))
        self._bind_outputs()
        return 0

    def Update(self, CodeScores, CarrScores, win, chan, stream=None):
        """win: BCM_WINDOW_DTYPE [W]; chan: CHAN_END_DTYPE [W,K]; banks: device pointers from BCS."""
        if not self.Started:
            raise DpeError("[BatchCorrManifold] Error: Update() Failed due to module not initialized")
        win = np.ascontiguousarray(np.atleast_1d(win))
        chan = np.ascontiguousarray(chan)
        if chan.ndim == 1:
            chan = chan[None, :]
        W, K = chan.shape
        assert win.shape[0] == W
        _check(lib().dpe_bcm_update(self._h, _ptr(CodeScores), _ptr(CarrScores), C.c_int32(W), C.c_int32(K),
                                    win.ctypes.data_as(C.POINTER(BcmWindow)),
                                    chan.ctypes.data_as(C.POINTER(ChanEnd)), _stream(stream)))
        self._W = W
        keys = C.c_void_p()   # the key sets alternate between Updates: refresh the device pointer
        _check(lib().dpe_bcm_keys(self._h, C.byref(keys)))
        self.Keys = keys.value
        return 0

    def UpdateDev(self, CodeScores, CarrScores, n_chan, ports, dim_t, rx_time, stream=None):
        """One window with the inputs in DEVICE arrays (dpe_bcm_update_dev): ports = {field: device pointer} with the
        pointer fields of dpe_bcm_ports_dev."""
        if not self.Started:
            raise DpeError("[BatchCorrManifold] Error: Update() Failed due to module not initialized")
        p = BcmPortsDev(dimT=int(dim_t), reserved=0, **{k: _ptr(v).value for k, v in ports.items()})
        _check(lib().dpe_bcm_update_dev(self._h, _ptr(CodeScores), _ptr(CarrScores), C.c_int32(n_chan), C.byref(p),
                                        C.c_double(rx_time), _stream(stream)))
        self._W = 1
        keys = C.c_void_p()
        _check(lib().dpe_bcm_keys(self._h, C.byref(keys)))
        self.Keys = keys.value
        return 0

    def UpdatePrepared(self, CodeScores, CarrScores, n_chan, stream=None):
        """One window whose coefficient blocks an attached ChanMgrDev has already written on the device."""
        _check(lib().dpe_bcm_update_prepared(self._h, _ptr(CodeScores), _ptr(CarrScores), C.c_int32(n_chan), _stream(stream)))
        self._W = 1
        return 0

    def results(self, stream=None):
        res = (BcmResult * self._W)()
        _check(lib().dpe_bcm_results(self._h, res, _stream(stream)))
        return [dict(zVal=np.array(r.zVal), RVal=np.eye(8), posIndex=r.posIndex, velIndex=r.velIndex,
                     posScore=r.posScore, velScore=r.velScore, posOutOfWindow=r.posOutOfWindow,
                     velOutOfWindow=r.velOutOfWindow, zValMean=np.array(r.zValMean),
                     weightedSums=np.array([list(r.weightedSums[0]), list(r.weightedSums[1])])) for r in res]

    def exchange_keys(self, comm, stream=None, to_host=True):
        """All-reduce(MAX) of the last Update's packed keys across the ranks of `comm` (dpe_bcm_exchange_keys);
        returns the reduced host keys [W, 2] for results_from_keys (to_host=False: in place on the device only, asynchronous)."""
        if not to_host:
            _check(lib().dpe_bcm_exchange_keys(self._h, comm._h, None, _stream(stream)))
            return None
        keys = np.zeros((self._W, 2), dtype=np.uint64)
        _check(lib().dpe_bcm_exchange_keys(self._h, comm._h, keys.ctypes.data_as(C.POINTER(C.c_uint64)), _stream(stream)))
        return keys

    def results_from_keys(self, keys_host, pos_grid_global, vel_grid_global):
        keys_host = np.ascontiguousarray(keys_host, dtype=np.uint64)
        W = keys_host.shape[0]
        pg = np.ascontiguousarray(pos_grid_global, dtype=np.float64)
        vg = np.ascontiguousarray(vel_grid_global, dtype=np.float64)
        res = (BcmResult * W)()
        _check(lib().dpe_bcm_results_from_keys(self._h, keys_host.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int32(W),
                                               pg.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(pg.shape[0]),
                                               vg.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(vg.shape[0]), res))
        return [dict(zVal=np.array(r.zVal), posIndex=r.posIndex, velIndex=r.velIndex, posScore=r.posScore,
                     velScore=r.velScore) for r in res]

    def read_scores(self, stream=None):
        Gp, Gv = self.pos_grid.shape[0], self.vel_grid.shape[0]
        ps = d2h(self.PosScores, self._W * self.PosScoresPitch * 4, np.float32, stream).reshape(self._W, self.PosScoresPitch)[:, :Gp]
        vs = d2h(self.VelScores, self._W * self.VelScoresPitch * 4, np.float32, stream).reshape(self._W, self.VelScoresPitch)[:, :Gv]
        return np.ascontiguousarray(ps), np.ascontiguousarray(vs)

    def export_scores_f64(self, window, pos_dev, vel_dev, stream=None):
        """PosScores / its velocity twin as the reference's dense fp64 rows (dpe_bcm_export_scores_f64)."""
        _check(lib().dpe_bcm_export_scores_f64(self._h, C.c_int32(window), _ptr(pos_dev) if pos_dev is not None else None,
                                               _ptr(vel_dev) if vel_dev is not None else None, _stream(stream)))

    def set_graph(self, enable=True):
        """Replay repeated Updates as one hipGraph launch (needs a created stream, see dpe_hip.h)."""
        _check(lib().dpe_bcm_set_graph(self._h, C.c_int32(1 if enable else 0)))

    def profile(self, enable=True):
        ms, cnt = (C.c_float * 2)(), (C.c_int32 * 2)()
        _check(lib().dpe_bcm_profile(self._h, C.c_int32(1 if enable else 0), ms, cnt))
        return {"bcm_scan": (ms[0], cnt[0])}   # one fused launch scores both manifolds

    def Stop(self):
        if self.Started:
            if getattr(self, "_owned", True):      # (a lane of a Pipe is destroyed by the pipe)
                _check(lib().dpe_bcm_destroy(self._h))
            self._h = C.c_void_p(None)
            self.Started = False
        return 0

    def _bind_outputs(self):
        self.PosScores = self.VelScores = None
        if self.write_scores:
            ps, vs = C.c_void_p(), C.c_void_p()
            _check(lib().dpe_bcm_scores(self._h, C.byref(ps), C.byref(vs)))
            self.PosScores, self.VelScores = ps.value, vs.value
        pp, vp = C.c_int64(), C.c_int64()
        _check(lib().dpe_bcm_scores_pitch(self._h, C.byref(pp), C.byref(vp)))
        self.PosScoresPitch, self.VelScoresPitch = pp.value, vp.value     # floats between the rows of consecutive windows
        keys = C.c_void_p()
        _check(lib().dpe_bcm_keys(self._h, C.byref(keys)))
        self.Keys = keys.value
        self.Started = True
        self._W = 0

    @classmethod
    def _adopt(cls, handle, *args, **kw):
        """Python face of a handle that a dpe_pipe owns."""
        self = cls(*args, **kw)
        self._h, self._owned = C.c_void_p(handle), False
        self._bind_outputs()
        return self

    def __del__(self):
        try:            # at interpreter shutdown module globals may already be gone
            self.Stop()
        except Exception:
            pass


class Pipe:
    """dpe_pipe: `in_flight` batches of the BatchCorrScores -> BatchCorrManifold path on the device at once (the reference's
    overlap of ingest and compute: sampleblock.cu:327-447, batchcorrscores.h:60-64).  submit() deals a batch to the next lane and
    returns its ticket; results(ticket) waits for that batch only.  lane(ticket) -> (BatchCorrScores, BatchCorrManifold, stream)
    of the batch (banks, scores, keys), valid until `in_flight` later batches have been issued."""

    def __init__(self, SamplingFrequency, samples_per_window, pos_grid, vel_grid, lag_half_width=8, bin_half_width=48,
                 max_windows=1, max_channels=8, in_flight=2, LPower=1, write_scores=True, pos_index_offset=0, vel_index_offset=0,
                 bcm_max_windows=None, weighted_mean=False):
        self.fs, self.S = float(SamplingFrequency), int(samples_per_window)
        self.L, self.B = int(lag_half_width), int(bin_half_width)
        self.max_windows, self.max_channels = int(max_windows), int(max_channels)
        self.bcm_max_windows = int(bcm_max_windows) if bcm_max_windows else self.max_windows   # (stage 1 sharded by window: the scan sees all)
        self.pos_grid = np.ascontiguousarray(pos_grid, dtype=np.float64)
        self.vel_grid = np.ascontiguousarray(vel_grid, dtype=np.float64)
        self._bcm_kw = dict(LPower=int(LPower), lag_half_width=self.L, bin_half_width=self.B, max_windows=self.bcm_max_windows,
                            max_channels=self.max_channels, write_scores=bool(write_scores), pos_index_offset=int(pos_index_offset),
                            vel_index_offset=int(vel_index_offset), weighted_mean=bool(weighted_mean))
        bcs = BcsConfig(self.S, self.L, self.B, self.max_windows, self.max_channels, 0, self.fs)
        dp = C.POINTER(C.c_double)
        bcm = BcmConfig(self.S, self.L, self.B, int(LPower), self.bcm_max_windows, self.max_channels, carr_fft_len(self.S), self.fs,
                        self.pos_grid.ctypes.data_as(dp), self.vel_grid.ctypes.data_as(dp), self.pos_grid.shape[0],
                        self.vel_grid.shape[0], int(pos_index_offset), int(vel_index_offset), 1 if write_scores else 0,
                        1 if weighted_mean else 0, 0, 0)
        self._h = C.c_void_p(None)
        _check(lib().dpe_pipe_create(C.byref(bcs), C.byref(bcm), C.c_int32(in_flight), C.byref(self._h)))
        self.lanes = self.in_flight = int(in_flight)
        self._faces = {}      # handle pair -> (BatchCorrScores, BatchCorrManifold) faces of a lane
        self._nw = {}         # ticket -> (windows, channels) of the batches the lanes hold

    def set_in_flight(self, n):
        """Deal to the first n lanes only (1: one stream)."""
        _check(lib().dpe_pipe_set_in_flight(self._h, C.c_int32(n)))
        self.in_flight = int(n)

    def lane_at(self, i):
        """(BatchCorrScores, BatchCorrManifold, stream) of lane i, for set-up calls (profile, set_graph)."""
        b, m, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().dpe_pipe_lane_at(self._h, C.c_int32(i), C.byref(b), C.byref(m), C.byref(st)))
        return self._lane_faces(b, m, st)

    def _note(self, ticket, W, K):
        self._nw[ticket] = (W, K)
        for t in [t for t in self._nw if t <= ticket - self.lanes]:
            del self._nw[t]

    def _lane_faces(self, bcs_h, bcm_h, st):
        key = (bcs_h.value, bcm_h.value)
        if key not in self._faces:
            b = BatchCorrScores._adopt(bcs_h.value, self.fs, self.S, self.L, self.B, self.max_windows, self.max_channels)
            m = BatchCorrManifold._adopt(bcm_h.value, self.fs, self.S, b.NumFFTPoints, self.pos_grid, self.vel_grid, **self._bcm_kw)
            self._faces[key] = (b, m)
        b, m = self._faces[key]
        return b, m, st.value

    def submit(self, Samples, chan_start, win, chan_end, window_stride=None, stream=None):
        """One batch: Samples device int16 [W, 2 S]; chan_start CHAN_START_DTYPE [W, K]; win BCM_WINDOW_DTYPE [W]; chan_end
        CHAN_END_DTYPE [W, K].  `stream`: the stream that produced Samples (the lane waits for it on the device).  -> ticket"""
        cs = np.ascontiguousarray(chan_start)
        ce = np.ascontiguousarray(chan_end)
        win = np.ascontiguousarray(np.atleast_1d(win))
        if cs.ndim == 1:
            cs, ce = cs[None, :], ce[None, :]
        W, K = cs.shape
        assert ce.shape == (W, K) and win.shape[0] == W
        t = C.c_int64(-1)
        stride = self.S if window_stride is None else int(window_stride)
        _check(lib().dpe_pipe_submit(self._h, _ptr(Samples), C.c_int64(stride), C.c_int32(W), C.c_int32(K),
                                     cs.ctypes.data_as(C.POINTER(ChanStart)), win.ctypes.data_as(C.POINTER(BcmWindow)),
                                     ce.ctypes.data_as(C.POINTER(ChanEnd)), _stream(stream), C.byref(t)))
        self._note(t.value, W, K)
        return t.value

    def acquire(self, stream=None):
        """The next lane for a host that drives the two stages itself (multi-GPU exchanges in between):
        -> (ticket, BatchCorrScores, BatchCorrManifold, lane stream); finish with commit(ticket, n_windows)."""
        t, b, m, st = C.c_int64(-1), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().dpe_pipe_acquire(self._h, _stream(stream), C.byref(t), C.byref(b), C.byref(m), C.byref(st)))
        return (t.value,) + self._lane_faces(b, m, st)

    def mark_stage1(self, ticket):
        _check(lib().dpe_pipe_mark_stage1(self._h, C.c_int64(ticket)))

    def commit(self, ticket, n_windows, n_chan=None):
        _check(lib().dpe_pipe_commit(self._h, C.c_int64(ticket), C.c_int32(n_windows)))
        self._note(ticket, int(n_windows), n_chan)

    def lane(self, ticket):
        b, m, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().dpe_pipe_lane(self._h, C.c_int64(ticket), C.byref(b), C.byref(m), C.byref(st)))
        fb, fm, s = self._lane_faces(b, m, st)
        held = self._nw.get(ticket)
        if held:
            fm._W = held[0]
            fb._W, fb._K = min(held[0], self.max_windows), held[1] or fb._K
        return fb, fm, s

    def results(self, ticket):
        held = self._nw.get(ticket)
        W = held[0] if held else self.bcm_max_windows
        res = (BcmResult * W)()
        _check(lib().dpe_pipe_results(self._h, C.c_int64(ticket), res))
        return [dict(zVal=np.array(r.zVal), RVal=np.eye(8), posIndex=r.posIndex, velIndex=r.velIndex,
                     posScore=r.posScore, velScore=r.velScore, posOutOfWindow=r.posOutOfWindow,
                     velOutOfWindow=r.velOutOfWindow, zValMean=np.array(r.zValMean),
                     weightedSums=np.array([list(r.weightedSums[0]), list(r.weightedSums[1])])) for r in res]

    def samples_consumed(self, ticket, stream=None):
        _check(lib().dpe_pipe_samples_consumed(self._h, C.c_int64(ticket), _stream(stream)))

    def join(self, stream=None):
        _check(lib().dpe_pipe_join(self._h, _stream(stream)))

    def synchronize(self):
        _check(lib().dpe_pipe_synchronize(self._h))

    def close(self):
        if self._h:
            for b, m in self._faces.values():
                b.Stop(); m.Stop()          # (faces only: the pipe owns the handles)
            self._faces = {}
            _check(lib().dpe_pipe_destroy(self._h))
            self._h = C.c_void_p(None)

    Stop = close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChmConfig(C.Structure):
    _fields_ = [("nChan", C.c_int32), ("dopplerSign", C.c_int32), ("sampleLength", C.c_double), ("rxTime", C.c_double)]


CHM_INIT_DTYPE = np.dtype([("prn", "<i4"), ("cpElapsed", "<i4"), ("cpReference", "<i4"), ("cpRefTOW", "<i4"),
                           ("codePhase", "<f8"), ("carrierPhase", "<f8"), ("codeFrequency", "<f8"),
                           ("carrierFrequency", "<f8"), ("eph", "<f8", (21,))])


class ChanMgr:
    """Module "cuChanMgr" (cuchanmgr.cu:930-1268), host fp64.  Start()/Update() then outputs()
    returns (chan_start[K], chan_end[K], bcm_window[1]) ready for BatchCorrScores / BatchCorrManifold."""

    def __init__(self, prn, rc, ri, fc, fi, cp, cp_ref, cp_ref_tow, eph, rx_time, T, DopplerSign=1):
        K = len(prn)
        init = np.zeros(K, dtype=CHM_INIT_DTYPE)
        init["prn"], init["cpElapsed"], init["cpReference"], init["cpRefTOW"] = prn, cp, cp_ref, cp_ref_tow
        init["codePhase"], init["carrierPhase"], init["codeFrequency"], init["carrierFrequency"] = rc, ri, fc, fi
        init["eph"] = eph
        self.K = K
        self._h = C.c_void_p(None)
        cfg = ChmConfig(K, int(DopplerSign), float(T), float(rx_time))
        _check(lib().dpe_chm_create(C.byref(cfg), init.ctypes.data_as(C.c_void_p), C.byref(self._h)))

    @classmethod
    def from_handoff(cls, ho, T, K=None):
        sl = slice(0, K)
        return cls(ho["prn_list"][sl], ho["rc"][sl], ho["ri"][sl], ho["fc"][sl], ho["fi"][sl], ho["cp"][sl],
                   ho["cp_timestamp"][sl], ho["TOW"][sl], ho["eph"][sl], ho["rxTime"], T)

    def _step(self, fn, x_k1k1, x_kk1, time_grid):
        a = np.ascontiguousarray(x_k1k1, dtype=np.float64)
        b = np.ascontiguousarray(x_kk1, dtype=np.float64)
        tg = np.ascontiguousarray(time_grid, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _check(fn(self._h, a.ctypes.data_as(dp), b.ctypes.data_as(dp), tg.ctypes.data_as(dp), C.c_int32(tg.size)))
        self._dimT = tg.size

    def Start(self, x_k1k1, x_kk1, time_grid=(0.0,)):
        self._step(lib().dpe_chm_start, x_k1k1, x_kk1, time_grid)
        return 0

    def Update(self, x_k1k1, x_kk1, time_grid=(0.0,)):
        self._step(lib().dpe_chm_update, x_k1k1, x_kk1, time_grid)
        return 0

    def outputs(self, with_batch=False):
        start = np.zeros(self.K, dtype=CHAN_START_DTYPE)
        end = np.zeros(self.K, dtype=CHAN_END_DTYPE)
        win = np.zeros(1, dtype=BCM_WINDOW_DTYPE)
        batch = np.zeros((self.K, self._dimT, 8)) if with_batch else None
        _check(lib().dpe_chm_outputs(self._h, start.ctypes.data_as(C.c_void_p), end.ctypes.data_as(C.c_void_p),
                                     win.ctypes.data_as(C.c_void_p),
                                     batch.ctypes.data_as(C.c_void_p) if with_batch else None))
        return (start, end, win, batch) if with_batch else (start, end, win)

    def Stop(self):
        if self._h:
            lib().dpe_chm_destroy(self._h)
            self._h = C.c_void_p(None)
        return 0

    def __del__(self):
        try:            # at interpreter shutdown module globals may already be gone
            self.Stop()
        except Exception:
            pass


class FixRecord(C.Structure):     # dpe_fix_record
    _fields_ = [("seq", C.c_uint64), ("zVal", C.c_double * 8), ("rxTime", C.c_double), ("posIndex", C.c_int64), ("velIndex", C.c_int64),
                ("posOutOfWindow", C.c_int64), ("velOutOfWindow", C.c_int64), ("posScore", C.c_float), ("velScore", C.c_float),
                ("status", C.c_int32), ("reserved", C.c_int32)]


class ChanMgrDev:
    """Module "cuChanMgr" as the reference has it: state and port arrays in DEVICE memory, one small kernel per window
    (dpe_chm_dev_*; cuchanmgr.cu:1100-1132,1237-1264).  attach(bcs, bcm) makes it write their parameter blocks and form the
    measurement from the scan's keys: the closed loop is then bcs.UpdatePrepared -> bcm.UpdatePrepared -> step()."""

    def __init__(self, prn, rc, ri, fc, fi, cp, cp_ref, cp_ref_tow, eph, rx_time, T, time_grid=(0.0,), DopplerSign=1):
        K = len(prn)
        init = np.zeros(K, dtype=CHM_INIT_DTYPE)
        init["prn"], init["cpElapsed"], init["cpReference"], init["cpRefTOW"] = prn, cp, cp_ref, cp_ref_tow
        init["codePhase"], init["carrierPhase"], init["codeFrequency"], init["carrierFrequency"] = rc, ri, fc, fi
        init["eph"] = eph
        self.K = K
        tg = np.ascontiguousarray(time_grid, dtype=np.float64)
        self._dimT = tg.size
        self._h = C.c_void_p(None)
        cfg = ChmConfig(K, int(DopplerSign), float(T), float(rx_time))
        _check(lib().dpe_chm_dev_create(C.byref(cfg), init.ctypes.data_as(C.c_void_p), tg.ctypes.data_as(C.c_void_p),
                                        C.c_int32(tg.size), C.byref(self._h)))

    @classmethod
    def from_handoff(cls, ho, T, K=None, time_grid=(0.0,)):
        sl = slice(0, K)
        return cls(ho["prn_list"][sl], ho["rc"][sl], ho["ri"][sl], ho["fc"][sl], ho["fi"][sl], ho["cp"][sl],
                   ho["cp_timestamp"][sl], ho["TOW"][sl], ho["eph"][sl], ho["rxTime"], T, time_grid)

    def attach(self, bcs=None, bcm=None, ring_depth=64):
        _check(lib().dpe_chm_dev_attach(self._h, bcs._h if bcs is not None else None, bcm._h if bcm is not None else None,
                                        C.c_int32(ring_depth)))

    def set_shard(self, comm, pos_grid_global, vel_grid_global):
        """dpe_chm_dev_set_shard: the attached BatchCorrManifold scans a shard; step() all-reduces the keys over `comm` before the
        measurement kernel, which decodes them against these GLOBAL grids ([G, 4] each)."""
        p = np.ascontiguousarray(pos_grid_global, dtype=np.float64)
        v = np.ascontiguousarray(vel_grid_global, dtype=np.float64)
        _check(lib().dpe_chm_dev_set_shard(self._h, comm._h, p.ctypes.data_as(C.c_void_p), C.c_int64(p.shape[0]),
                                           v.ctypes.data_as(C.c_void_p), C.c_int64(v.shape[0])))

    def set_ekf(self, T, x0, P0=None, couple_velocity=True):
        """dpe_chm_dev_set_ekf: cuEKF's filter (EnableEKF = true) inside the measurement kernel instead of the pass-through."""
        cfg = EkfConfig(float(T), 1 if couple_velocity else 0, 0)
        cfg.x0[:] = [float(v) for v in np.asarray(x0, dtype=np.float64)]
        cfg.P0[:] = [float(v) for v in (np.eye(8) if P0 is None else np.asarray(P0, dtype=np.float64)).reshape(64)]
        _check(lib().dpe_chm_dev_set_ekf(self._h, C.byref(cfg)))

    def ports(self):
        """-> (BcsPortsDev, BcmPortsDev, rxTime_dev, xk1k1_dev, xkk1_dev, zVal_dev): raw device pointers."""
        b, m = BcsPortsDev(), BcmPortsDev()
        rx, x1, xk, z = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().dpe_chm_dev_ports(self._h, C.byref(b), C.byref(m), C.byref(rx), C.byref(x1), C.byref(xk), C.byref(z)))
        return b, m, rx.value, x1.value, xk.value, z.value

    def Start(self, x0, stream=None):
        a = np.ascontiguousarray(x0, dtype=np.float64)
        _check(lib().dpe_chm_dev_start(self._h, a.ctypes.data_as(C.c_void_p), _stream(stream)))
        return 0

    def Update(self, x_k1k1_dev, x_kk1_dev, stream=None):
        _check(lib().dpe_chm_dev_update(self._h, _ptr(x_k1k1_dev), _ptr(x_kk1_dev), _stream(stream)))
        return 0

    def step(self, stream=None):
        _check(lib().dpe_chm_dev_step(self._h, _stream(stream)))
        return 0

    def fix(self, window, timeout_us=-1):
        """-> dict of the window's fix, or None when it has not arrived within timeout_us (>= 0)."""
        r = FixRecord()
        rc = lib().dpe_chm_dev_fix(self._h, C.c_int64(window), C.byref(r), C.c_int32(timeout_us))
        if rc == 1:
            return None
        _check(rc)
        return dict(zVal=np.array(r.zVal[:]), rxTime=r.rxTime, posIndex=r.posIndex, velIndex=r.velIndex, posScore=r.posScore,
                    velScore=r.velScore, posOutOfWindow=r.posOutOfWindow, velOutOfWindow=r.velOutOfWindow, status=r.status)

    def outputs(self, with_batch=False, stream=None):
        start = np.zeros(self.K, dtype=CHAN_START_DTYPE)
        end = np.zeros(self.K, dtype=CHAN_END_DTYPE)
        win = np.zeros(1, dtype=BCM_WINDOW_DTYPE)
        batch = np.zeros((self.K, self._dimT, 8))
        status = C.c_int32(0)
        _check(lib().dpe_chm_dev_read(self._h, start.ctypes.data_as(C.c_void_p), end.ctypes.data_as(C.c_void_p),
                                      win.ctypes.data_as(C.c_void_p), batch.ctypes.data_as(C.c_void_p), C.byref(status), _stream(stream)))
        self.status = status.value
        return (start, end, win, batch) if with_batch else (start, end, win)

    def Stop(self):
        if self._h:
            lib().dpe_chm_dev_destroy(self._h)
            self._h = C.c_void_p(None)
        return 0

    def __del__(self):
        try:
            self.Stop()
        except Exception:
            pass


class AcqConfig(C.Structure):
    _fields_ = [("samplesPerWindow", C.c_int32), ("nCodePeriods", C.c_int32), ("nBins", C.c_int32), ("nPrn", C.c_int32),
                ("mode", C.c_int32), ("prnChunk", C.c_int32), ("samplingFrequency", C.c_double),
                ("binStartHz", C.c_double), ("binStepHz", C.c_double), ("dopplerSign", C.c_double),
                ("prn", C.c_int32 * 37), ("reserved", C.c_int32)]


class AcqResult(C.Structure):
    _fields_ = [("prn", C.c_int32), ("found", C.c_int32), ("maxCodeIdx", C.c_int32), ("maxDoppIdx", C.c_int32),
                ("rc", C.c_double), ("fc", C.c_double), ("fi", C.c_double), ("cppr", C.c_double), ("cppm", C.c_double),
                ("peak", C.c_double)]


class AcqFineResult(C.Structure):
    _fields_ = [("prn", C.c_int32), ("maxCarrIdx", C.c_int32), ("rc", C.c_double), ("ri", C.c_double), ("fc", C.c_double),
                ("fi", C.c_double), ("peakRe", C.c_double), ("peakIm", C.c_double)]


class AcqTrackInit(C.Structure):
    _fields_ = [("prn", C.c_int32), ("found", C.c_int32), ("fromSecondWindow", C.c_int32), ("reserved", C.c_int32),
                ("rc", C.c_double), ("ri", C.c_double), ("fc", C.c_double), ("fi", C.c_double), ("cppr", C.c_double),
                ("cppm", C.c_double), ("cppmWindow", C.c_double * 2)]


class Acquisition:
    """Coarse acquisition over `prns` x Doppler bins x all code delays of one window.
    mode: "coherent" / "noncoherent" = Correlator.coarse_acquisition(coherent=True/False)
    (correlator.py:53-103); "textbook" = 1 ms coherent x N non-coherent (not in the reference)."""
    MODES = {"coherent": 0, "noncoherent": 1, "textbook": 2}

    def __init__(self, SamplingFrequency, samples_per_window, prns, bins_hz, mode="coherent", prn_chunk=0, ds=1.0):
        bins_hz = np.asarray(bins_hz, dtype=np.float64)
        step = float(bins_hz[1] - bins_hz[0]) if bins_hz.size > 1 else 0.0
        assert bins_hz.size == 1 or np.allclose(np.diff(bins_hz), step), "bins must be equally spaced"
        self.S, self.fs = int(samples_per_window), float(SamplingFrequency)
        self.N = int(round(self.S / self.fs / 1e-3))
        self.M = self.S // self.N
        self.prns, self.bins = [int(p) for p in prns], bins_hz
        cfg = AcqConfig(self.S, self.N, bins_hz.size, len(self.prns), self.MODES[mode], int(prn_chunk), self.fs,
                        float(bins_hz[0]), step, float(ds), (C.c_int32 * 37)(*self.prns), 0)
        self._h = C.c_void_p(None)
        _check(lib().dpe_acq_create(C.byref(cfg), C.byref(self._h)))
        surf, mp = C.c_void_p(), C.c_void_p()
        _check(lib().dpe_acq_surface(self._h, C.byref(surf), C.byref(mp)))
        self.Surface, self.MaxPerCode = surf.value, mp.value

    def search(self, Samples, stream=None):
        _check(lib().dpe_acq_search(self._h, _ptr(Samples), _stream(stream)))

    def results(self, stream=None):
        res = (AcqResult * len(self.prns))()
        _check(lib().dpe_acq_results(self._h, res, _stream(stream)))
        return [dict(prn=r.prn, found=bool(r.found), max_code_idx=r.maxCodeIdx, max_dopp_idx=r.maxDoppIdx, rc=r.rc, fc=r.fc,
                     fi=r.fi, cppr=r.cppr, cppm=r.cppm, peak=r.peak) for r in res]

    def fine(self, Samples, coarse, stream=None):
        """Correlator.fine_frequency_acquisition (correlator.py:105-133) for every PRN, from `coarse` = results()."""
        res = (AcqResult * len(self.prns))()
        for r, c in zip(res, coarse):
            r.prn, r.rc, r.fc, r.fi = c["prn"], c["rc"], c["fc"], c["fi"]
        out = (AcqFineResult * len(self.prns))()
        _check(lib().dpe_acq_fine(self._h, _ptr(Samples), res, out, _stream(stream)))
        return [dict(prn=r.prn, max_carr_idx=r.maxCarrIdx, rc=r.rc, ri=r.ri, fc=r.fc, fi=r.fi,
                     peak=complex(r.peakRe, r.peakIm)) for r in out]

    def search_signal(self, Samples, stream=None):
        """Correlator.search_signal (correlator.py:38-51): coarse then fine; one dict per PRN."""
        self.search(Samples, stream)
        coarse = self.results(stream)
        fine = self.fine(Samples, coarse, stream)
        return [dict(found=c["found"], rc=f["rc"], ri=f["ri"], fc=f["fc"], fi=f["fi"], cppr=c["cppr"], cppm=c["cppm"],
                     max_carr_idx=f["max_carr_idx"], max_code_idx=c["max_code_idx"], max_dopp_idx=c["max_dopp_idx"])
                for c, f in zip(coarse, fine)]

    def scalar_acquisition(self, Window0, Window1, stream=None):
        """Receiver.scalar_acquisition (receiver.py:452-520) on two consecutive windows (device buffers)."""
        out = (AcqTrackInit * len(self.prns))()
        _check(lib().dpe_acq_scalar_acquisition(self._h, _ptr(Window0), _ptr(Window1), out, _stream(stream)))
        return [dict(prn=r.prn, found=bool(r.found), from_second_window=bool(r.fromSecondWindow), rc=r.rc, ri=r.ri, fc=r.fc,
                     fi=r.fi, cppr=r.cppr, cppm=r.cppm, cppm_window=(r.cppmWindow[0], r.cppmWindow[1])) for r in out]

    def read_surface(self, stream=None):
        n = len(self.prns) * self.bins.size * self.M
        return d2h(self.Surface, n * 4, np.float32, stream).reshape(len(self.prns), self.bins.size, self.M)

    def close(self):
        if self._h:
            lib().dpe_acq_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:            # at interpreter shutdown module globals may already be gone
            self.close()
        except Exception:
            pass


class EkfConfig(C.Structure):
    _fields_ = [("sampleLength", C.c_double), ("coupleVelocity", C.c_int32), ("reserved", C.c_int32),
                ("x0", C.c_double * 8), ("P0", C.c_double * 64)]


class cuEKF:
    """Module "cuEKF" (cudarecv/modules/src/cuekf.cu): EnableEKF=False passes zVal through (EKF_PassMeas :147-159,
    the shipped flow); EnableEKF=True runs StepUpdate then StepPredict per Update (:560-599), host fp64."""

    def __init__(self, InitX, InitP=None, SampleLength=0.02, EnableEKF=False, couple_velocity=True):
        self.EnableEKF = bool(EnableEKF)
        self.xCurrk1k1 = np.array(InitX, dtype=np.float64).copy()
        self.xCurrkk1 = self.xCurrk1k1.copy()
        self._h = C.c_void_p(None)
        if self.EnableEKF:
            cfg = EkfConfig()
            cfg.sampleLength = float(SampleLength)
            cfg.coupleVelocity = 1 if couple_velocity else 0
            P0 = np.eye(8) if InitP is None else np.asarray(InitP, dtype=np.float64).reshape(8, 8)
            for i in range(8):
                cfg.x0[i] = float(self.xCurrk1k1[i])
            for i in range(64):
                cfg.P0[i] = float(P0.ravel()[i])
            _check(lib().dpe_ekf_create(C.byref(cfg), C.byref(self._h)))

    def Update(self, zVal, RVal=None):
        z = np.ascontiguousarray(zVal, dtype=np.float64)
        if not self.EnableEKF:
            self.xCurrk1k1 = z.copy(); self.xCurrkk1 = z.copy()
            return 0
        R = np.ascontiguousarray(np.eye(8) if RVal is None else RVal, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _check(lib().dpe_ekf_step_update(self._h, z.ctypes.data_as(dp), R.ctypes.data_as(dp)))
        _check(lib().dpe_ekf_step_predict(self._h))
        st = self.state()
        self.xCurrk1k1, self.xCurrkk1 = st["xk1k1"], st["xkk1"]
        return 0

    def step_update(self, zVal, RVal):
        z = np.ascontiguousarray(zVal, dtype=np.float64); R = np.ascontiguousarray(RVal, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _check(lib().dpe_ekf_step_update(self._h, z.ctypes.data_as(dp), R.ctypes.data_as(dp)))

    def step_predict(self):
        _check(lib().dpe_ekf_step_predict(self._h))

    def state(self):
        a = {k: np.zeros(n) for k, n in (("xk1k1", 8), ("xkk1", 8), ("Pk1k1", 64), ("Pkk1", 64), ("Q", 64), ("K", 64))}
        dp = C.POINTER(C.c_double)
        _check(lib().dpe_ekf_state(self._h, *[a[k].ctypes.data_as(dp) for k in ("xk1k1", "xkk1", "Pk1k1", "Pkk1", "Q", "K")]))
        return {k: (v if v.size == 8 else v.reshape(8, 8)) for k, v in a.items()}

    def Stop(self):
        if self._h:
            lib().dpe_ekf_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:            # at interpreter shutdown module globals may already be gone
            self.Stop()
        except Exception:
            pass


class HipEventTimer:
    """HIP events recorded on the stream the kernels are launched on (bench.py roofline leg)."""

    def __init__(self):
        self.a, self.b = C.c_void_p(), C.c_void_p()
        _check(lib().dpe_event_create(C.byref(self.a)))
        _check(lib().dpe_event_create(C.byref(self.b)))

    def start(self, stream=None):
        _check(lib().dpe_event_record(self.a, _stream(stream)))

    def stop(self, stream=None):
        _check(lib().dpe_event_record(self.b, _stream(stream)))

    def elapsed_ms(self):
        ms = C.c_float()
        _check(lib().dpe_event_elapsed_ms(self.a, self.b, C.byref(ms)))
        return ms.value
