"""ctypes loader for the two CPU checkers (svdf_oracle.h API).

TEST INFRASTRUCTURE, NOT PRODUCT.  Import only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  ``kind="port"`` loads oracle/libsvdf_oracle.so (the plain-C
restatement), ``kind="reference"`` loads oracle/_ref/libsvdf_ref.so (the reference's own classes,
compiled from /root/reference by oracle/Makefile).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "libsvdf_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsvdf_ref.so")
REF_FULL_SO = os.path.join(HERE, "_ref", "libsvdf_ref_full.so")   # the reference's default factory: variant solvers (extend_type 2 / 15)

VIEW = {"u_bias": 0, "W_user": 1, "i_bias": 2, "W_item": 3, "g_bias": 4, "ufeedback_bias": 5, "W_ufeedback": 6}

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the checkers (building the checker is not using it)."""
    if force or not os.path.exists(PORT_SO) or (
            os.path.exists("/root/reference/apex_svd.h") and not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", HERE], stdout=subprocess.DEVNULL)


def have_reference():
    return os.path.exists(REF_SO)


def have_reference_full():
    return os.path.exists(REF_FULL_SO)


_libs = {}


def _load(kind):
    if kind in _libs:
        return _libs[kind]
    path = PORT_SO if kind == "port" else (REF_FULL_SO if kind == "reference_full" else REF_SO)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    P = C.c_void_p
    lib.svdo_create.restype = P
    lib.svdo_create.argtypes = [C.c_int] * 4
    lib.svdo_destroy.argtypes = [P]
    lib.svdo_set_param.argtypes = [P, C.c_char_p, C.c_char_p]
    lib.svdo_seed.argtypes = [C.c_uint]
    for f in ("svdo_init_model", "svdo_init_trainer", "svdo_finish_round"):
        getattr(lib, f).argtypes = [P]
    lib.svdo_set_round.argtypes = [P, C.c_int]
    lib.svdo_save_model_path.argtypes = [P, C.c_char_p, C.c_int]
    lib.svdo_load_model_path.argtypes = [P, C.c_char_p, C.c_int]
    lib.svdo_update_csr.argtypes = [P, C.c_float, C.c_int, C.c_int, C.c_int, _u32p, _f32p]
    lib.svdo_predict_csr.argtypes = lib.svdo_update_csr.argtypes
    lib.svdo_predict_csr.restype = C.c_float
    lib.svdo_update_csr_batch.argtypes = [P, C.c_int, _f32p, _i32p, _u32p, _f32p]
    lib.svdo_predict_csr_batch.argtypes = [P, C.c_int, _f32p, _i32p, _u32p, _f32p, _f32p]
    lib.svdo_update_csr_batch_stale.argtypes = [P, C.c_int, _f32p, _i32p, _u32p, _f32p, _f32p, _f32p, _f32p]
    lib.svdo_update_csr_batch_stale.restype = C.c_int
    if hasattr(lib, "svdo_update_window_substeps"):   # (the port only: the compiled reference's shim has no window steps beyond the stale one)
        lib.svdo_update_window_substeps.argtypes = [P, C.c_int, _f32p, _i32p, _u32p, _f32p, C.c_int]
        lib.svdo_update_window_substeps.restype = C.c_int
    lib.svdo_update_block_stale.argtypes = [P, C.c_int, C.c_int, _u32p, _f32p, C.c_int, _f32p, _i32p, _u32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p]
    lib.svdo_update_block_stale.restype = C.c_int
    lib.svdo_set_stale_rounding.argtypes = [P, C.c_int]
    lib.svdo_set_stale_rounding.restype = None
    lib.svdo_update_block.argtypes = [P, C.c_int, C.c_int, _u32p, _f32p, C.c_int, _f32p, _i32p, _u32p, _f32p]
    lib.svdo_predict_block.argtypes = lib.svdo_update_block.argtypes + [_f32p]
    lib.svdo_get_view.argtypes = [P, C.c_int, _f32p, C.c_long]
    lib.svdo_get_view.restype = C.c_long
    lib.svdo_view_shape.argtypes = [P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.svdo_set_view.argtypes = [P, C.c_int, _f32p, C.c_long]
    lib.svdo_set_view.restype = C.c_long
    lib.svdo_kind.restype = C.c_int
    RP = C.c_void_p
    lib.svdo_ranker_create.restype = RP
    lib.svdo_ranker_create.argtypes = [C.c_int] * 4
    lib.svdo_ranker_destroy.argtypes = [RP]
    lib.svdo_ranker_set_param.argtypes = [RP, C.c_char_p, C.c_char_p]
    lib.svdo_ranker_load_model_path.argtypes = [RP, C.c_char_p, C.c_int]
    lib.svdo_ranker_init.argtypes = [RP, C.c_int]
    lib.svdo_ranker_process_csr.restype = C.c_long
    lib.svdo_ranker_process_csr.argtypes = [RP, C.c_float, C.c_int, C.c_int, C.c_int, _u32p, _f32p, _i32p, C.c_long]
    lib.svdo_ranker_process_block.restype = C.c_long
    lib.svdo_ranker_process_block.argtypes = [RP, C.c_int, C.c_int, _u32p, _f32p, C.c_int, _f32p, _i32p, _u32p, _f32p, _i32p, C.c_long]
    lib.svdo_sum_sq_err.restype = C.c_double
    lib.svdo_sum_sq_err.argtypes = [_f32p, _f32p, C.c_long, C.c_float]
    lib.svdo_libm_expf.argtypes = [C.c_void_p, C.c_uint, C.c_uint, _f32p, C.c_long]
    _libs[kind] = lib
    return lib


def libm_expf(x=None, first=0, step=1, n=None):
    """expf of the host libm (the reference's link functions call it) over an array, or over bit patterns first + j*step."""
    lib = _load("port")
    if x is not None:
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(len(x), np.float32)
        lib.svdo_libm_expf(x.ctypes.data_as(C.c_void_p), 0, 0, out, len(x))
        return out
    out = np.empty(n, np.float32)
    lib.svdo_libm_expf(None, first, step, out, n)
    return out


def _pad(a, dtype):
    """ctypes ndpointer rejects zero-length views of some arrays; always hand over >=1 element."""
    a = np.ascontiguousarray(a, dtype=dtype)
    return a if a.size else np.zeros(1, dtype=dtype)


class OracleTrainer:
    """ISVDTrainer-shaped handle on one of the CPU checkers (apex_svd.h:33-107)."""

    def __init__(self, kind="port", format_type=0, active_type=0, extend_type=0, variant_type=0, params=None):
        self.kind = kind
        self.lib = _load(kind)
        self.h = self.lib.svdo_create(format_type, active_type, extend_type, variant_type)
        for k, v in (params or {}).items():
            self.set_param(k, v)

    def close(self):
        if self.h:
            self.lib.svdo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_param(self, name, val):
        self.lib.svdo_set_param(self.h, str(name).encode(), str(val).encode())

    def seed(self, s):
        self.lib.svdo_seed(int(s))

    def init_model(self):
        self.lib.svdo_init_model(self.h)

    def init_trainer(self):
        self.lib.svdo_init_trainer(self.h)

    def set_round(self, r):
        self.lib.svdo_set_round(self.h, int(r))

    def finish_round(self):
        self.lib.svdo_finish_round(self.h)

    def save_model(self, path, with_type_header=True):
        assert self.lib.svdo_save_model_path(self.h, str(path).encode(), int(with_type_header)) == 0

    def load_model(self, path, with_type_header=True):
        assert self.lib.svdo_load_model_path(self.h, str(path).encode(), int(with_type_header)) == 0

    def update_csr(self, label, ng, nu, ni, index, value):
        self.lib.svdo_update_csr(self.h, float(label), ng, nu, ni, _pad(index, np.uint32), _pad(value, np.float32))

    def predict_csr(self, label, ng, nu, ni, index, value):
        return self.lib.svdo_predict_csr(self.h, float(label), ng, nu, ni, _pad(index, np.uint32), _pad(value, np.float32))

    def update_batch(self, d):
        self.lib.svdo_update_csr_batch(self.h, d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int32),
                                       _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32))

    def update_batch_stale(self, d, delta=None):
        """The window-minibatch checker step (svdf_oracle.h: svdo_update_csr_batch_stale): every row is the reference's
        update_inner on (current user side, window-start replicated side); the replicated side stays as it is and its change is
        accumulated in file order.  Returns (dW_item [num_item x k], di_bias, dg_bias); pass the previous result as `delta`
        to keep accumulating into it."""
        if delta is None:
            shp = {}
            for name in ("W_item", "i_bias", "g_bias"):
                rows, cols = C.c_int(), C.c_int()
                self.lib.svdo_view_shape(self.h, VIEW[name], C.byref(rows), C.byref(cols))
                shp[name] = (max(rows.value, 0), max(cols.value, 1))
            delta = (np.zeros(shp["W_item"], np.float32), np.zeros(shp["i_bias"][0], np.float32), np.zeros(shp["g_bias"][0], np.float32))
        dW, db, dg = delta
        rc = self.lib.svdo_update_csr_batch_stale(self.h, d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int32),
                                                  _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32),
                                                  dW.reshape(-1) if dW.size else np.zeros(1, np.float32), db if db.size else np.zeros(1, np.float32),
                                                  dg if dg.size else np.zeros(1, np.float32))
        assert rc == 0, "window-minibatch step: configuration not supported by this checker"
        return delta

    def update_window_substeps(self, d, sub=128):
        """One window of the one-GPU window step with ordered sub-steps on the item side (svdf_oracle.c: svdo_update_window_substeps): the model
        moves in place -- users exactly, against the window-start item rows; every item in sub-steps of at most `sub` of its rows, in file order."""
        rc = self.lib.svdo_update_window_substeps(self.h, d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int32),
                                                  _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32), int(sub))
        assert rc == 0, "window step with sub-steps: configuration / rows not supported by this checker"

    def set_stale_rounding(self, bf16):
        """checker steps: round row contributions to bfloat16 before summing (the HIP engine's amd:contrib = bf16)"""
        self.lib.svdo_set_stale_rounding(self.h, 1 if bf16 else 0)

    def stale_delta_zero(self):
        """zeroed delta arrays of the window-minibatch checker step for a user-group trainer:
        (dW_item, di_bias, dg_bias, dW_ufeedback, dufeedback_bias), unpadded"""
        shp = {}
        for name in ("W_item", "i_bias", "g_bias", "W_ufeedback", "ufeedback_bias"):
            rows, cols = C.c_int(), C.c_int()
            self.lib.svdo_view_shape(self.h, VIEW[name], C.byref(rows), C.byref(cols))
            shp[name] = (max(rows.value, 0), max(cols.value, 1))
        return (np.zeros(shp["W_item"], np.float32), np.zeros(shp["i_bias"][0], np.float32), np.zeros(shp["g_bias"][0], np.float32),
                np.zeros(shp["W_ufeedback"], np.float32), np.zeros(shp["ufeedback_bias"][0], np.float32))

    def update_block_stale(self, b, delta=None):
        """The window-minibatch checker step for one user-group block (svdf_oracle.h: svdo_update_block_stale): the reference's
        update(block) on (the user's private state, the window-start replicated side); the replicated side stays as it is and its change
        is accumulated in file order into `delta` (stale_delta_zero()), which is returned."""
        if delta is None:
            delta = self.stale_delta_zero()
        d = b.data
        z = np.zeros(1, np.float32)
        arrs = [x.reshape(-1) if x.size else z for x in delta]
        rc = self.lib.svdo_update_block_stale(self.h, b.num_ufeedback, b.extend_tag, _pad(b.index_ufeedback, np.uint32), _pad(b.value_ufeedback, np.float32),
                                              d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int32), _pad(d.feat_index, np.uint32),
                                              _pad(d.feat_value, np.float32), *arrs)
        assert rc == 0, "window-minibatch block step: configuration not supported by this checker"
        return delta

    def predict_batch(self, d):
        out = np.zeros(max(d.num_row, 1), dtype=np.float32)
        self.lib.svdo_predict_csr_batch(self.h, d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int32),
                                        _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32), out)
        return out[:d.num_row]

    def update_block(self, b):
        d = b.data
        self.lib.svdo_update_block(self.h, b.num_ufeedback, b.extend_tag, _pad(b.index_ufeedback, np.uint32),
                                   _pad(b.value_ufeedback, np.float32), d.num_row, _pad(d.row_label, np.float32),
                                   _pad(d.row_ptr, np.int32), _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32))

    def predict_block(self, b):
        d = b.data
        out = np.zeros(max(d.num_row, 1), dtype=np.float32)
        self.lib.svdo_predict_block(self.h, b.num_ufeedback, b.extend_tag, _pad(b.index_ufeedback, np.uint32),
                                    _pad(b.value_ufeedback, np.float32), d.num_row, _pad(d.row_label, np.float32),
                                    _pad(d.row_ptr, np.int32), _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32), out)
        return out[:d.num_row]

    def view(self, name):
        rows, cols = C.c_int(), C.c_int()
        self.lib.svdo_view_shape(self.h, VIEW[name], C.byref(rows), C.byref(cols))
        if rows.value < 0:
            return None
        out = np.zeros(max(rows.value * cols.value, 1), dtype=np.float32)
        n = self.lib.svdo_get_view(self.h, VIEW[name], out, out.size)
        assert n == rows.value * cols.value
        out = out[:n]
        return out.reshape(rows.value, cols.value) if cols.value > 1 or name.startswith("W_") else out

    def set_view(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32).ravel()
        assert self.lib.svdo_set_view(self.h, VIEW[name], _pad(a, np.float32), a.size) == a.size


def sum_sq_err(pred, label, scale=1.0, kind="port"):
    """RMSEEvaluator's accumulator (svd_feature_infer.cpp:38-56): sequential long double sum of ((pred - label) * scale)^2."""
    lib = _load(kind)
    pred, label = np.ascontiguousarray(pred, np.float32), np.ascontiguousarray(label, np.float32)
    return float(lib.svdo_sum_sq_err(_pad(pred, np.float32), _pad(label, np.float32), len(pred), float(scale)))


class OracleRanker:
    """ISVDRanker-shaped handle on one of the CPU checkers (apex_svd.h:160-197)."""

    def __init__(self, kind="port", format_type=0, active_type=0, extend_type=0, variant_type=0):
        self.lib = _load(kind)
        self.h = self.lib.svdo_ranker_create(format_type, active_type, extend_type, variant_type)
        self.cap = 16

    def close(self):
        if self.h:
            self.lib.svdo_ranker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_param(self, name, val):
        self.lib.svdo_ranker_set_param(self.h, str(name).encode(), str(val).encode())

    def load_model(self, path, with_type_header=True):
        assert self.lib.svdo_ranker_load_model_path(self.h, str(path).encode(), int(with_type_header)) == 0

    def init_ranker(self, num_item_set):
        self.cap = max(16, int(num_item_set) + 16)
        self.lib.svdo_ranker_init(self.h, int(num_item_set))

    def process(self, label, ng, nu, ni, index, value):
        out = np.zeros(self.cap, np.int32)
        n = self.lib.svdo_ranker_process_csr(self.h, float(label), ng, nu, ni, _pad(index, np.uint32), _pad(value, np.float32), out, self.cap)
        return out[:n].copy()

    def process_rows(self, d):
        """every row of a CSRData through process(); returns the concatenated results"""
        res = []
        for r in range(d.num_row):
            res.append(self.process(*d.row(r)))
        return np.concatenate(res) if res else np.zeros(0, np.int32)

    def process_block(self, b):
        d = b.data
        cap = self.cap * max(1, d.num_row)
        out = np.zeros(cap, np.int32)
        n = self.lib.svdo_ranker_process_block(self.h, b.num_ufeedback, b.extend_tag, _pad(b.index_ufeedback, np.uint32),
                                               _pad(b.value_ufeedback, np.float32), d.num_row, _pad(d.row_label, np.float32),
                                               _pad(d.row_ptr, np.int32), _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32), out, cap)
        return out[:n].copy()
