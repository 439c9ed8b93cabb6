"""svdfeature_amd -- MI355X-native engine for the apex_svd SGD hot path of Gnnng/SVDFeature.

The product is ``libsvdfeature_amd.so`` (hand-written gfx950 HIP kernels + C++ host engine behind
the C ABI of ``include/svdfeature_amd.h``).  This module is only the ctypes binding the tests and
bench.py use; ``Trainer`` mirrors the reference's ``ISVDTrainer`` surface (apex_svd.h:33-107)
method for method.  There is no CPU fallback: if the shared library is missing or no GPU is
visible, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

from .data import BlockArrays, CSRData, PlusBlock, pairs_as_csr  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsvdfeature_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "svdfeature_amd.h")

VIEW = {"u_bias": 0, "W_user": 1, "i_bias": 2, "W_item": 3, "g_bias": 4, "ufeedback_bias": 5, "W_ufeedback": 6}

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

_lib = None
_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]


class SvdfError(RuntimeError):
    pass


def load_library():
    """dlopen libsvdfeature_amd.so and declare every prototype of include/svdfeature_amd.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvdfError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64; a process must hold ONE HIP runtime, so when torch is
    # installed it is imported first and libsvdfeature_amd.so binds to the runtime torch loaded
    # (otherwise torch.cuda reports "no GPUs" after /opt/rocm's copy has been initialised).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    P = C.c_void_p
    lib.svdf_version.restype = C.c_char_p
    lib.svdf_last_error.restype = C.c_char_p
    lib.svdf_set_error_mode.argtypes = [C.c_int]
    lib.svdf_device_count.restype = C.c_int
    lib.svdf_create.restype = P
    lib.svdf_create.argtypes = [C.c_uint8] * 4 + [C.c_int]
    lib.svdf_destroy.argtypes = [P]
    lib.svdf_set_param.argtypes = [P, C.c_char_p, C.c_char_p]
    lib.svdf_seed.argtypes = [C.c_uint]
    for f in ("svdf_init_model", "svdf_init_trainer", "svdf_finish_round", "svdf_synchronize",
              "svdf_item_delta_begin", "svdf_item_delta_apply"):
        getattr(lib, f).argtypes = [P]
    lib.svdf_set_round.argtypes = [P, C.c_int]
    lib.svdf_load_model.argtypes = [P, C.c_void_p]
    lib.svdf_save_model.argtypes = [P, C.c_void_p]
    lib.svdf_save_model_begin.argtypes = [P, C.c_void_p]
    lib.svdf_save_model_end.argtypes = [P]
    lib.svdf_update_csr.argtypes = [P, C.c_float, C.c_int, C.c_int, C.c_int, _u32p, _f32p]
    lib.svdf_predict_csr.argtypes = lib.svdf_update_csr.argtypes
    lib.svdf_predict_csr.restype = C.c_float
    lib.svdf_update_csr_batch.argtypes = [P, C.c_int, _f32p, _i32p, _u32p, _f32p]
    lib.svdf_predict_csr_batch.argtypes = [P, C.c_int, _f32p, _i32p, _u32p, _f32p, _f32p]
    lib.svdf_update_block.argtypes = [P, C.c_int, C.c_int, _u32p, _f32p, C.c_int, _f32p, _i32p, _u32p, _f32p]
    lib.svdf_predict_block.argtypes = lib.svdf_update_block.argtypes + [_f32p]
    lib.svdf_dataset_from_csr.restype = P
    lib.svdf_dataset_from_csr.argtypes = [P, C.c_long, _f32p, _i64p, _u32p, _f32p]
    lib.svdf_dataset_from_triples.restype = P
    lib.svdf_dataset_from_triples.argtypes = [P, C.c_long, _u32p, _u32p, _f32p]
    lib.svdf_dataset_from_pairs.restype = P
    lib.svdf_dataset_from_pairs.argtypes = [P, C.c_long, _u32p, _u32p, _u32p]
    lib.svdf_dataset_window_from_triples.restype = P
    lib.svdf_dataset_window_from_triples.argtypes = [P, C.c_long, _u32p, _u32p, _f32p]
    lib.svdf_dataset_window_from_pairs.restype = P
    lib.svdf_dataset_window_from_pairs.argtypes = [P, C.c_long, _u32p, _u32p, _u32p]
    lib.svdf_dataset_window_from_csr.restype = P
    lib.svdf_dataset_window_from_csr.argtypes = [P, C.c_long, _f32p, _i64p, _u32p, _f32p]
    lib.svdf_dataset_window_from_blocks.restype = P
    lib.svdf_dataset_window_from_blocks.argtypes = [P, C.c_long, _i32p, _i64p, _u32p, _f32p, _i64p, _f32p, _i64p, _u32p, _f32p]
    lib.svdf_ipc_setup.argtypes = [P, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_char_p]
    lib.svdf_ipc_connect.argtypes = [P, C.c_char_p]
    lib.svdf_ipc_window_pack.argtypes = [P, P, C.c_int]
    lib.svdf_ipc_window_reduce.argtypes = [P, C.c_int]
    lib.svdf_ipc_window_apply.argtypes = [P, C.c_int]
    lib.svdf_ipc_block_send.argtypes = [P, C.c_int, C.c_int]
    lib.svdf_ipc_block_recv.argtypes = [P, C.c_int, C.c_int, C.c_uint]
    lib.svdf_ipc_status.argtypes = [P]
    lib.svdf_ipc_close.argtypes = [P]
    lib.svdf_stratum_step.argtypes = [P, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, P]
    lib.svdf_item_block_set_at.argtypes = [P, C.c_int, C.c_int, P]
    lib.svdf_rccl_unique_id.argtypes = [C.c_char_p]
    lib.svdf_rccl_init.argtypes = [P, C.c_char_p, C.c_int, C.c_int]
    lib.svdf_rccl_window_allreduce.argtypes = [P, P, C.c_int]
    lib.svdf_rccl_block_handoff.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.svdf_rccl_block_arrive.argtypes = [P, C.c_int]
    lib.svdf_rccl_counter.restype = C.c_int64
    lib.svdf_rccl_counter.argtypes = [P, C.c_int]
    lib.svdf_rccl_close.argtypes = [P]
    lib.svdf_window_delta_pack.argtypes = [P, P, P, C.c_int, C.POINTER(C.c_int64)]
    lib.svdf_window_delta_apply.argtypes = [P, P, C.c_int]
    lib.svdf_window_delta_apply_local.argtypes = [P, P]
    lib.svdf_item_block_get.argtypes = [P, P, C.POINTER(C.c_int64)]
    lib.svdf_item_block_set.argtypes = [P, P]
    lib.svdf_set_view.restype = C.c_int64
    lib.svdf_set_view.argtypes = [P, C.c_int, _f32p, C.c_int64]
    lib.svdf_dataset_from_buffer_file.restype = P
    lib.svdf_dataset_from_buffer_file.argtypes = [P, C.c_char_p, C.c_int]
    lib.svdf_dataset_from_rank_buffer_file.restype = P
    lib.svdf_dataset_from_rank_buffer_file.argtypes = [P, C.c_char_p]
    lib.svdf_rank_prefetch_buffer_file.restype = C.c_int
    lib.svdf_rank_prefetch_buffer_file.argtypes = [P, C.c_char_p]
    lib.svdf_rank_sample_buffer_file.restype = C.c_int64
    lib.svdf_rank_sample_buffer_file.argtypes = [P, C.c_char_p, C.c_char_p]
    lib.svdf_dataset_from_blocks.restype = P
    lib.svdf_dataset_from_blocks.argtypes = [P, C.c_long, _i32p, _i64p, _u32p, _f32p, _i64p, _f32p, _i64p, _u32p, _f32p]
    lib.svdf_dataset_destroy.argtypes = [P]
    lib.svdf_train_dataset.argtypes = [P, P]
    lib.svdf_predict_dataset.argtypes = [P, P, _f32p]
    lib.svdf_dataset_info.restype = C.c_int64
    lib.svdf_dataset_info.argtypes = [P, C.c_int]
    lib.svdf_item_delta_buffer.restype = P
    lib.svdf_item_delta_buffer.argtypes = [P, C.POINTER(C.c_int64)]
    lib.svdf_item_delta_export.argtypes = [P, P]
    lib.svdf_item_delta_import.argtypes = [P, P]
    lib.svdf_item_delta_into.argtypes = [P, P, C.POINTER(C.c_int64)]
    lib.svdf_item_delta_apply_from.argtypes = [P, P]
    lib.svdf_item_delta_pack.argtypes = [P, P, C.c_int, C.POINTER(C.c_int64)]
    lib.svdf_item_delta_unpack.argtypes = [P, P, C.c_int, C.c_int]
    lib.svdf_set_stream.argtypes = [P, P]
    lib.svdf_item_delta_select.argtypes = [P, C.c_int, C.c_int]
    lib.svdf_get_view.restype = C.c_int64
    lib.svdf_get_view.argtypes = [P, C.c_int, _f32p, C.c_int64]
    lib.svdf_view_shape.argtypes = [P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.svdf_stream.restype = P
    lib.svdf_stream.argtypes = [P]
    lib.svdf_counter.restype = C.c_int64
    lib.svdf_counter.argtypes = [P, C.c_int]
    lib.svdf_set_knob.argtypes = [P, C.c_char_p, C.c_long]
    lib.svdf_schedule_resources.argtypes = [C.c_long, _i64p, _u32p, C.c_long, _i32p, _i64p, C.c_long]
    lib.svdf_eval_dataset.argtypes = [P, P, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.svdf_ranker_create.restype = P
    lib.svdf_ranker_create.argtypes = [C.c_uint8] * 4 + [C.c_int]
    lib.svdf_ranker_destroy.argtypes = [P]
    lib.svdf_ranker_set_param.argtypes = [P, C.c_char_p, C.c_char_p]
    lib.svdf_ranker_load_model.argtypes = [P, C.c_void_p]
    lib.svdf_ranker_init.argtypes = [P, C.c_int]
    lib.svdf_ranker_process_csr.restype = C.c_int64
    lib.svdf_ranker_process_csr.argtypes = [P, C.c_float, C.c_int, C.c_int, C.c_int, _u32p, _f32p, _i32p, C.c_int64]
    lib.svdf_ranker_process_rows.restype = C.c_int64
    lib.svdf_ranker_process_rows.argtypes = [P, C.c_int, _f32p, _i32p, _u32p, _f32p, _i32p, C.c_int64]
    lib.svdf_ranker_process_block.restype = C.c_int64
    lib.svdf_ranker_process_block.argtypes = [P, C.c_int, C.c_int, _u32p, _f32p, C.c_int, _f32p, _i32p, _u32p, _f32p, _i32p, C.c_int64]
    lib.svdf_ranker_counter.restype = C.c_int64
    lib.svdf_ranker_counter.argtypes = [P, C.c_int]
    lib.svdf_rand_peek.argtypes = [C.c_long, _i32p]
    lib.svdf_rand_skip.argtypes = [C.c_long]
    lib.svdf_device_expf.argtypes = [C.c_void_p, C.c_uint, C.c_uint, _f32p, C.c_long]
    lib.svdf_debug_sort_labels.argtypes = [C.c_long, _f32p, _i32p, _i32p]
    lib.svdf_debug_sort_scores.argtypes = [C.c_long, _f32p, C.c_int, _i32p, _i32p]
    lib.svdf_set_error_mode(1)   # python callers get exceptions instead of exit(-1)
    _lib = lib
    return lib


def device_count():
    return load_library().svdf_device_count()


def rccl_unique_id():
    """ncclGetUniqueId (rank 0): 128 bytes for svdf_rccl_init on every rank"""
    buf = C.create_string_buffer(128)
    if load_library().svdf_rccl_unique_id(buf) != 0:
        raise SvdfError(load_library().svdf_last_error().decode())
    return buf.raw


def rand_peek(n):
    """the next n libc rand() results, without consuming them (svdf_rand_peek)"""
    out = np.zeros(max(int(n), 1), np.int32)
    if load_library().svdf_rand_peek(int(n), out) != 0:
        raise SvdfError(load_library().svdf_last_error().decode())
    return out[:n]


def rand_skip(n):
    """advance libc rand() by n draws (svdf_rand_skip)"""
    if load_library().svdf_rand_skip(int(n)) != 0:
        raise SvdfError(load_library().svdf_last_error().decode())


def debug_sort_labels(label):
    """(ids sorted by label with the restated libstdc++ std::sort of svdf_stdsort.h, the same with the C++ library's std::sort)"""
    label = np.ascontiguousarray(label, np.float32)
    a, b = np.zeros(max(len(label), 1), np.int32), np.zeros(max(len(label), 1), np.int32)
    if load_library().svdf_debug_sort_labels(len(label), _pad(label, np.float32), a, b) != 0:
        raise SvdfError(load_library().svdf_last_error().decode())
    return a[:len(label)], b[:len(label)]


def debug_sort_scores(score, threads=8):
    """(ids by descending score with the ranker's threaded restatement of std::sort, the same with the library's std::sort over the
    reference's Entry struct)"""
    score = np.ascontiguousarray(score, np.float32)
    a, b = np.zeros(max(len(score), 1), np.int32), np.zeros(max(len(score), 1), np.int32)
    if load_library().svdf_debug_sort_scores(len(score), _pad(score, np.float32), int(threads), a, b) != 0:
        raise SvdfError(load_library().svdf_last_error().decode())
    return a[:len(score)], b[:len(score)]


def device_expf(x=None, first=0, step=1, n=None):
    """The expf the sigmoid links use ON THE GPU (svdf_device.h: glibc_expf), over an array or over the floats with bit
    patterns first + j*step; tests compare it with the host libm bit for bit."""
    lib = load_library()
    if x is not None:
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(len(x), np.float32)
        rc = lib.svdf_device_expf(x.ctypes.data_as(C.c_void_p), 0, 0, out, len(x))
    else:
        out = np.empty(n, np.float32)
        rc = lib.svdf_device_expf(None, first, step, out, n)
    if rc != 0:
        raise SvdfError("svdf_device_expf failed")
    return out


def _pad(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a if a.size else np.zeros(1, dtype=dtype)


def schedule_resources(res_ptr, res, num_res):
    """Host scheduler (no GPU needed): returns (order, level_ptr)."""
    lib = load_library()
    n = len(res_ptr) - 1
    order = np.zeros(max(n, 1), np.int32)
    level_ptr = np.zeros(n + 2, np.int64)
    nl = lib.svdf_schedule_resources(n, _pad(res_ptr, np.int64), _pad(res, np.uint32), int(num_res), order, level_ptr, n + 2)
    if nl < 0:
        raise SvdfError(lib.svdf_last_error().decode())
    return order[:n], level_ptr[:nl + 1]


class Dataset:
    """A scheduled, HBM-resident training set (svdf_dataset)."""

    def __init__(self, trainer, handle):
        self.trainer, self.h = trainer, handle

    def info(self, what):
        return int(self.trainer.lib.svdf_dataset_info(self.h, what))

    num_row = property(lambda s: s.info(0))
    num_batches = property(lambda s: s.info(1))
    max_batch = property(lambda s: s.info(2))
    kind = property(lambda s: s.info(3))
    algorithmic_bytes = property(lambda s: s.info(4))
    num_units = property(lambda s: s.info(5))
    num_simple_units = property(lambda s: s.info(6))

    def close(self):
        if self.h:
            self.trainer.lib.svdf_dataset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Trainer:
    """ISVDTrainer over the HIP engine.  device=-1: current GPU (required); device=-2: host-only
    handle for config / model-file / staging logic (compute calls raise)."""

    def __init__(self, format_type=0, active_type=0, extend_type=0, variant_type=0, params=None, device=-1):
        self.lib = load_library()
        self.h = self.lib.svdf_create(format_type, active_type, extend_type, variant_type, device)
        if not self.h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        self.mtype = bytes([format_type, active_type, extend_type, variant_type])
        for k, v in (params or {}).items():
            self.set_param(k, v)

    # -- plumbing
    def _ok(self, rc):
        if rc != 0:
            raise SvdfError(self.lib.svdf_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_async_fo", None) is not None:   # an asynchronous save still open: join the writer and close its file (buffered bytes!)
                try:
                    self.save_model_end()
                except Exception:
                    pass
            self.lib.svdf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ISVDTrainer
    def set_param(self, name, val):
        self._ok(self.lib.svdf_set_param(self.h, str(name).encode(), str(val).encode()))

    def seed(self, s):
        self.lib.svdf_seed(int(s))

    def init_model(self):
        self._ok(self.lib.svdf_init_model(self.h))

    def init_trainer(self):
        self._ok(self.lib.svdf_init_trainer(self.h))

    def set_round(self, r):
        self._ok(self.lib.svdf_set_round(self.h, int(r)))

    def finish_round(self):
        self._ok(self.lib.svdf_finish_round(self.h))

    def save_model_begin(self, path, with_type_header=True):
        """svdf_save_model_begin: the model as of NOW goes to `path` on a writer thread while training continues; save_model_end() joins it and closes the file."""
        assert getattr(self, "_async_fo", None) is None, "one asynchronous save at a time"
        fo = _libc.fopen(str(path).encode(), b"wb")
        if not fo:
            raise SvdfError("can not open file \"%s\"" % path)
        if with_type_header:
            hdr = (C.c_char * 4).from_buffer_copy(self.mtype)
            _libc.fwrite(hdr, 1, 4, C.c_void_p(fo))
        self._async_fo = fo
        try:
            self._ok(self.lib.svdf_save_model_begin(self.h, fo))
        except Exception:
            _libc.fclose(fo)
            self._async_fo = None
            raise

    def save_model_end(self):
        fo = getattr(self, "_async_fo", None)
        if fo is None:
            return
        try:
            self._ok(self.lib.svdf_save_model_end(self.h))
        finally:
            _libc.fclose(fo)
            self._async_fo = None

    def save_model(self, path, with_type_header=True):
        """Caller-side protocol of svd_feature.cpp:184-191: fopen, 4-byte SVDTypeParam, save_model(fo)."""
        fo = _libc.fopen(str(path).encode(), b"wb")
        if not fo:
            raise SvdfError("can not open file \"%s\"" % path)
        try:
            if with_type_header:
                hdr = (C.c_char * 4).from_buffer_copy(self.mtype)
                _libc.fwrite(hdr, 1, 4, C.c_void_p(fo))
            self._ok(self.lib.svdf_save_model(self.h, fo))
        finally:
            _libc.fclose(fo)

    def load_model(self, path, with_type_header=True):
        fi = _libc.fopen(str(path).encode(), b"rb")
        if not fi:
            raise SvdfError("can not open file \"%s\"" % path)
        try:
            if with_type_header:
                buf = (C.c_char * 4)()
                _libc.fread(buf, 1, 4, C.c_void_p(fi))
            self._ok(self.lib.svdf_load_model(self.h, fi))
        finally:
            _libc.fclose(fi)

    def update_csr(self, label, ng, nu, ni, index, value):
        self._ok(self.lib.svdf_update_csr(self.h, float(label), ng, nu, ni, _pad(index, np.uint32), _pad(value, np.float32)))

    def predict_csr(self, label, ng, nu, ni, index, value):
        self.lib.svdf_last_error()
        return self.lib.svdf_predict_csr(self.h, float(label), ng, nu, ni, _pad(index, np.uint32), _pad(value, np.float32))

    def update_batch(self, d):
        self._ok(self.lib.svdf_update_csr_batch(self.h, d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int32),
                                                _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32)))

    def predict_batch(self, d):
        out = np.zeros(max(d.num_row, 1), dtype=np.float32)
        self._ok(self.lib.svdf_predict_csr_batch(self.h, d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int32),
                                                 _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32), out))
        return out[:d.num_row]

    def update_block(self, b):
        d = b.data
        self._ok(self.lib.svdf_update_block(self.h, b.num_ufeedback, b.extend_tag, _pad(b.index_ufeedback, np.uint32),
                                            _pad(b.value_ufeedback, np.float32), d.num_row, _pad(d.row_label, np.float32),
                                            _pad(d.row_ptr, np.int32), _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32)))

    def predict_block(self, b):
        d = b.data
        out = np.zeros(max(d.num_row, 1), dtype=np.float32)
        self._ok(self.lib.svdf_predict_block(self.h, b.num_ufeedback, b.extend_tag, _pad(b.index_ufeedback, np.uint32),
                                             _pad(b.value_ufeedback, np.float32), d.num_row, _pad(d.row_label, np.float32),
                                             _pad(d.row_ptr, np.int32), _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32), out))
        return out[:d.num_row]

    # -- resident datasets
    def dataset_from_csr(self, d):
        h = self.lib.svdf_dataset_from_csr(self.h, d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int64),
                                           _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32))
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    def dataset_from_triples(self, user, item, label):
        n = len(label)
        h = self.lib.svdf_dataset_from_triples(self.h, n, _pad(user, np.uint32), _pad(item, np.uint32), _pad(label, np.float32))
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    def dataset_window_from_triples(self, user, item, label):
        """One exchange window of a rank's shard for the window-minibatch step (svdf_dataset_window_from_triples)."""
        h = self.lib.svdf_dataset_window_from_triples(self.h, len(label), _pad(user, np.uint32), _pad(item, np.uint32), _pad(label, np.float32))
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    def dataset_window_from_pairs(self, user, pos, neg):
        """One exchange window of a rank's rank pairs for the window-minibatch step (svdf_dataset_window_from_pairs)."""
        h = self.lib.svdf_dataset_window_from_pairs(self.h, len(user), _pad(user, np.uint32), _pad(pos, np.uint32), _pad(neg, np.uint32))
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    def dataset_window_from_csr(self, d):
        """One exchange window of rows with global features / several item entries (svdf_dataset_window_from_csr)."""
        h = self.lib.svdf_dataset_window_from_csr(self.h, d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int64),
                                                  _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32))
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    def dataset_window_from_blocks(self, ba):
        """One exchange window of user-group (SVD++) blocks, a data.BlockArrays (svdf_dataset_window_from_blocks)."""
        h = self.lib.svdf_dataset_window_from_blocks(self.h, ba.num_block, _pad(ba.extend_tag, np.int32), _pad(ba.fb_ptr, np.int64),
                                                     _pad(ba.fb_index, np.uint32), _pad(ba.fb_value, np.float32), _pad(ba.block_row_ptr, np.int64),
                                                     _pad(ba.row_label, np.float32), _pad(ba.row_ptr, np.int64), _pad(ba.feat_index, np.uint32),
                                                     _pad(ba.feat_value, np.float32))
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    # -- cross-process direct exchange through IPC-mapped buffers (svdf_ipc.cpp)
    def ipc_setup(self, rank, world, wire_bytes, block_floats=0):
        """allocates + exports this rank's wire buffer / flag page; returns the 128 handle bytes to all-gather"""
        buf = C.create_string_buffer(128)
        self._ok(self.lib.svdf_ipc_setup(self.h, rank, world, int(wire_bytes), int(block_floats), buf))
        return buf.raw

    def ipc_connect(self, all_handles):
        self._ok(self.lib.svdf_ipc_connect(self.h, bytes(all_handles)))

    def ipc_window_pack(self, ds, half=False):
        self._ok(self.lib.svdf_ipc_window_pack(self.h, ds.h, 1 if half else 0))

    def ipc_window_reduce(self, half=False):
        self._ok(self.lib.svdf_ipc_window_reduce(self.h, 1 if half else 0))

    def ipc_window_apply(self, half=False):
        self._ok(self.lib.svdf_ipc_window_apply(self.h, 1 if half else 0))

    def ipc_block_send(self, dst, slot):
        self._ok(self.lib.svdf_ipc_block_send(self.h, dst, slot))

    def ipc_block_recv(self, src, slot, seq):
        self._ok(self.lib.svdf_ipc_block_recv(self.h, src, slot, seq))

    def ipc_status(self):
        return int(self.lib.svdf_ipc_status(self.h))

    def ipc_close(self):
        self._ok(self.lib.svdf_ipc_close(self.h))

    # -- the exchanges issued from C++ straight into RCCL (svdf_rccl.cpp): the rank's own communicator
    def rccl_init(self, unique_id, rank, world):
        self._ok(self.lib.svdf_rccl_init(self.h, bytes(unique_id), rank, world))

    def rccl_window_allreduce(self, ds, half=False):
        self._ok(self.lib.svdf_rccl_window_allreduce(self.h, ds.h, 1 if half else 0))

    def rccl_block_handoff(self, dst, src, slot, in_block, nblocks):
        self._ok(self.lib.svdf_rccl_block_handoff(self.h, dst, src, slot, in_block, nblocks))

    def rccl_block_arrive(self, slot):
        self._ok(self.lib.svdf_rccl_block_arrive(self.h, slot))

    def rccl_counter(self, what):
        return int(self.lib.svdf_rccl_counter(self.h, what))

    def rccl_close(self):
        self._ok(self.lib.svdf_rccl_close(self.h))

    def window_delta_pack(self, ds, device_ptr, half=False):
        """Sum of the trained window's item-side contributions into the wire buffer at device_ptr; returns its element count."""
        n = C.c_int64()
        self._ok(self.lib.svdf_window_delta_pack(self.h, ds.h, C.c_void_p(device_ptr), 1 if half else 0, C.byref(n)))
        return n.value

    def window_delta_apply(self, device_ptr, half=False):
        """replicated ranges += the (all-reduced) wire buffer"""
        self._ok(self.lib.svdf_window_delta_apply(self.h, C.c_void_p(device_ptr), 1 if half else 0))

    def window_delta_apply_local(self, ds):
        """stratified schedule: the active item block (item_delta_select) += the trained window's per-item sums, in place"""
        self._ok(self.lib.svdf_window_delta_apply_local(self.h, ds.h))

    def item_block_count(self):
        n = C.c_int64()
        self._ok(self.lib.svdf_item_block_get(self.h, None, C.byref(n)))
        return n.value

    def item_block_get(self, device_ptr):
        n = C.c_int64()
        self._ok(self.lib.svdf_item_block_get(self.h, C.c_void_p(device_ptr), C.byref(n)))
        return n.value

    def item_block_set(self, device_ptr):
        self._ok(self.lib.svdf_item_block_set(self.h, C.c_void_p(device_ptr)))

    def stratum_step(self, handles, n, block, nblocks, out_ptr):
        """handles: a ctypes array of the window data sets' handles (svdf_stratum_step)"""
        self._ok(self.lib.svdf_stratum_step(self.h, handles, n, block, nblocks, C.c_void_p(out_ptr) if out_ptr else None))

    def item_block_set_at(self, block, nblocks, device_ptr):
        self._ok(self.lib.svdf_item_block_set_at(self.h, block, nblocks, C.c_void_p(device_ptr)))

    def dataset_from_pairs(self, user, pos, neg):
        """Rank pairs (user, positive item, negative item), see svdf_dataset_from_pairs."""
        h = self.lib.svdf_dataset_from_pairs(self.h, len(user), _pad(user, np.uint32), _pad(pos, np.uint32), _pad(neg, np.uint32))
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    def dataset_from_blocks(self, blocks):
        """blocks: list of PlusBlock in file order (one pass of a user-group buffer), or a flat BlockArrays."""
        ba = blocks if isinstance(blocks, BlockArrays) else BlockArrays.from_blocks(blocks)
        h = self.lib.svdf_dataset_from_blocks(self.h, ba.num_block, _pad(ba.extend_tag, np.int32), ba.fb_ptr, _pad(ba.fb_index, np.uint32),
                                              _pad(ba.fb_value, np.float32), ba.block_row_ptr, _pad(ba.row_label, np.float32),
                                              _pad(ba.row_ptr, np.int64), _pad(ba.feat_index, np.uint32), _pad(ba.feat_value, np.float32))
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    def dataset_from_buffer_file(self, path, user_group=False):
        """Resident dataset straight from a reference buffer file (make_feature_buffer / make_ugroup_buffer output)."""
        h = self.lib.svdf_dataset_from_buffer_file(self.h, str(path).encode(), 1 if user_group else 0)
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    def dataset_from_rank_buffer_file(self, path):
        """One pass of the reference's input_type = 2 (user-group buffer file through the rank-pair sampler)."""
        h = self.lib.svdf_dataset_from_rank_buffer_file(self.h, str(path).encode())
        if not h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return Dataset(self, h)

    def rank_prefetch_buffer_file(self, path):
        """Draw the next rank-pair pass on a background thread; the next dataset_from_rank_buffer_file takes it."""
        self._ok(self.lib.svdf_rank_prefetch_buffer_file(self.h, str(path).encode()))

    def rank_sample_buffer_file(self, in_path, out_path):
        """The same pass written as a user-group buffer file (host only); returns the number of generated rows."""
        n = self.lib.svdf_rank_sample_buffer_file(self.h, str(in_path).encode(), str(out_path).encode())
        if n < 0:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return int(n)

    def train_dataset(self, ds):
        self._ok(self.lib.svdf_train_dataset(self.h, ds.h))

    def eval_dataset(self, ds, scale_score=1.0):
        """RMSEEvaluator over a resident data set (svd_feature_infer.cpp:38-56, 243-277): (sum of squared errors, count)."""
        ss, cnt = C.c_double(), C.c_int64()
        self._ok(self.lib.svdf_eval_dataset(self.h, ds.h, float(scale_score), C.byref(ss), C.byref(cnt)))
        return ss.value, cnt.value

    def predict_dataset(self, ds):
        out = np.zeros(max(ds.num_row, 1), dtype=np.float32)
        self._ok(self.lib.svdf_predict_dataset(self.h, ds.h, out))
        return out[:ds.num_row]

    # -- multi-GPU item-side delta
    def item_delta_begin(self):
        self._ok(self.lib.svdf_item_delta_begin(self.h))

    def item_delta_buffer(self):
        """(device pointer, float count) of the packed item-side delta."""
        n = C.c_int64()
        p = self.lib.svdf_item_delta_buffer(self.h, C.byref(n))
        if not p:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return p, n.value

    def item_delta_apply(self):
        self._ok(self.lib.svdf_item_delta_apply(self.h))

    def item_delta_export(self, device_ptr):
        self._ok(self.lib.svdf_item_delta_export(self.h, C.c_void_p(device_ptr)))

    def item_delta_count(self):
        """number of elements of the packed item-side delta"""
        n = C.c_int64()
        self._ok(self.lib.svdf_item_delta_pack(self.h, None, 0, C.byref(n)))
        return n.value

    def item_delta_select(self, part, nparts):
        self._ok(self.lib.svdf_item_delta_select(self.h, int(part), int(nparts)))

    def item_delta_pack(self, device_ptr, half=False):
        n = C.c_int64()
        self._ok(self.lib.svdf_item_delta_pack(self.h, C.c_void_p(device_ptr), 1 if half else 0, C.byref(n)))
        return n.value

    def item_delta_unpack(self, device_ptr, half=False, refresh_snapshot=True):
        self._ok(self.lib.svdf_item_delta_unpack(self.h, C.c_void_p(device_ptr), 1 if half else 0, 1 if refresh_snapshot else 0))

    def item_delta_into(self, device_ptr):
        n = C.c_int64()
        self._ok(self.lib.svdf_item_delta_into(self.h, C.c_void_p(device_ptr), C.byref(n)))
        return n.value

    def item_delta_apply_from(self, device_ptr):
        self._ok(self.lib.svdf_item_delta_apply_from(self.h, C.c_void_p(device_ptr)))

    def set_stream(self, hip_stream):
        self._ok(self.lib.svdf_set_stream(self.h, C.c_void_p(hip_stream)))

    def item_delta_import(self, device_ptr):
        self._ok(self.lib.svdf_item_delta_import(self.h, C.c_void_p(device_ptr)))

    # -- introspection
    def view(self, name):
        rows, cols = C.c_int(), C.c_int()
        self._ok(self.lib.svdf_view_shape(self.h, VIEW[name], C.byref(rows), C.byref(cols)))
        if rows.value < 0:
            return None
        out = np.zeros(max(rows.value * cols.value, 1), dtype=np.float32)
        n = self.lib.svdf_get_view(self.h, VIEW[name], out, out.size)
        if n < 0:
            raise SvdfError(self.lib.svdf_last_error().decode())
        out = out[:n]
        return out.reshape(rows.value, cols.value) if name.startswith("W_") else out

    def set_view(self, name, values):
        v = np.ascontiguousarray(values, np.float32).ravel()
        if self.lib.svdf_set_view(self.h, VIEW[name], _pad(v, np.float32), v.size) < 0:
            raise SvdfError(self.lib.svdf_last_error().decode() or "set_view: shape mismatch")

    def synchronize(self):
        self._ok(self.lib.svdf_synchronize(self.h))

    def stream(self):
        return self.lib.svdf_stream(self.h)

    def counter(self, what):
        return int(self.lib.svdf_counter(self.h, what))

    def set_knob(self, name, value):
        self._ok(self.lib.svdf_set_knob(self.h, name.encode(), int(value)))


_libc.fwrite.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
_libc.fread.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]


class Ranker:
    """ISVDRanker over the HIP engine (apex_svd.h:160-197; SVDFeatureRanker apex_svd_base.h:597-813)."""

    def __init__(self, format_type=0, active_type=0, extend_type=0, variant_type=0, device=-1):
        self.lib = load_library()
        self.h = self.lib.svdf_ranker_create(format_type, active_type, extend_type, variant_type, device)
        if not self.h:
            raise SvdfError(self.lib.svdf_last_error().decode())
        self.cap = 16
        self.top_k = 0

    def close(self):
        if self.h:
            self.lib.svdf_ranker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ok(self, rc):
        if rc < 0:
            raise SvdfError(self.lib.svdf_last_error().decode())
        return rc

    def set_param(self, name, val):
        self._ok(self.lib.svdf_ranker_set_param(self.h, str(name).encode(), str(val).encode()))
        if str(name) == "top_k":
            self.top_k = int(val)

    def load_model(self, path, with_type_header=True):
        fi = _libc.fopen(str(path).encode(), b"rb")
        if not fi:
            raise SvdfError("cannot open %s" % path)
        try:
            if with_type_header:
                buf = C.create_string_buffer(4)
                _libc.fread(buf, 1, 4, C.c_void_p(fi))
            self._ok(self.lib.svdf_ranker_load_model(self.h, fi))
        finally:
            _libc.fclose(fi)

    def init_ranker(self, num_item_set):
        self.cap = max(16, int(num_item_set) + 16)
        self._ok(self.lib.svdf_ranker_init(self.h, int(num_item_set)))

    @staticmethod
    def _taken(out, n, cap):
        """the native calls return the number of results of the line(s) and write at most `cap` of them: more than the buffer
        holds is an error here, never a silent truncation"""
        if n > cap:
            raise SvdfError("ranker: the call produced %d results, the result buffer holds %d (several PROCESS lines in one call?)" % (n, cap))
        return out[:n].copy()

    def process(self, label, ng, nu, ni, index, value):
        out = np.zeros(self.cap, np.int32)
        n = self._ok(self.lib.svdf_ranker_process_csr(self.h, float(label), ng, nu, ni, _pad(index, np.uint32), _pad(value, np.float32), out, self.cap))
        return self._taken(out, n, self.cap)

    def process_rows(self, d):
        """every row of a CSRData through process() in ONE native call (svdf_ranker_process_rows: sections pipelined on the
        device); returns the concatenated results"""
        lab = np.asarray(d.row_label)
        nproc = int(np.count_nonzero(lab == 4.0))
        rp = np.asarray(d.row_ptr, np.int64)
        npos = int(((rp[2::3] - rp[1:-1:3])[lab == 1.0]).sum()) if d.num_row else 0   # ids listed by the POS_SAMPLE lines
        per = self.top_k if self.top_k > 0 else self.counter(2) + npos
        cap = max(16, nproc * per + 16)
        out = np.zeros(cap, np.int32)
        n = self._ok(self.lib.svdf_ranker_process_rows(self.h, d.num_row, _pad(d.row_label, np.float32), _pad(d.row_ptr, np.int32),
                                                       _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32), out, cap))
        return self._taken(out, n, cap)

    def process_block(self, b):
        d = b.data
        cap = self.cap * max(1, d.num_row)
        out = np.zeros(cap, np.int32)
        n = self._ok(self.lib.svdf_ranker_process_block(self.h, b.num_ufeedback, b.extend_tag, _pad(b.index_ufeedback, np.uint32),
                                                        _pad(b.value_ufeedback, np.float32), d.num_row, _pad(d.row_label, np.float32),
                                                        _pad(d.row_ptr, np.int32), _pad(d.feat_index, np.uint32), _pad(d.feat_value, np.float32),
                                                        out, cap))
        return self._taken(out, n, cap)

    def counter(self, what):
        return int(self.lib.svdf_ranker_counter(self.h, what))
