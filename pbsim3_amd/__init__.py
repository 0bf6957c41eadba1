"""pbsim3_amd -- Python (ctypes) binding of the C ABI in include/pbsim3_amd.h.

The product is the HIP library pbsim3_amd/lib/libpbsim3_amd.so; this module
only loads it and mirrors its entry points (same names, same SUCCEEDED(1) /
FAILED(0) convention as the reference, pbsim.cpp:16-17).  There is no Python
or CPU implementation of the path behind it: if the library is missing, or no
gfx950 device is usable, the calls raise.
"""
import ctypes as C
import os

from . import build as _build

STRATEGY_WGS, STRATEGY_TRANS, STRATEGY_TEMPL = 1, 2, 3
METHOD_QS, METHOD_ERR, METHOD_SAMPLE = 1, 2, 3


class Params(C.Structure):
    _fields_ = [
        ("strategy", C.c_int32), ("method", C.c_int32), ("seed", C.c_uint32), ("pass_num", C.c_int32),
        ("depth", C.c_double), ("accuracy_mean", C.c_double), ("len_mean", C.c_double), ("len_sd", C.c_double),
        ("hp_del_bias", C.c_double), ("len_min", C.c_int64), ("len_max", C.c_int64),
        ("sub_ratio", C.c_int64), ("ins_ratio", C.c_int64), ("del_ratio", C.c_int64),
        ("id_prefix", C.c_char * 64),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("res_num", C.c_int64), ("res_pass_num", C.c_int64), ("res_len_total", C.c_int64),
        ("res_len_min", C.c_int64), ("res_len_max", C.c_int64),
        ("res_sub_num", C.c_int64), ("res_ins_num", C.c_int64), ("res_del_num", C.c_int64),
        ("res_depth", C.c_double), ("res_len_mean", C.c_double), ("res_len_sd", C.c_double),
        ("res_accuracy_mean", C.c_double), ("res_accuracy_sd", C.c_double),
        ("res_sub_rate", C.c_double), ("res_ins_rate", C.c_double), ("res_del_rate", C.c_double),
    ]


class BatchInfo(C.Structure):
    _fields_ = [
        ("first_read", C.c_int64), ("n_reads", C.c_int64), ("n_final", C.c_int64),
        ("quota_reached", C.c_int32), ("need_truncated_read", C.c_int32),
        ("len_total_after", C.c_int64), ("bases", C.c_int64),
        ("read_text_bytes", C.c_int64), ("maf_text_bytes", C.c_int64),
        ("ref_bases", C.c_int64), ("maf_columns", C.c_int64),
    ]


SINK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_char), C.c_int64)


class Sink(C.Structure):
    _fields_ = [("user", C.c_void_p), ("on_read_text", SINK_CB), ("on_maf_text", SINK_CB)]


OP_SUM, OP_MIN, OP_MAX = 0, 1, 2
GATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64))
REDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.c_int32)
BCAST_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)
ABORT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)


class Comm(C.Structure):
    """pbsim_comm: blocking collectives over the ranks of a job (one context per GPU)."""
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32),
                ("all_gather_i64", GATHER_CB), ("all_reduce_i64", REDUCE_CB), ("broadcast", BCAST_CB),
                ("abort", ABORT_CB)]


REC_TEXT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_char), C.c_int64, C.c_int64)
REC_DONE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(Stats), C.c_int64, C.c_int64)


class RecordSink(C.Structure):
    _fields_ = [("user", C.c_void_p), ("on_read_text", REC_TEXT_CB), ("on_maf_text", REC_TEXT_CB),
                ("on_record_done", REC_DONE_CB)]


# every symbol include/pbsim3_amd.h declares: (name, restype, argtypes)
API = [
    ("pbsim_job_add_record", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("pbsim_job_add_record_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("pbsim_job_add_record_lines", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    ("pbsim_job_add_record_comm", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(Comm), C.c_int32]),
    ("pbsim_job_records", C.c_int64, [C.c_void_p]),
    ("pbsim_job_begin", C.c_int, [C.c_void_p, C.c_int64]),
    ("pbsim_job_clear", C.c_int, [C.c_void_p]),
    ("pbsim_job_run", C.c_int, [C.c_void_p, C.POINTER(Comm), C.POINTER(RecordSink)]),
    ("pbsim_job_sam_header", C.c_int64, [C.c_void_p, C.c_int64, C.c_char_p, C.c_int64]),
    ("pbsim_job_bam_header", C.c_int64, [C.c_void_p, C.c_int64, C.c_char_p, C.c_int64]),
    ("pbsim_job_counters", C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    ("pbsim_job_progress", C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    ("pbsim_job_set_interleave", C.c_int, [C.c_void_p, C.c_int]),
    ("pbsim_job_expect", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    ("pbsim_job_feed_abort", C.c_int, [C.c_void_p, C.c_char_p]),
    ("pbsim_scratch_state", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("pbsim_batch_fetch_lengths", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pbsim_job_breakdown", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("pbsim_bind_host_to_device", C.c_int, [C.c_int, C.c_char_p, C.c_int64]),
    ("pbsim_rccl_unique_id", C.c_int64, [C.c_void_p, C.c_int64]),
    ("pbsim_rccl_comm_create", C.POINTER(Comm), [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    ("pbsim_rccl_comm_create_file", C.POINTER(Comm), [C.c_char_p, C.c_int32, C.c_int32, C.c_int32]),
    ("pbsim_rccl_comm_info", C.c_int, [C.POINTER(Comm), C.POINTER(C.c_int64)]),
    ("pbsim_rccl_comm_destroy", None, [C.POINTER(Comm)]),
    ("pbsim_stats_keep_values", C.c_int, [C.c_void_p, C.c_int]),
    ("pbsim_stats_merge", C.c_int, [C.c_void_p, C.POINTER(Comm)]),
    ("pbsim_stats_add_tasks", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    ("pbsim_format_stats", C.c_int64, [C.POINTER(Params), C.POINTER(Stats), C.c_int64, C.c_char_p, C.c_int64]),
    ("pbsim_cli_main", C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(Comm), C.c_int]),
    ("pbsim_params_default", None, [C.POINTER(Params)]),
    ("pbsim_create", C.c_void_p, [C.POINTER(Params), C.c_int]),
    ("pbsim_destroy", None, [C.c_void_p]),
    ("pbsim_last_error", C.c_char_p, []),
    ("pbsim_version", C.c_char_p, []),
    ("pbsim_load_errhmm", C.c_int, [C.c_void_p, C.c_char_p]),
    ("pbsim_load_qshmm", C.c_int, [C.c_void_p, C.c_char_p]),
    ("pbsim_set_reference", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    ("pbsim_set_reference_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    ("pbsim_add_hp_census", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("pbsim_finish_hp_census", C.c_int, [C.c_void_p]),
    ("pbsim_set_transcripts", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    ("pbsim_set_templates", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_int64)]),
    ("pbsim_simulate_templ", C.c_int, [C.c_void_p, C.POINTER(Sink)]),
    ("pbsim_prefetch_reference", C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    ("pbsim_prefetch_reference_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("pbsim_simulate_units_range", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(Sink)]),
    ("pbsim_unit_reads", C.c_int64, [C.c_void_p]),
    ("pbsim_load_transcript_file", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    ("pbsim_load_template_file", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    ("pbsim_simulate_wgs", C.c_int, [C.c_void_p, C.POINTER(Sink)]),
    ("pbsim_simulate_trans", C.c_int, [C.c_void_p, C.POINTER(Sink)]),
    ("pbsim_get_stats", C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    ("pbsim_sam_header", C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    ("pbsim_set_bam_output", C.c_int, [C.c_void_p, C.c_int]),
    ("pbsim_bam_header", C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    ("pbsim_set_sample_profile", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    ("pbsim_simulate_sample", C.c_int, [C.c_void_p, C.POINTER(Sink)]),
    ("pbsim_simulate_sample_comm", C.c_int, [C.c_void_p, C.POINTER(Comm), C.POINTER(RecordSink)]),
    ("pbsim_set_deflate", C.c_int, [C.c_void_p, C.c_int]),
    ("pbsim_deflate_bound", C.c_int64, [C.c_int64]),
    ("pbsim_batch_fetch_deflated", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                             C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("pbsim_deflate_buffer", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    ("pbsim_batch_walk", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    ("pbsim_slot_count", C.c_int, []),
    ("pbsim_select_slot", C.c_int, [C.c_void_p, C.c_int]),
    ("pbsim_batch_walk_begin", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    ("pbsim_batch_walk_end", C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    ("pbsim_batch_finalize", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(BatchInfo)]),
    ("pbsim_batch_fetch", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pbsim_batch_account", C.c_int, [C.c_void_p]),
    ("pbsim_reset_stats", C.c_int, [C.c_void_p]),
    ("pbsim_unit_quota", C.c_int64, [C.c_void_p]),
    ("pbsim_batch_capacity", C.c_int64, [C.c_void_p]),
    ("pbsim_set_scratch_bytes", C.c_int, [C.c_void_p, C.c_int64]),
    ("pbsim_release_pools", C.c_int, [C.c_void_p]),
    ("pbsim_prof_reset", C.c_int, [C.c_void_p]),
    ("pbsim_prof_get", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    ("pbsim_prof_walk_busy", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("pbsim_prof_tail", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("pbsim_prof_secondary", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("pbsim_prof_wave_launches", C.c_int64, [C.c_void_p]),
    ("pbsim_stream", C.c_void_p, [C.c_void_p]),
    ("pbsim_device_synchronize", C.c_int, [C.c_void_p]),
    ("pbsim_philox4x32_10", None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("pbsim_dump_table", C.c_int64, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
]

_lib = None


def lib_path():
    return _build.LIB


def _torch_runtime_first():
    """PyTorch-ROCm wheels bundle their own libhsa-runtime64 / libamdhip64; this library links the system copies.  The two
    coexist in one process only when torch's are mapped first, so if torch is installed but not imported yet, map its two
    runtime libraries now (no `import torch`): a later `import torch` in the same process then still finds its device.
    PBSIM_TORCH_COMPAT=0 skips this."""
    import sys
    if "torch" in sys.modules or os.environ.get("PBSIM_TORCH_COMPAT", "1") == "0":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if not spec or not spec.origin:
            return
        libdir = os.path.join(os.path.dirname(spec.origin), "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:       # best effort: the product itself does not depend on it
        pass


def load(build_if_missing=True):
    """Loads libpbsim3_amd.so (building it in-tree with hipcc when absent)."""
    global _lib
    if _lib is not None:
        return _lib
    _torch_runtime_first()
    if not os.path.exists(_build.LIB):
        if not build_if_missing:
            raise RuntimeError("libpbsim3_amd.so is not built (python -m pbsim3_amd.build)")
        _build.build()
    lib = C.CDLL(_build.LIB)
    for name, res, args in API:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class PbsimError(RuntimeError):
    pass


def _check(ok):
    if not ok:
        raise PbsimError(load().pbsim_last_error().decode(errors="replace"))


def default_params(**kw):
    p = Params()
    load().pbsim_params_default(C.byref(p))
    for k, v in kw.items():
        if k == "id_prefix":
            v = v.encode() if isinstance(v, str) else v
        setattr(p, k, v)
    return p


# empty BGZF block: the end-of-file marker of a BAM file (SAMv1 4.1.2), also a valid empty gzip member
BGZF_EOF = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def make_comm(rank, world, all_gather, all_reduce, broadcast=None, abort=None):
    """pbsim_comm from Python callables (numpy int64 arrays in and out; the statistics merge moves millions of values):
         all_gather(array[n]) -> array[world, n] (rank-major), all_reduce(array[n], op) -> array[n],
         broadcast(ptr:int, nbytes:int, root:int, on_device:bool) -> None (optional, C1).
    The returned Comm keeps the ctypes trampolines alive (comm._keep)."""
    def _g(user, send, n, recv):
        try:
            import numpy as np
            if n == 0:
                return 1
            a = np.ctypeslib.as_array(send, shape=(n,))
            out = np.ctypeslib.as_array(recv, shape=(world * n,))
            out[:] = np.asarray(all_gather(a), dtype=np.int64).reshape(-1)   # rank-major
            return 1
        except Exception:
            import traceback
            traceback.print_exc()
            return 0

    def _r(user, buf, n, op):
        try:
            import numpy as np
            a = np.ctypeslib.as_array(buf, shape=(n,)) if n else None
            if n:
                a[:] = all_reduce(a, op)
            return 1
        except Exception:
            import traceback
            traceback.print_exc()
            return 0

    def _b(user, ptr, nbytes, root, on_device):
        try:
            broadcast(ptr, nbytes, root, bool(on_device))
            return 1
        except Exception:
            import traceback
            traceback.print_exc()
            return 0

    def _a(user):
        try:
            abort()
            return 1
        except Exception:
            import traceback
            traceback.print_exc()
            return 0

    cbs = (GATHER_CB(_g), REDUCE_CB(_r), BCAST_CB(_b) if broadcast else BCAST_CB(), ABORT_CB(_a) if abort else ABORT_CB())
    comm = Comm(None, rank, world, *cbs)
    comm._keep = cbs
    return comm


def torch_comm(dist, device):
    """pbsim_comm over torch.distributed (backend nccl = RCCL on the GPUs, gloo on the CPU): the job's integer collectives
    (C3 all_gather, C2 all_reduce).  `device`: the torch device the collectives run on (this rank's GPU for nccl, cpu for
    gloo).  No broadcast callback: a caller that holds the records as torch tensors broadcasts them itself (C1,
    dist.broadcast) and hands every rank its copy (pbsim_job_add_record_device); pbsim_cli_main has every rank read the
    <prefix>_NNNN.ref files instead."""
    import numpy as np
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()

    def all_gather(arr):
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.stack(out).cpu().numpy()

    def all_reduce(arr, op):
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
        dist.all_reduce(t, op={OP_SUM: dist.ReduceOp.SUM, OP_MIN: dist.ReduceOp.MIN, OP_MAX: dist.ReduceOp.MAX}[op])
        return t.cpu().numpy()

    def abort():
        # one process per rank (torchrun): the other ranks wait in a collective this rank will never enter.  Ending this
        # process is what tears the group down -- the launcher then ends the others (include/pbsim3_amd.h, pbsim_comm.abort).
        import sys
        sys.stderr.write("pbsim3_amd: rank %d failed inside the job (%s): leaving the process group\n"
                         % (rank, load().pbsim_last_error().decode(errors="replace")))
        sys.stderr.flush()
        os._exit(1)

    return make_comm(rank, world, all_gather, all_reduce, None, abort)


RCCL_ID_BYTES = 128


class RcclComm:
    """pbsim_comm over RCCL for ONE PROCESS PER GPU (pbsim_rccl_comm_create = ncclCommInitRank): the communicator bench.py
    --gpus N and run_multi use by default.  `exchange(id_or_None) -> id`: the launcher's side channel -- rank 0 passes the id it
    made and every rank gets it back (torch's store: RcclComm.from_torch; a rendezvous file: RcclComm.from_file).
    .ref is what pbsim_job_run / pbsim_cli_main take (C.byref of the library's own struct); .comm the struct itself."""

    def __init__(self, ptr):
        if not ptr:
            raise PbsimError(load().pbsim_last_error().decode(errors="replace"))
        self.ptr = ptr
        self.comm = ptr.contents
        self.ref = ptr

    @classmethod
    def create(cls, rank, world, device, exchange):
        lib = load()
        ident = None
        if rank == 0:
            buf = C.create_string_buffer(RCCL_ID_BYTES)
            if lib.pbsim_rccl_unique_id(buf, RCCL_ID_BYTES) != RCCL_ID_BYTES:
                # the others wait in `exchange`: tell them (an empty id) before raising
                exchange(b"")
                raise PbsimError(lib.pbsim_last_error().decode(errors="replace"))
            ident = buf.raw
        ident = exchange(ident)
        if len(ident) != RCCL_ID_BYTES:
            raise PbsimError("rank 0 could not make an RCCL id")
        return cls(lib.pbsim_rccl_comm_create(ident, len(ident), rank, world, device))

    @classmethod
    def from_torch(cls, dist, device, key="pbsim_rccl_id"):
        """the id through torch.distributed's key-value store (TCP, the rendezvous torchrun already made)"""
        return cls.create(dist.get_rank(), dist.get_world_size(), device, store_exchange(dist, key))

    @classmethod
    def from_file(cls, path, rank, world, device):
        return cls(load().pbsim_rccl_comm_create_file(os.fsencode(path), rank, world, device))

    def info(self):
        out = (C.c_int64 * 4)()
        _check(load().pbsim_rccl_comm_info(self.ptr, out))
        return {"ranks_seen": out[0], "rank": out[1], "device": out[2], "collectives": out[3]}

    def close(self):
        if self.ptr:
            load().pbsim_rccl_comm_destroy(self.ptr)
            self.ptr = self.ref = None


_store_round = [0]


def store_exchange(dist, key="pbsim_rccl_id"):
    """exchange(id_or_None) -> id over torch.distributed's key-value store: rank 0 sets the bytes under a key of this call's
    own (a process that makes several communicators gets several keys -- every rank counts the calls alike), the others block
    in get() until it is there.  Device-free: works under any backend (tests/test_multi_gloo.py)."""
    rank = dist.get_rank()
    store = dist.distributed_c10d._get_default_store()
    _store_round[0] += 1
    k = "%s/%d" % (key, _store_round[0])

    def exchange(ident):
        if rank == 0:
            store.set(k, ident)
            return ident
        return bytes(store.get(k))
    return exchange


def comm_latency(comm_ref, n_words=8, iters=1000, warm=50):
    """microseconds per all_gather_i64 / all_reduce_i64 of `n_words` through the function pointers of a pbsim_comm exactly as
    job.cpp calls them (blocking, one after the other); comm_ref: C.byref(Comm) or a POINTER(Comm).  Collective: every rank
    of the communicator calls it with the same arguments."""
    import time
    cm = comm_ref.contents if hasattr(comm_ref, "contents") else comm_ref._obj
    send = (C.c_int64 * n_words)(*range(n_words))
    recv = (C.c_int64 * (n_words * cm.world))()
    res = {"world": cm.world, "words": n_words, "iters": iters}
    for name, call in (("all_gather_us", lambda: cm.all_gather_i64(cm.user, send, n_words, recv)),
                       ("all_reduce_us", lambda: cm.all_reduce_i64(cm.user, send, n_words, OP_MAX))):
        for _ in range(warm):
            if not call():
                raise PbsimError("comm_latency: the collective failed")
        t0 = time.perf_counter()
        for _ in range(iters):
            call()
        res[name] = (time.perf_counter() - t0) / iters * 1e6
    return res


def bind_host_to_device(device):
    """Binds the calling thread (and the threads and pinned allocations it creates from now on) to the CPUs and the memory of
    the NUMA node `device`'s PCIe slot hangs off; call before the first HIP call of the process.  Returns a description of what
    was done ("" when there is nothing to bind to).  PBSIM_NUMA_BIND=0 turns it off."""
    buf = C.create_string_buffer(512)
    load().pbsim_bind_host_to_device(device, buf, 512)
    return buf.value.decode(errors="replace")


def cli_main(argv, comm=None, device=-1):
    """pbsim_cli_main: the whole command line for this rank (argv without the program name)."""
    args = [b"pbsim"] + [os.fsencode(a) for a in argv]
    arr = (C.c_char_p * (len(args) + 1))(*args, None)
    ref = None if comm is None else (C.byref(comm) if isinstance(comm, Comm) else comm)   # a Comm, or a POINTER(Comm) (RcclComm.ref)
    return load().pbsim_cli_main(len(args), arr, ref, device)


class Context:
    """One GPU context (pbsim_ctx).  device=-1 gives a tables-only context that
    can build and dump host tables but refuses every compute call."""

    def __init__(self, params, device=0):
        self.lib = load()
        self.params = params
        self.h = self.lib.pbsim_create(C.byref(params), device)
        if not self.h:
            raise PbsimError(self.lib.pbsim_last_error().decode(errors="replace"))

    def close(self):
        if self.h:
            self.lib.pbsim_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def load_errhmm(self, path):
        _check(self.lib.pbsim_load_errhmm(self.h, os.fsencode(path)))

    def load_qshmm(self, path):
        _check(self.lib.pbsim_load_qshmm(self.h, os.fsencode(path)))

    def set_reference(self, seq: bytes, record_index: int):
        buf = C.create_string_buffer(seq, len(seq))
        _check(self.lib.pbsim_set_reference(self.h, C.cast(buf, C.c_void_p), len(seq), record_index))

    def set_reference_device(self, ptr: int, length: int, record_index: int):
        _check(self.lib.pbsim_set_reference_device(self.h, C.c_void_p(ptr), length, record_index))

    def prefetch_reference_device(self, ptr: int, length: int):
        """upload + prepare the NEXT record beside the current simulation; set_reference_device(ptr, length, ...) adopts it"""
        _check(self.lib.pbsim_prefetch_reference_device(self.h, C.c_void_p(ptr), length))

    def add_hp_census(self, seq: bytes):
        buf = C.create_string_buffer(seq, len(seq))
        _check(self.lib.pbsim_add_hp_census(self.h, C.cast(buf, C.c_void_p), len(seq)))

    def finish_hp_census(self):
        _check(self.lib.pbsim_finish_hp_census(self.h))

    def simulate_wgs(self, collect=True):
        """Runs the quota loop of the current record; returns (read_text, maf_text)."""
        reads, mafs = [], []

        def on_read(user, text, n):
            reads.append(C.string_at(text, n))
            return 1

        def on_maf(user, text, n):
            mafs.append(C.string_at(text, n))
            return 1

        sink = Sink(None, SINK_CB(on_read), SINK_CB(on_maf))
        _check(self.lib.pbsim_simulate_wgs(self.h, C.byref(sink) if collect else None))
        return b"".join(reads), b"".join(mafs)

    def sam_header(self):
        n = self.lib.pbsim_sam_header(self.h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.lib.pbsim_sam_header(self.h, buf, n + 1)
        return buf.raw[:n]

    def set_sample_profile(self, quals):
        """quals: list[bytes], the filtered quality strings in file order (pbsim3_amd.args.read_sample_fastq)"""
        n = len(quals)
        keep = [C.create_string_buffer(q, len(q)) for q in quals]
        ptrs = (C.c_void_p * n)(*[C.cast(b, C.c_void_p).value for b in keep])
        lens = (C.c_int64 * n)(*[len(q) for q in quals])
        _check(self.lib.pbsim_set_sample_profile(self.h, n, ptrs, lens))

    def simulate_sample(self, collect=True):
        return self._simulate(self.lib.pbsim_simulate_sample, collect)

    def _simulate(self, fn, collect):
        reads, mafs = [], []

        def on_read(user, text, n):
            reads.append(C.string_at(text, n))
            return 1

        def on_maf(user, text, n):
            mafs.append(C.string_at(text, n))
            return 1

        sink = Sink(None, SINK_CB(on_read), SINK_CB(on_maf))
        _check(fn(self.h, C.byref(sink) if collect else None))
        return b"".join(reads), b"".join(mafs)

    def set_deflate(self, mask=3):
        """bit 0: read sink, bit 1: MAF sink receive gzip members compressed on the GPU"""
        _check(self.lib.pbsim_set_deflate(self.h, 3 if mask is True else int(mask)))

    def deflate_buffer(self, data: bytes) -> bytes:
        """gzip members (BGZF-framed) of `data`, compressed by the GPU kernels."""
        cap = self.lib.pbsim_deflate_bound(len(data)) + 64
        dst = C.create_string_buffer(cap)
        n = C.c_int64(0)
        src = C.create_string_buffer(data, len(data)) if data else None
        _check(self.lib.pbsim_deflate_buffer(self.h, src, len(data), dst, cap, C.byref(n)))
        return dst.raw[:n.value]

    def set_transcripts(self, ids, plus, minus, seqs):
        """ids: list[str]; plus/minus: expression counts; seqs: list[bytes]."""
        n = len(ids)
        c_ids = (C.c_char_p * n)(*[i.encode() for i in ids])
        c_plus = (C.c_int64 * n)(*plus)
        c_minus = (C.c_int64 * n)(*minus)
        self._keep = [C.create_string_buffer(s, len(s)) for s in seqs]
        c_seqs = (C.c_void_p * n)(*[C.cast(b, C.c_void_p).value for b in self._keep])
        c_lens = (C.c_int64 * n)(*[len(s) for s in seqs])
        _check(self.lib.pbsim_set_transcripts(self.h, n, c_ids, c_plus, c_minus, c_seqs, c_lens))
        self._keep = None

    def load_transcript_file(self, path):
        """--transcript file -> units; returns (transcripts, total expression value)"""
        st = (C.c_int64 * 2)()
        _check(self.lib.pbsim_load_transcript_file(self.h, os.fsencode(path), st))
        return st[0], st[1]

    def load_template_file(self, path):
        """--template FASTA -> units; returns (templates, total length)"""
        st = (C.c_int64 * 2)()
        _check(self.lib.pbsim_load_template_file(self.h, os.fsencode(path), st))
        return st[0], st[1]

    def unit_reads(self):
        return self.lib.pbsim_unit_reads(self.h)

    def simulate_units_range(self, first_read, n_reads):
        """reads first_read .. first_read + n_reads - 1 of the unit set (one rank's shard); returns (read text, maf text)"""
        reads, mafs = [], []

        def on_read(user, text, n):
            reads.append(C.string_at(text, n))
            return 1

        def on_maf(user, text, n):
            mafs.append(C.string_at(text, n))
            return 1

        sink = Sink(None, SINK_CB(on_read), SINK_CB(on_maf))
        _check(self.lib.pbsim_simulate_units_range(self.h, first_read, n_reads, C.byref(sink)))
        return b"".join(reads), b"".join(mafs)

    def simulate_trans(self, collect=True):
        reads, mafs = [], []

        def on_read(user, text, n):
            reads.append(C.string_at(text, n))
            return 1

        def on_maf(user, text, n):
            mafs.append(C.string_at(text, n))
            return 1

        sink = Sink(None, SINK_CB(on_read), SINK_CB(on_maf))
        _check(self.lib.pbsim_simulate_trans(self.h, C.byref(sink) if collect else None))
        return b"".join(reads), b"".join(mafs)

    def stats(self):
        s = Stats()
        _check(self.lib.pbsim_get_stats(self.h, C.byref(s)))
        return s

    def batch_walk(self, first_read, n_reads, truncate_remaining=-1):
        out = C.c_int64(0)
        _check(self.lib.pbsim_batch_walk(self.h, first_read, n_reads, truncate_remaining, C.byref(out)))
        return out.value

    def select_slot(self, slot):
        _check(self.lib.pbsim_select_slot(self.h, slot))

    def batch_walk_begin(self, first_read, n_reads, truncate_remaining=-1):
        _check(self.lib.pbsim_batch_walk_begin(self.h, first_read, n_reads, truncate_remaining))

    def batch_walk_end(self):
        out = C.c_int64(0)
        _check(self.lib.pbsim_batch_walk_end(self.h, C.byref(out)))
        return out.value

    def batch_finalize(self, len_total_before):
        bi = BatchInfo()
        _check(self.lib.pbsim_batch_finalize(self.h, len_total_before, C.byref(bi)))
        return bi

    def batch_fetch(self, info):
        r = C.create_string_buffer(max(1, info.read_text_bytes))
        m = C.create_string_buffer(max(1, info.maf_text_bytes))
        _check(self.lib.pbsim_batch_fetch(self.h, C.cast(r, C.c_void_p), C.cast(m, C.c_void_p)))
        return r.raw[:info.read_text_bytes], m.raw[:info.maf_text_bytes]

    def batch_fetch_deflated(self, info):
        """(read members, maf members): the batch's text as BGZF-framed gzip members compressed on the GPU"""
        cr = self.lib.pbsim_deflate_bound(info.read_text_bytes) + 16
        cm = self.lib.pbsim_deflate_bound(info.maf_text_bytes) + 16
        r, m = C.create_string_buffer(cr), C.create_string_buffer(cm)
        nr, nm = C.c_int64(0), C.c_int64(0)
        _check(self.lib.pbsim_batch_fetch_deflated(self.h, C.cast(r, C.c_void_p), cr, C.cast(m, C.c_void_p), cm,
                                                   C.byref(nr), C.byref(nm)))
        return r.raw[:nr.value], m.raw[:nm.value]

    def set_bam_output(self, on=True):
        _check(self.lib.pbsim_set_bam_output(self.h, 1 if on else 0))

    def bam_header(self):
        n = self.lib.pbsim_bam_header(self.h, None, 0)
        buf = C.create_string_buffer(n)
        self.lib.pbsim_bam_header(self.h, buf, n)
        return buf.raw[:n]

    # ---- the whole job (pbsim_job_*)
    def job_begin(self, first_record=1):
        _check(self.lib.pbsim_job_begin(self.h, first_record))

    def job_add_record(self, seq: bytes):
        buf = C.create_string_buffer(seq, len(seq))
        _check(self.lib.pbsim_job_add_record(self.h, C.cast(buf, C.c_void_p), len(seq)))

    def job_add_record_lines(self, lines: bytes):
        """the record as its FASTA sequence lines (line feeds included); they are squeezed out on the GPU"""
        buf = C.create_string_buffer(lines, len(lines))
        _check(self.lib.pbsim_job_add_record_lines(self.h, C.cast(buf, C.c_void_p), len(lines), len(lines) - lines.count(b"\n")))

    def job_expect(self, lens):
        """announce the job's records (their lengths): job_run may start before they have all been added, another thread adds them"""
        a = (C.c_int64 * len(lens))(*lens)
        _check(self.lib.pbsim_job_expect(self.h, len(lens), a))

    def job_feed_abort(self, why="the feeding thread gave up"):
        _check(self.lib.pbsim_job_feed_abort(self.h, why.encode()))

    def job_add_record_device(self, ptr: int, length: int):
        _check(self.lib.pbsim_job_add_record_device(self.h, C.c_void_p(ptr), length))

    def job_run(self, comm=None, collect=True, on_done=None):
        """Runs every record of the job.  collect: returns {record: [read bytes, maf bytes]} assembled from the positional
        pieces this rank received (holes stay zero: other ranks' ranges), plus {record: (Stats, read_bytes, maf_bytes)}."""
        pieces, done = {}, {}

        def put(which, rec, text, n, off):
            pieces.setdefault(rec, [[], []])[which].append((off, C.string_at(text, n)))
            return 1

        def fin(user, rec, st, rb, mb):
            s = Stats()
            C.memmove(C.byref(s), st, C.sizeof(Stats))
            done[rec] = (s, rb, mb)
            if on_done:
                on_done(rec, s, rb, mb)
            return 1

        cbs = (REC_TEXT_CB(lambda u, r, t, n, o: put(0, r, t, n, o)), REC_TEXT_CB(lambda u, r, t, n, o: put(1, r, t, n, o)),
               REC_DONE_CB(fin))
        sink = RecordSink(None, *cbs) if collect else RecordSink(None, REC_TEXT_CB(), REC_TEXT_CB(), cbs[2])
        _check(self.lib.pbsim_job_run(self.h, C.byref(comm) if comm is not None else None, C.byref(sink)))
        out = {}
        for rec, (s, rb, mb) in done.items():
            bufs = [bytearray(rb), bytearray(mb)]
            for which in (0, 1):
                for off, data in pieces.get(rec, [[], []])[which]:
                    bufs[which][off:off + len(data)] = data
            out[rec] = bufs
        return out, done

    def job_progress(self):
        """(phase, record, first_read, n_per, world, len_total, quota, next_read) of the exchange pbsim_job_run is about to enter"""
        a = (C.c_int64 * 8)()
        _check(self.lib.pbsim_job_progress(self.h, a))
        return tuple(a)

    def batch_fetch_lengths(self, n_reads):
        """(rawlen, len, pass-0 bases) of the reads of the walked batch on the selected slot, numpy int32 arrays"""
        import numpy as np
        r, l, o = (np.empty(n_reads, dtype=np.int32) for _ in range(3))
        _check(self.lib.pbsim_batch_fetch_lengths(self.h, r.ctypes.data, l.ctypes.data, o.ctypes.data))
        return r, l, o

    def scratch_state(self):
        """(factor the next batches' rows are laid out with, largest need seen, batches walked twice)"""
        a = (C.c_double * 3)()
        _check(self.lib.pbsim_scratch_state(self.h, a))
        return tuple(a)

    def job_counters(self):
        a = (C.c_int64 * 8)()
        _check(self.lib.pbsim_job_counters(self.h, a))
        return dict(reads_walked=a[0], reads_delivered=a[1], rounds=a[2], bases=a[3], wall_us=a[4], comm_us=a[5],
                    ref_bases=a[6], maf_columns=a[7])

    BREAKDOWN = ("wall", "wait_walk", "finalize", "wait_bytes", "collectives", "account", "tail_block", "drain", "slot_wait",
                 "merge", "begin", "tail_steps", "worker_busy", "topup_rounds", "tail_reads", "depth")

    def job_breakdown(self):
        """where the round loop of the last job_run spent its wall time: {name: microseconds} (+ three counters)"""
        a = (C.c_double * 16)()
        _check(self.lib.pbsim_job_breakdown(self.h, a))
        return dict(zip(self.BREAKDOWN, a))

    def job_sam_header(self, record):
        n = self.lib.pbsim_job_sam_header(self.h, record, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.lib.pbsim_job_sam_header(self.h, record, buf, n + 1)
        return buf.raw[:n]

    def stats_keep_values(self, on=True):
        _check(self.lib.pbsim_stats_keep_values(self.h, 1 if on else 0))

    def stats_merge(self, comm):
        _check(self.lib.pbsim_stats_merge(self.h, C.byref(comm)))

    def stats_add_tasks(self, first_task, out_len, nsub, nins, ndel, qsum=None):
        import numpy as np
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (out_len, nsub, nins, ndel)]
        q = np.ascontiguousarray(qsum, dtype=np.float64) if qsum is not None else None
        p32 = C.POINTER(C.c_int32)
        _check(self.lib.pbsim_stats_add_tasks(self.h, first_task, len(a[0]), *[x.ctypes.data_as(p32) for x in a],
                                              q.ctypes.data_as(C.POINTER(C.c_double)) if q is not None else None))

    def format_stats(self, stats, unit=0):
        buf = C.create_string_buffer(2048)
        n = self.lib.pbsim_format_stats(C.byref(self.params), C.byref(stats), unit, buf, 2048)
        return buf.raw[:n].decode()

    def batch_account(self):
        _check(self.lib.pbsim_batch_account(self.h))

    def reset_stats(self):
        _check(self.lib.pbsim_reset_stats(self.h))

    def unit_quota(self):
        return self.lib.pbsim_unit_quota(self.h)

    def batch_capacity(self):
        return self.lib.pbsim_batch_capacity(self.h)

    def set_scratch_bytes(self, n):
        _check(self.lib.pbsim_set_scratch_bytes(self.h, n))

    def release_pools(self):
        _check(self.lib.pbsim_release_pools(self.h))

    def prof_reset(self):
        _check(self.lib.pbsim_prof_reset(self.h))

    def prof_get(self):
        a, b, c = C.c_double(0), C.c_int64(0), C.c_double(0)
        _check(self.lib.pbsim_prof_get(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def prof_tail(self):
        a, b = C.c_double(0), C.c_int64(0)
        _check(self.lib.pbsim_prof_tail(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def prof_secondary(self):
        a = (C.c_double * 8)()
        _check(self.lib.pbsim_prof_secondary(self.h, a))
        return dict(text_ms=a[0], text_launches=int(a[1]), text_in=int(a[2]), text_out=int(a[3]),
                    deflate_ms=a[4], deflate_launches=int(a[5]), deflate_in=int(a[6]), deflate_out=int(a[7]))

    def prof_wave_launches(self):
        return self.lib.pbsim_prof_wave_launches(self.h)

    def prof_walk_busy(self):
        a = C.c_double(0)
        _check(self.lib.pbsim_prof_walk_busy(self.h, C.byref(a)))
        return a.value

    def dump_table(self, which):
        n = self.lib.pbsim_dump_table(self.h, which, None, 0)
        if n < 0:
            raise PbsimError(self.lib.pbsim_last_error().decode(errors="replace"))
        buf = C.create_string_buffer(n)
        self.lib.pbsim_dump_table(self.h, which, C.cast(buf, C.c_void_p), n)
        return buf.raw
