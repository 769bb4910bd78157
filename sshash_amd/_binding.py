"""ctypes binding of include/sshash_amd.h. Host-side mirror of the reference's dictionary interface."""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass
from typing import Iterable, Optional, Sequence, Union

import numpy as np

INVALID_U64 = 0xFFFFFFFFFFFFFFFF  # reference include/constants.hpp:5

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libsshash_amd.so"


def library_path() -> str:
    # SSHASH_AMD_LIBRARY: another build of the same HIP library (tools/debug builds variants of it side by side)
    return os.environ.get("SSHASH_AMD_LIBRARY") or os.path.join(_HERE, _LIB_NAME)


class SSHashError(RuntimeError):
    """Raised for every non-OK sshash_status; ``.status`` holds the code (include/sshash_amd.h)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"[sshash status {status}] {message}")
        self.status = status


class _BuildConfig(C.Structure):
    _fields_ = [
        ("k", C.c_uint32),
        ("m", C.c_uint32),
        ("seed", C.c_uint64),
        ("canonical", C.c_uint32),
        ("num_threads", C.c_uint32),
        ("lambda_", C.c_double),
        ("verbose", C.c_uint32),
        ("weighted", C.c_uint32),
        ("num_shards", C.c_uint32),
        ("shard_id", C.c_uint32),
    ]


class _Info(C.Structure):
    _fields_ = [
        ("version", C.c_uint8 * 3),
        ("canonical", C.c_uint8),
        ("k", C.c_uint32),
        ("m", C.c_uint32),
        ("words_per_kmer", C.c_uint32),
        ("num_kmers", C.c_uint64),
        ("num_strings", C.c_uint64),
        ("num_bases", C.c_uint64),
        ("num_minimizers", C.c_uint64),
        ("num_bits", C.c_uint64),
        ("skew_partitions", C.c_uint32),
        ("weighted", C.c_uint32),
        ("num_shards", C.c_uint32),
        ("shard_id", C.c_uint32),
    ]


class _Results(C.Structure):
    _fields_ = [
        ("kmer_id", C.c_void_p),
        ("kmer_id_in_string", C.c_void_p),
        ("kmer_offset", C.c_void_p),
        ("string_id", C.c_void_p),
        ("string_begin", C.c_void_p),
        ("string_end", C.c_void_p),
        ("kmer_orientation", C.c_void_p),
        ("minimizer_found", C.c_void_p),
    ]


EXCHANGE_COUNTS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
EXCHANGE_DATA = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p)


class _Exchange(C.Structure):  # sshash_exchange
    _fields_ = [("ctx", C.c_void_p), ("counts", EXCHANGE_COUNTS), ("data", EXCHANGE_DATA)]


class _Report(C.Structure):
    _fields_ = [
        ("num_kmers", C.c_uint64),
        ("num_positive_kmers", C.c_uint64),
        ("num_negative_kmers", C.c_uint64),
        ("num_invalid_kmers", C.c_uint64),
        ("num_searches", C.c_uint64),
        ("num_extensions", C.c_uint64),
    ]


_lib: Optional[C.CDLL] = None


def _preload_hip_runtime() -> None:
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME
    as the system one); if libsshash_amd.so pulled in /opt/rocm's copy first, a later `import torch`
    would load a second runtime and fail with "No HIP GPUs are available" -- and streams / events /
    synchronisation would not be shared between the two. So when torch is installed, its bundled
    runtime is loaded first and libsshash_amd.so binds to it (SONAME match); without torch the
    system ROCm runtime is used."""
    if "torch" in sys.modules:
        return  # torch already loaded its runtime; ours will resolve to it
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def _load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C sshash_amd/csrc`). "
            "There is no pure-Python / CPU fallback for the lookup path."
        )
    _preload_hip_runtime()
    lib = C.CDLL(path)
    P = C.c_void_p
    sigs = {
        "sshash_last_error": (C.c_char_p, []),
        "sshash_build_info": (C.c_char_p, []),
        "sshash_build_config_default": (None, [C.POINTER(_BuildConfig)]),
        "sshash_build_from_fasta": (C.c_int, [C.c_char_p, C.POINTER(_BuildConfig), C.POINTER(P)]),
        "sshash_build_from_packed": (C.c_int, [P, P, C.c_uint64, C.POINTER(_BuildConfig), C.POINTER(P)]),
        "sshash_save": (C.c_int, [P, C.c_char_p]),
        "sshash_load": (C.c_int, [C.c_char_p, C.POINTER(P)]),
        "sshash_free": (None, [P]),
        "sshash_get_info": (C.c_int, [P, C.POINTER(_Info)]),
        "sshash_bucket_stats": (C.c_int, [P, C.POINTER(C.c_uint64 * 64)]),
        "sshash_device_count": (C.c_int, []),
        "sshash_to_device": (C.c_int, [P, C.c_int]),
        "sshash_to_device_table_shard": (C.c_int, [P, C.c_int, C.c_uint32, C.c_uint32]),
        "sshash_device_bytes": (C.c_int, [P, C.c_int, C.POINTER(C.c_uint64)]),
        "sshash_device_stats": (C.c_int, [P, C.c_int, C.POINTER(C.c_uint64 * 16)]),
        "sshash_device_table_histogram": (C.c_int, [P, C.c_int, C.POINTER(C.c_uint64 * 32)]),
        "sshash_sharded_lookup_device": (C.c_int, [P, C.c_int, C.c_uint32, C.c_int, P, C.c_uint64, C.c_int, P, C.POINTER(_Exchange), P]),
        "sshash_sharded_lookup_rccl": (C.c_int, [P, C.c_int, P, C.c_int, P, C.c_uint64, C.c_int, P, P]),
        "sshash_streaming_lookup_device": (C.c_int, [P, C.c_int, P, P, C.c_uint64, C.c_uint64, C.POINTER(_Results), P, P]),
        "sshash_streaming_lookup": (C.c_int, [P, P, P, C.c_uint64, C.POINTER(_Results), C.POINTER(_Report)]),
        "sshash_lookup_packed_device": (C.c_int, [P, C.c_int, P, C.c_uint64, C.c_int, C.POINTER(_Results), P]),
        "sshash_lookup_ascii_device": (C.c_int, [P, C.c_int, P, C.c_uint64, C.c_int, C.POINTER(_Results), P]),
        "sshash_lookup_packed": (C.c_int, [P, P, C.c_uint64, C.c_int, C.POINTER(_Results)]),
        "sshash_lookup_ascii": (C.c_int, [P, P, C.c_uint64, C.c_int, C.POINTER(_Results)]),
        "sshash_neighbours_packed_device": (C.c_int, [P, C.c_int, P, C.c_uint64, C.c_int, C.POINTER(_Results), P]),
        "sshash_neighbours_packed": (C.c_int, [P, P, C.c_uint64, C.c_int, C.POINTER(_Results)]),
        "sshash_string_neighbours": (C.c_int, [P, P, C.c_uint64, C.c_int, C.POINTER(_Results)]),
        "sshash_string_size": (C.c_int, [P, P, C.c_uint64, P]),
        "sshash_string_offsets": (C.c_int, [P, P, C.c_uint64, P, P]),
        "sshash_is_member_packed_device": (C.c_int, [P, C.c_int, P, C.c_uint64, C.c_int, P, P]),
        "sshash_is_member_packed": (C.c_int, [P, P, C.c_uint64, C.c_int, P]),
        "sshash_is_member_ascii": (C.c_int, [P, P, C.c_uint64, C.c_int, P]),
        "sshash_weight": (C.c_int, [P, P, C.c_uint64, P]),
        "sshash_weight_device": (C.c_int, [P, C.c_int, P, C.c_uint64, P, P]),
        "sshash_access": (C.c_int, [P, C.c_uint64, P]),
        "sshash_access_packed": (C.c_int, [P, P, C.c_uint64, P]),
        "sshash_access_packed_device": (C.c_int, [P, C.c_int, P, C.c_uint64, P, P]),
        "sshash_streaming_query_from_file": (C.c_int, [P, C.c_char_p, C.c_int, C.POINTER(_Report)]),
        "sshash_streaming_query": (C.c_int, [P, P, P, C.c_uint64, C.POINTER(_Report)]),
        "sshash_streaming_query_device": (C.c_int, [P, C.c_int, P, P, C.c_uint64, C.c_uint64, P, P]),
        "sshash_route_packed_device": (C.c_int, [P, C.c_int, P, C.c_uint64, C.c_uint32, P, P, P]),
        "sshash_route_bucket_device": (C.c_int, [P, C.c_int, P, C.c_uint64, C.c_uint32, C.c_int, P, P, P, P]),
        "sshash_route_bucket_by_key_device": (C.c_int, [P, C.c_int, P, C.c_uint64, C.c_uint32, P, P, P, P]),
        "sshash_route_combine_device": (C.c_int, [P, C.c_int, P, P, C.c_uint64, P, P]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here == ABI symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


C_ABI_SYMBOLS = (
    "sshash_last_error sshash_build_info sshash_build_config_default sshash_build_from_fasta sshash_build_from_packed sshash_save "
    "sshash_load sshash_free sshash_get_info sshash_bucket_stats sshash_device_count sshash_to_device sshash_to_device_table_shard sshash_device_bytes sshash_device_stats sshash_device_table_histogram "
    "sshash_lookup_packed_device sshash_lookup_ascii_device sshash_lookup_packed sshash_lookup_ascii "
    "sshash_neighbours_packed_device sshash_neighbours_packed sshash_string_neighbours sshash_string_size sshash_string_offsets "
    "sshash_is_member_packed_device sshash_is_member_packed sshash_is_member_ascii sshash_access sshash_access_packed "
    "sshash_access_packed_device sshash_weight sshash_weight_device "
    "sshash_streaming_query_from_file sshash_streaming_query sshash_streaming_query_device "
    "sshash_streaming_lookup sshash_streaming_lookup_device sshash_sharded_lookup_device sshash_sharded_lookup_rccl "
    "sshash_route_packed_device sshash_route_bucket_device sshash_route_bucket_by_key_device sshash_route_combine_device"
).split()


def _check(status: int) -> None:
    if status != 0:
        raise SSHashError(status, _load().sshash_last_error().decode("utf-8", "replace"))


def build_info() -> dict:
    """sshash_build_info(): {"isa_guard": "guarded" | "plain", "arch": "gfx950"}"""
    return dict(kv.split("=", 1) for kv in _load().sshash_build_info().decode().split(";") if "=" in kv)


def device_count() -> int:
    return int(_load().sshash_device_count())


@dataclass
class LookupResult:
    """Struct-of-arrays form of lookup_result (reference include/util.hpp:38-62)."""

    kmer_id: np.ndarray
    kmer_id_in_string: Optional[np.ndarray] = None
    kmer_offset: Optional[np.ndarray] = None
    kmer_orientation: Optional[np.ndarray] = None
    string_id: Optional[np.ndarray] = None
    string_begin: Optional[np.ndarray] = None
    string_end: Optional[np.ndarray] = None
    minimizer_found: Optional[np.ndarray] = None


@dataclass
class StreamingQueryReport:
    """streaming_query_report (reference include/util.hpp:21-36)."""

    num_kmers: int = 0
    num_positive_kmers: int = 0
    num_negative_kmers: int = 0
    num_invalid_kmers: int = 0
    num_searches: int = 0
    num_extensions: int = 0


def encode_kmers(kmers: Sequence[Union[str, bytes]], k: int) -> np.ndarray:
    """ASCII k-mers -> contiguous (n, k) uint8 array as the *_ascii entry points expect."""
    buf = bytearray()
    for s in kmers:
        b = s.encode("ascii") if isinstance(s, str) else bytes(s)
        if len(b) < k:
            raise ValueError(f"k-mer shorter than k={k}: {s!r}")
        buf += b[:k]
    return np.frombuffer(bytes(buf), dtype=np.uint8).reshape(len(kmers), k)


KmerBatch = Union[np.ndarray, Sequence[str], Sequence[bytes], str, bytes]


class Dictionary:
    """Handle on one SSHash dictionary (host index + optional HBM replicas)."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)
        info = _Info()
        _check(_load().sshash_get_info(self._h, C.byref(info)))
        self._info = info

    # ---- construction / persistence ------------------------------------------------------
    @staticmethod
    def _config(k, m, seed, canonical, num_threads, lambda_, verbose, num_shards=1, shard_id=0) -> _BuildConfig:
        cfg = _BuildConfig()
        _load().sshash_build_config_default(C.byref(cfg))
        cfg.num_shards, cfg.shard_id = int(num_shards), int(shard_id)
        cfg.k, cfg.m, cfg.seed = int(k), int(m), int(seed)
        cfg.canonical = 1 if canonical else 0
        cfg.num_threads = int(num_threads)
        cfg.lambda_ = float(lambda_)
        cfg.verbose = 1 if verbose else 0
        return cfg

    @classmethod
    def build(cls, input_filename: str, k: int = 31, m: int = 20, seed: int = 1, canonical: bool = False,
              num_threads: int = 0, lambda_: float = 5.0, verbose: bool = False, num_shards: int = 1,
              shard_id: int = 0, weighted: bool = False) -> "Dictionary":
        """dictionary::build(input_filename, build_configuration) -- reference include/dictionary.hpp:28.
        num_shards > 1: build only the part of the sparse-and-skew index owned by `shard_id`.
        weighted: the FASTA headers carry the k-mer abundances (build_configuration::weighted)."""
        cfg = cls._config(k, m, seed, canonical, num_threads, lambda_, verbose, num_shards, shard_id)
        cfg.weighted = 1 if weighted else 0
        h = C.c_void_p()
        _check(_load().sshash_build_from_fasta(os.fsencode(input_filename), C.byref(cfg), C.byref(h)))
        return cls(h.value)

    @classmethod
    def build_from_packed(cls, words: np.ndarray, endpoints: np.ndarray, k: int = 31, m: int = 20, seed: int = 1,
                          canonical: bool = False, num_threads: int = 0, lambda_: float = 5.0,
                          verbose: bool = False, num_shards: int = 1, shard_id: int = 0) -> "Dictionary":
        words = np.ascontiguousarray(words, dtype=np.uint64)
        endpoints = np.ascontiguousarray(endpoints, dtype=np.uint64)
        need = (2 * int(endpoints[-1]) + 63) // 64
        if words.size < need:
            raise ValueError("packed words shorter than endpoints[-1] bases")
        cfg = cls._config(k, m, seed, canonical, num_threads, lambda_, verbose, num_shards, shard_id)
        h = C.c_void_p()
        _check(_load().sshash_build_from_packed(words.ctypes.data, endpoints.ctypes.data, endpoints.size - 1,
                                                C.byref(cfg), C.byref(h)))
        return cls(h.value)

    @classmethod
    def load(cls, index_filename: str) -> "Dictionary":
        h = C.c_void_p()
        _check(_load().sshash_load(os.fsencode(index_filename), C.byref(h)))
        return cls(h.value)

    def save(self, index_filename: str) -> None:
        _check(_load().sshash_save(self._h, os.fsencode(index_filename)))

    def close(self) -> None:
        if self._h:
            _load().sshash_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- accessors (reference include/dictionary.hpp:31-38) --------------------------------
    def k(self) -> int: return int(self._info.k)
    def m(self) -> int: return int(self._info.m)
    def canonical(self) -> bool: return bool(self._info.canonical)
    def num_kmers(self) -> int: return int(self._info.num_kmers)
    def num_strings(self) -> int: return int(self._info.num_strings)
    def num_bases(self) -> int: return int(self._info.num_bases)
    def num_minimizers(self) -> int: return int(self._info.num_minimizers)
    def num_bits(self) -> int: return int(self._info.num_bits)
    def words_per_kmer(self) -> int: return int(self._info.words_per_kmer)
    def vnum(self): return tuple(self._info.version)
    def num_shards(self) -> int: return int(self._info.num_shards)
    def shard_id(self) -> int: return int(self._info.shard_id)
    def weighted(self) -> bool: return bool(self._info.weighted)

    # ---- device residency -----------------------------------------------------------------
    def to_device(self, device: int = 0, table_shards: int = 1, table_shard_id: int = 0) -> "Dictionary":
        """Upload to `device`. table_shards > 1: the replica's super-k-mer table holds only its share of the keys
        (see sharded.py)."""
        if table_shards > 1:
            _check(_load().sshash_to_device_table_shard(self._h, int(device), int(table_shards), int(table_shard_id)))
            return self
        _check(_load().sshash_to_device(self._h, int(device)))
        return self

    def device_bytes(self, device: int = 0) -> int:
        out = C.c_uint64()
        _check(_load().sshash_device_bytes(self._h, int(device), C.byref(out)))
        return int(out.value)

    def bucket_stats(self) -> dict:
        """The statistics `sshash build --verbose` prints for the sparse and skew index (sshash_bucket_stats)."""
        out = (C.c_uint64 * 64)()
        _check(_load().sshash_bucket_stats(self._h, C.byref(out)))
        v = [int(x) for x in out]
        return {"num_minimizers": v[0], "num_minimizer_positions": v[1], "num_buckets_larger_than_1_not_in_skew_index": v[2],
                "num_minimizer_positions_of_buckets_larger_than_1": v[3], "num_buckets_in_skew_index": v[4],
                "num_minimizer_positions_of_buckets_in_skew_index": v[5], "num_kmers_in_skew_index": v[6], "max_bucket_size": v[7],
                "num_kmers_in_skew_partition": v[8:8 + v[35]], "buckets_with_n_positions": v[16:32], "num_kmers": v[32],
                "num_strings": v[33], "num_bases": v[34], "max_string_length": v[36]}

    def device_stats(self, device: int = 0) -> dict:
        out = (C.c_uint64 * 16)()
        _check(_load().sshash_device_stats(self._h, int(device), C.byref(out)))
        reasons = {0: None, 1: "disabled", 2: "minimizer shard", 3: "more than 2^39 bases", 4: "too many items for one build pass",
                   5: "not enough free HBM"}
        return {"bytes": int(out[0]), "directory_sectors": int(out[1]), "directory_overflow_sectors": int(out[2]),
                "directory_keys": int(out[3]), "sk_slots": int(out[4]), "sk_keys": int(out[5]), "sk_inline_keys": int(out[6]),
                "sk_deferred_keys": int(out[7]), "sk_slots_used": int(out[8]), "sk_heavy_keys": int(out[9]),
                "sk_heavy_kmers": int(out[10]), "sk_absent_reason": reasons.get(int(out[11]), str(int(out[11]))),
                "sk_bytes": int(out[12]), "sk_key_length": int(out[13]), "sk_load_factor": round(int(out[8]) / int(out[4]), 4) if int(out[4]) else 0.0}

    def device_table_histogram(self, device: int = 0) -> dict:
        """Keys of the super-k-mer table by number of occurrences (sshash_device_table_histogram)."""
        out = (C.c_uint64 * 32)()
        _check(_load().sshash_device_table_histogram(self._h, int(device), C.byref(out)))
        v = [int(x) for x in out]
        bins = ["1", "2", "3", "4", "5-8", "9-16", "17-64", "65-1024", ">1024"]
        return {"keys_by_occurrences": dict(zip(bins, v[0:9])), "super_kmers_by_occurrences_of_their_key": dict(zip(bins, v[9:18])),
                "super_kmers": v[18], "slots_asked_for": v[19]}

    def _as_batch(self, kmers: KmerBatch):
        """-> (is_ascii, contiguous ndarray, n)"""
        if isinstance(kmers, (str, bytes)):
            kmers = [kmers]
        if isinstance(kmers, np.ndarray) and kmers.dtype == np.uint64:
            a = np.ascontiguousarray(kmers)
            w = self.words_per_kmer()
            if a.size % w:
                raise ValueError("packed k-mer array length is not a multiple of words_per_kmer")
            return False, a, a.size // w
        if isinstance(kmers, np.ndarray) and kmers.dtype == np.uint8:
            a = np.ascontiguousarray(kmers)
            if a.size % self.k():
                raise ValueError("ASCII k-mer buffer length is not a multiple of k")
            return True, a, a.size // self.k()
        a = encode_kmers(list(kmers), self.k())
        return True, np.ascontiguousarray(a), a.shape[0]

    def lookup(self, kmers: KmerBatch, check_reverse_complement: bool = True, full: bool = False,
               out: Optional[np.ndarray] = None) -> LookupResult:
        """Batched dictionary::lookup (reference src/dictionary.cpp:58-78). Host buffers in/out. `out`: a uint64 array
        of n entries to receive the ids (e.g. page-locked memory: with input and output both page-locked the library
        copies from and to them directly instead of staging through its own pinned lanes)."""
        is_ascii, a, n = self._as_batch(kmers)
        if out is not None and (out.dtype != np.uint64 or out.size != n or not out.flags["C_CONTIGUOUS"]):
            raise ValueError("out must be a contiguous uint64 array with one entry per k-mer")
        res = LookupResult(kmer_id=out if out is not None else np.empty(n, dtype=np.uint64))
        r = _Results()
        r.kmer_id = res.kmer_id.ctypes.data
        if full:
            for name in ("kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end"):
                arr = np.empty(n, dtype=np.uint64)
                setattr(res, name, arr)
                setattr(r, name, arr.ctypes.data)
            res.kmer_orientation = np.empty(n, dtype=np.int8)
            res.minimizer_found = np.empty(n, dtype=np.uint8)
            r.kmer_orientation = res.kmer_orientation.ctypes.data
            r.minimizer_found = res.minimizer_found.ctypes.data
        fn = _load().sshash_lookup_ascii if is_ascii else _load().sshash_lookup_packed
        _check(fn(self._h, a.ctypes.data, n, 1 if check_reverse_complement else 0, C.byref(r)))
        return res

    def is_member(self, kmers: KmerBatch, check_reverse_complement: bool = True) -> np.ndarray:
        """Batched dictionary::is_member (reference src/dictionary.cpp:80-88)."""
        is_ascii, a, n = self._as_batch(kmers)
        out = np.empty(n, dtype=np.uint8)
        fn = _load().sshash_is_member_ascii if is_ascii else _load().sshash_is_member_packed
        _check(fn(self._h, a.ctypes.data, n, 1 if check_reverse_complement else 0, out.ctypes.data))
        return out.astype(bool)

    def lookup_device(self, device: int, d_kmers: int, n: int, d_kmer_id: int, check_reverse_complement: bool = True,
                      stream: int = 0, ascii_input: bool = False, **optional_outputs: int) -> None:
        """Device-pointer entry point: all pointers are raw HBM addresses (e.g. torch ``.data_ptr()``)."""
        r = _Results()
        r.kmer_id = d_kmer_id
        for name, ptr in optional_outputs.items():
            setattr(r, name, ptr)
        fn = _load().sshash_lookup_ascii_device if ascii_input else _load().sshash_lookup_packed_device
        _check(fn(self._h, int(device), C.c_void_p(d_kmers), int(n), 1 if check_reverse_complement else 0, C.byref(r),
                  C.c_void_p(stream)))

    def neighbours(self, kmers: np.ndarray, check_reverse_complement: bool = True, full: bool = False) -> LookupResult:
        """Batched dictionary::kmer_neighbours (reference src/dictionary.cpp:111-126,176-187) over packed k-mers:
        every array of the result has 8 entries per query -- forward neighbours with A,C,T,G appended (index = 2-bit
        code of the character, the reference's alphabet order), then backward neighbours with A,C,T,G prepended."""
        a = np.ascontiguousarray(kmers, dtype=np.uint64)
        n = a.size // self.words_per_kmer()
        res = LookupResult(kmer_id=np.empty(8 * n, dtype=np.uint64))
        r = _Results()
        r.kmer_id = res.kmer_id.ctypes.data
        if full:
            for name in ("kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end"):
                arr = np.empty(8 * n, dtype=np.uint64)
                setattr(res, name, arr)
                setattr(r, name, arr.ctypes.data)
            res.kmer_orientation = np.empty(8 * n, dtype=np.int8)
            res.minimizer_found = np.empty(8 * n, dtype=np.uint8)
            r.kmer_orientation = res.kmer_orientation.ctypes.data
            r.minimizer_found = res.minimizer_found.ctypes.data
        _check(_load().sshash_neighbours_packed(self._h, a.ctypes.data, n, 1 if check_reverse_complement else 0, C.byref(r)))
        return res

    def string_size(self, string_ids: Iterable[int]) -> np.ndarray:
        """dictionary::string_size (reference src/dictionary.cpp:102-109): k-mers per string."""
        sids = np.ascontiguousarray(np.asarray(string_ids, dtype=np.uint64))
        out = np.empty(sids.size, dtype=np.uint64)
        _check(_load().sshash_string_size(self._h, sids.ctypes.data, sids.size, out.ctypes.data))
        return out

    def string_offsets(self, string_ids: Iterable[int]):
        """dictionary::string_offsets (reference include/dictionary.hpp:105-108): [begin, end) of each string, in bases."""
        sids = np.ascontiguousarray(np.asarray(string_ids, dtype=np.uint64))
        begin, end = np.empty(sids.size, dtype=np.uint64), np.empty(sids.size, dtype=np.uint64)
        _check(_load().sshash_string_offsets(self._h, sids.ctypes.data, sids.size, begin.ctypes.data, end.ctypes.data))
        return begin, end

    def string_neighbours(self, string_ids: Iterable[int], check_reverse_complement: bool = True) -> np.ndarray:
        """Batched dictionary::string_neighbours (reference src/dictionary.cpp:189-201): ids, 8 per string."""
        sids = np.ascontiguousarray(np.asarray(string_ids, dtype=np.uint64))
        ids = np.empty(8 * sids.size, dtype=np.uint64)
        r = _Results()
        r.kmer_id = ids.ctypes.data
        _check(_load().sshash_string_neighbours(self._h, sids.ctypes.data, sids.size, 1 if check_reverse_complement else 0, C.byref(r)))
        return ids

    def neighbours_device(self, device: int, d_kmers: int, n: int, d_kmer_id: int, check_reverse_complement: bool = True,
                          stream: int = 0, **optional_outputs: int) -> None:
        """Device-pointer form: d_kmer_id (and every optional output) has room for 8*n entries."""
        r = _Results()
        r.kmer_id = d_kmer_id
        for name, ptr in optional_outputs.items():
            setattr(r, name, ptr)
        _check(_load().sshash_neighbours_packed_device(self._h, int(device), C.c_void_p(d_kmers), int(n),
                                                       1 if check_reverse_complement else 0, C.byref(r), C.c_void_p(stream)))

    def is_member_device(self, device: int, d_kmers: int, n: int, d_out: int, check_reverse_complement: bool = True,
                         stream: int = 0) -> None:
        _check(_load().sshash_is_member_packed_device(self._h, int(device), C.c_void_p(d_kmers), int(n),
                                                      1 if check_reverse_complement else 0, C.c_void_p(d_out),
                                                      C.c_void_p(stream)))

    # ---- access (host) ----------------------------------------------------------------------
    def access(self, kmer_id: int) -> str:
        buf = C.create_string_buffer(self.k())
        _check(_load().sshash_access(self._h, int(kmer_id), buf))
        return buf.raw.decode("ascii")

    def access_packed(self, kmer_ids: Iterable[int]) -> np.ndarray:
        ids = np.ascontiguousarray(np.asarray(kmer_ids, dtype=np.uint64))
        out = np.empty(ids.size * self.words_per_kmer(), dtype=np.uint64)
        _check(_load().sshash_access_packed(self._h, ids.ctypes.data, ids.size, out.ctypes.data))
        return out

    # ---- weights ---------------------------------------------------------------------------------
    def weight(self, kmer_ids: Iterable[int]) -> np.ndarray:
        """Batched dictionary::weight (reference src/dictionary.cpp:96-100, include/weights.hpp:147-152), host side."""
        ids = np.ascontiguousarray(np.asarray(kmer_ids, dtype=np.uint64))
        out = np.empty(ids.size, dtype=np.uint64)
        _check(_load().sshash_weight(self._h, ids.ctypes.data, ids.size, out.ctypes.data))
        return out

    def weight_device(self, device: int, d_kmer_ids: int, n: int, d_out: int, stream: int = 0) -> None:
        _check(_load().sshash_weight_device(self._h, int(device), C.c_void_p(d_kmer_ids), int(n), C.c_void_p(d_out),
                                            C.c_void_p(stream)))

    def route_bucket_device(self, device: int, d_kmers: int, n: int, num_shards: int, d_cursors: int, d_send: int = 0,
                            d_slots: int = 0, check_reverse_complement: bool = True, stream: int = 0) -> None:
        """Count (d_send == 0) or scatter the messages of a routed batch; see include/sshash_amd.h."""
        _check(_load().sshash_route_bucket_device(self._h, int(device), C.c_void_p(d_kmers), int(n), int(num_shards),
                                                  1 if check_reverse_complement else 0, C.c_void_p(d_cursors),
                                                  C.c_void_p(d_send or None), C.c_void_p(d_slots or None), C.c_void_p(stream)))

    def route_bucket_by_key_device(self, device: int, d_kmers: int, n: int, num_shards: int, d_cursors: int, d_send: int = 0,
                                   d_slots: int = 0, stream: int = 0) -> None:
        """The same for table shards: one message per query, to the owner of its table key."""
        _check(_load().sshash_route_bucket_by_key_device(self._h, int(device), C.c_void_p(d_kmers), int(n), int(num_shards),
                                                         C.c_void_p(d_cursors), C.c_void_p(d_send or None),
                                                         C.c_void_p(d_slots or None), C.c_void_p(stream)))

    def route_combine_device(self, device: int, d_replies: int, d_slots: int, m: int, d_out: int, stream: int = 0) -> None:
        _check(_load().sshash_route_combine_device(self._h, int(device), C.c_void_p(d_replies), C.c_void_p(d_slots), int(m),
                                                   C.c_void_p(d_out), C.c_void_p(stream)))

    def sharded_lookup_device(self, device: int, num_ranks: int, by_table_key: bool, d_kmers: int, n: int, d_kmer_id: int,
                              counts_fn, data_fn, check_reverse_complement: bool = True, stream: int = 0) -> None:
        """sshash_sharded_lookup_device: route -> exchange -> lookup -> return -> combine in one call; the exchange is
        `counts_fn(send: list[int]) -> list[int]` and `data_fn(send_ptr, send_counts, recv_ptr, recv_counts, elem_bytes,
        stream) -> None` (device pointers)."""
        errors = []

        def counts(_ctx, send, recv):
            try:
                got = counts_fn([int(send[p]) for p in range(num_ranks)])
                for p in range(num_ranks):
                    recv[p] = int(got[p])
                return 0
            except Exception as e:  # noqa: BLE001 -- must not unwind through the C frames
                errors.append(e)
                return 1

        def data(_ctx, send, send_counts, recv, recv_counts, elem_bytes, hip_stream):
            try:
                data_fn(int(send or 0), [int(send_counts[p]) for p in range(num_ranks)], int(recv or 0),
                        [int(recv_counts[p]) for p in range(num_ranks)], int(elem_bytes), int(hip_stream or 0))
                return 0
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                return 1

        x = _Exchange(None, EXCHANGE_COUNTS(counts), EXCHANGE_DATA(data))
        status = _load().sshash_sharded_lookup_device(self._h, int(device), int(num_ranks), 1 if by_table_key else 0, C.c_void_p(d_kmers),
                                                      int(n), 1 if check_reverse_complement else 0, C.c_void_p(d_kmer_id), C.byref(x),
                                                      C.c_void_p(stream))
        if errors:
            raise errors[0]
        _check(status)

    def route_device(self, device: int, d_kmers: int, n: int, num_shards: int, d_owner_fwd: int, d_owner_rc: int,
                     stream: int = 0) -> None:
        """Owner shard of each query's forward / reverse-complement minimizer (uint32 device arrays)."""
        _check(_load().sshash_route_packed_device(self._h, int(device), C.c_void_p(d_kmers), int(n), int(num_shards),
                                                  C.c_void_p(d_owner_fwd), C.c_void_p(d_owner_rc), C.c_void_p(stream)))

    def access_packed_device(self, device: int, d_ids: int, n: int, d_out: int, stream: int = 0) -> None:
        _check(_load().sshash_access_packed_device(self._h, int(device), C.c_void_p(d_ids), int(n), C.c_void_p(d_out),
                                                   C.c_void_p(stream)))

    # ---- streaming query ----------------------------------------------------------------------
    @staticmethod
    def _report(r: _Report) -> StreamingQueryReport:
        return StreamingQueryReport(r.num_kmers, r.num_positive_kmers, r.num_negative_kmers, r.num_invalid_kmers,
                                    r.num_searches, r.num_extensions)

    def streaming_query_from_file(self, filename: str, multiline: bool = False) -> StreamingQueryReport:
        """dictionary::streaming_query_from_file (reference include/dictionary.hpp:81-82, src/query.cpp:118-175)."""
        r = _Report()
        _check(_load().sshash_streaming_query_from_file(self._h, os.fsencode(filename), 1 if multiline else 0, C.byref(r)))
        return self._report(r)

    def streaming_query(self, reads: Sequence[Union[str, bytes]]) -> StreamingQueryReport:
        """One streaming_query per read (reset between reads), reads given in memory."""
        chunks = [s.encode("ascii", "replace") if isinstance(s, str) else bytes(s) for s in reads]
        offsets = np.zeros(len(chunks) + 1, dtype=np.uint64)
        if chunks:
            offsets[1:] = np.cumsum([len(c) for c in chunks], dtype=np.uint64)
        bases = np.frombuffer(b"".join(chunks) or b"\0", dtype=np.uint8)
        r = _Report()
        _check(_load().sshash_streaming_query(self._h, bases.ctypes.data, offsets.ctypes.data, len(chunks), C.byref(r)))
        return self._report(r)

    def streaming_lookup(self, reads: Sequence[Union[str, bytes]], full: bool = False):
        """streaming_query::lookup for every k-mer of every read (reference include/streaming_query.hpp:56-109), batched.
        -> (list of LookupResult, one per read, len(read) - k + 1 entries each; StreamingQueryReport)."""
        chunks = [s.encode("ascii", "replace") if isinstance(s, str) else bytes(s) for s in reads]
        offsets = np.zeros(len(chunks) + 1, dtype=np.uint64)
        if chunks:
            offsets[1:] = np.cumsum([len(c) for c in chunks], dtype=np.uint64)
        bases = np.frombuffer(b"".join(chunks) or b"\0", dtype=np.uint8)
        total = max(int(offsets[-1]), 1)
        arrays = {"kmer_id": np.full(total, INVALID_U64, dtype=np.uint64)}
        if full:
            for name in ("kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end"):
                arrays[name] = np.full(total, INVALID_U64, dtype=np.uint64)
            arrays["kmer_orientation"] = np.ones(total, dtype=np.int8)
        r = _Results()
        for name, a in arrays.items():
            setattr(r, name, a.ctypes.data)
        rep = _Report()
        _check(_load().sshash_streaming_lookup(self._h, bases.ctypes.data, offsets.ctypes.data, len(chunks), C.byref(r), C.byref(rep)))
        per_read = []
        k = self.k()
        for i, c in enumerate(chunks):
            lo = int(offsets[i])
            n = max(0, len(c) - k + 1)
            per_read.append(LookupResult(**{name: a[lo:lo + n] for name, a in arrays.items()}))
        return per_read, self._report(rep)

    def streaming_lookup_device(self, device: int, d_bases: int, d_read_offsets: int, num_reads: int, total_bases: int,
                                d_kmer_id: int, d_report: int = 0, stream: int = 0, **optional_outputs: int) -> None:
        r = _Results()
        r.kmer_id = d_kmer_id
        for name, ptr in optional_outputs.items():
            setattr(r, name, ptr)
        _check(_load().sshash_streaming_lookup_device(self._h, int(device), C.c_void_p(d_bases), C.c_void_p(d_read_offsets),
                                                      int(num_reads), int(total_bases), C.byref(r), C.c_void_p(d_report), C.c_void_p(stream)))

    def streaming_query_device(self, device: int, d_bases: int, d_read_offsets: int, num_reads: int, d_report: int,
                               stream: int = 0, total_bases: int = 0) -> None:
        """`total_bases` = read_offsets[num_reads] when the caller knows it: the call then only enqueues work (0: it reads the
        eight bytes back and waits for the stream first)."""
        _check(_load().sshash_streaming_query_device(self._h, int(device), C.c_void_p(d_bases), C.c_void_p(d_read_offsets),
                                                     int(num_reads), int(total_bases), C.c_void_p(d_report), C.c_void_p(stream)))
