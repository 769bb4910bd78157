"""ctypes wrapper of the CPU restatement (oracle/sshash_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsshash_oracle.so")

INVALID_U64 = 0xFFFFFFFFFFFFFFFF

RESULT_DTYPE = np.dtype(
    [
        ("kmer_id", "<u8"),
        ("kmer_id_in_string", "<u8"),
        ("kmer_offset", "<u8"),
        ("kmer_orientation", "<i8"),
        ("string_id", "<u8"),
        ("string_begin", "<u8"),
        ("string_end", "<u8"),
        ("minimizer_found", "u1"),
        ("pad", "u1", (7,)),
    ]
)
assert RESULT_DTYPE.itemsize == 64


class _Info(C.Structure):
    _fields_ = [
        ("k", C.c_uint32), ("m", C.c_uint32), ("canonical", C.c_uint32), ("words_per_kmer", C.c_uint32),
        ("num_kmers", C.c_uint64), ("num_strings", C.c_uint64), ("num_bases", C.c_uint64),
        ("num_minimizers", C.c_uint64), ("hash_magic", C.c_uint64),
    ]


_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement (and oracle/_ref when the reference checkout is present)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "sshash_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"], stdout=subprocess.DEVNULL)
    return _LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        P = C.c_void_p
        L.oracle_load.restype = C.c_int
        L.oracle_load.argtypes = [C.c_char_p, C.POINTER(P), C.c_char_p, C.c_int]
        L.oracle_free.argtypes = [P]
        L.oracle_get_info.argtypes = [P, C.POINTER(_Info)]
        L.oracle_lookup_packed.argtypes = [P, P, C.c_uint64, C.c_int, P]
        L.oracle_lookup_ascii.argtypes = [P, P, C.c_uint64, C.c_int, P]
        L.oracle_lookup_ids.argtypes = [P, P, C.c_uint64, C.c_int, P, C.c_int]
        L.oracle_count_bytes.restype = C.c_uint64
        L.oracle_count_bytes.argtypes = [P, P, C.c_uint64, C.c_int]
        L.oracle_access.argtypes = [P, C.c_uint64, P]
        L.oracle_weights.argtypes = [P, P, C.c_uint64, P]
        L.oracle_weights.restype = C.c_int
        L.oracle_streaming_query.argtypes = [P, P, P, C.c_uint64, P]
        L.oracle_streaming_read.argtypes = [P, P, C.c_uint64, P]
        L.oracle_streaming_count_bytes.restype = C.c_uint64
        L.oracle_streaming_count_bytes.argtypes = [P, P, P, C.c_uint64]
        L.oracle_encode_kmer.argtypes = [C.c_char_p, C.c_uint32, P]
        L.oracle_revcomp.argtypes = [P, C.c_uint32, C.c_int, P]
        L.oracle_minimizer.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, P, P]
        L.oracle_city128.argtypes = [P, C.c_int, C.c_uint64, P]
        L.oracle_xxh64_u64.restype = C.c_uint64
        L.oracle_xxh64_u64.argtypes = [C.c_uint64, C.c_uint64]
        L.oracle_is_valid_base.restype = C.c_int
        L.oracle_is_valid_base.argtypes = [C.c_char]
        _lib = L
    return _lib


class OracleIndex:
    """The CPU restatement loaded from an index file written by ``Dictionary.save``."""

    def __init__(self, filename: str):
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        status = lib().oracle_load(os.fsencode(filename), C.byref(self._h), err, 512)
        if status != 0:
            raise RuntimeError(f"[oracle status {status}] {err.value.decode()}")
        info = _Info()
        lib().oracle_get_info(self._h, C.byref(info))
        self.k, self.m, self.canonical, self.W = info.k, info.m, bool(info.canonical), info.words_per_kmer
        self.num_kmers, self.num_strings, self.num_bases = info.num_kmers, info.num_strings, info.num_bases
        self.num_minimizers, self.hash_magic = info.num_minimizers, info.hash_magic

    def close(self):
        if self._h:
            lib().oracle_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lookup_packed(self, kmers: np.ndarray, check_rc: bool = True) -> np.ndarray:
        a = np.ascontiguousarray(kmers, dtype=np.uint64)
        n = a.size // self.W
        out = np.zeros(n, dtype=RESULT_DTYPE)
        lib().oracle_lookup_packed(self._h, a.ctypes.data, n, int(check_rc), out.ctypes.data)
        return out

    def lookup_ascii(self, kmers: np.ndarray, check_rc: bool = True) -> np.ndarray:
        a = np.ascontiguousarray(kmers, dtype=np.uint8)
        n = a.size // self.k
        out = np.zeros(n, dtype=RESULT_DTYPE)
        lib().oracle_lookup_ascii(self._h, a.ctypes.data, n, int(check_rc), out.ctypes.data)
        return out

    def lookup_ids(self, kmers: np.ndarray, check_rc: bool = True, num_threads: int = 1) -> np.ndarray:
        a = np.ascontiguousarray(kmers, dtype=np.uint64)
        n = a.size // self.W
        ids = np.empty(n, dtype=np.uint64)
        lib().oracle_lookup_ids(self._h, a.ctypes.data, n, int(check_rc), ids.ctypes.data, int(num_threads))
        return ids

    def count_bytes(self, kmers: np.ndarray, check_rc: bool = True) -> int:
        a = np.ascontiguousarray(kmers, dtype=np.uint64)
        return int(lib().oracle_count_bytes(self._h, a.ctypes.data, a.size // self.W, int(check_rc)))

    def weights(self, kmer_ids) -> np.ndarray:
        """weights::weight for each id (reference include/weights.hpp:147-152)."""
        ids = np.ascontiguousarray(kmer_ids, dtype=np.uint64)
        out = np.empty(ids.size, dtype=np.uint64)
        if not lib().oracle_weights(self._h, ids.ctypes.data, ids.size, out.ctypes.data):
            raise ValueError("the dictionary does not store weights (or id out of range)")
        return out

    def access(self, kmer_id: int) -> str:
        buf = C.create_string_buffer(self.k)
        lib().oracle_access(self._h, int(kmer_id), buf)
        return buf.raw.decode("ascii")

    def streaming_query(self, reads) -> dict:
        chunks = [s.encode("ascii", "replace") if isinstance(s, str) else bytes(s) for s in reads]
        offsets = np.zeros(len(chunks) + 1, dtype=np.uint64)
        if chunks:
            offsets[1:] = np.cumsum([len(c) for c in chunks], dtype=np.uint64)
        bases = np.frombuffer(b"".join(chunks) or b"\0", dtype=np.uint8)
        rep = np.zeros(6, dtype=np.uint64)
        lib().oracle_streaming_query(self._h, bases.ctypes.data, offsets.ctypes.data, len(chunks), rep.ctypes.data)
        names = ["num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers", "num_searches", "num_extensions"]
        return dict(zip(names, (int(x) for x in rep)))

    def streaming_count_bytes(self, reads) -> int:
        """Algorithmic bytes of streaming_query(reads): 8 B per distinct index word the reference's state machine dereferences per
        k-mer (the lookups of a seed() that is not cut short, the strings' next k-mer of an extension) + 1 B per base."""
        chunks = [s.encode("ascii", "replace") if isinstance(s, str) else bytes(s) for s in reads]
        offsets = np.zeros(len(chunks) + 1, dtype=np.uint64)
        if chunks:
            offsets[1:] = np.cumsum([len(c) for c in chunks], dtype=np.uint64)
        bases = np.frombuffer(b"".join(chunks) or b"\0", dtype=np.uint8)
        return int(lib().oracle_streaming_count_bytes(self._h, bases.ctypes.data, offsets.ctypes.data, len(chunks)))

    def streaming_read(self, read) -> np.ndarray:
        b = read.encode("ascii", "replace") if isinstance(read, str) else bytes(read)
        n = max(0, len(b) - self.k + 1)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        buf = np.frombuffer(b or b"\0", dtype=np.uint8)
        lib().oracle_streaming_read(self._h, buf.ctypes.data, len(b), out.ctypes.data)
        return out


# ---- primitives -------------------------------------------------------------------------------

def encode_kmer(s: str, k: int):
    out = (C.c_uint64 * 2)()
    lib().oracle_encode_kmer(s.encode("ascii"), k, out)
    return int(out[0]), int(out[1])


def revcomp(lo: int, hi: int, k: int, words: int):
    a = (C.c_uint64 * 2)(lo, hi)
    out = (C.c_uint64 * 2)()
    lib().oracle_revcomp(a, k, words, out)
    return int(out[0]), int(out[1])


def minimizer(lo: int, hi: int, k: int, m: int, magic: int, words: int):
    a = (C.c_uint64 * 2)(lo, hi)
    v, p = C.c_uint64(), C.c_uint64()
    lib().oracle_minimizer(a, k, m, magic, words, C.byref(v), C.byref(p))
    return int(v.value), int(p.value)


def city128(key: bytes, seed: int):
    out = (C.c_uint64 * 2)()
    buf = C.create_string_buffer(key, len(key))
    lib().oracle_city128(buf, len(key), seed, out)
    return int(out[0]), int(out[1])


def xxh64_u64(value: int, seed: int) -> int:
    return int(lib().oracle_xxh64_u64(value, seed))


def is_valid_base(c: str) -> bool:
    return bool(lib().oracle_is_valid_base(c.encode("latin1")))
