"""CPU: the C-ABI library loads, exports every symbol include/sshash_amd.h declares, and fails
loudly (never silently on a CPU path) when there is no GPU."""
from __future__ import annotations

import ctypes
import os
import re

import numpy as np
import pytest

import sshash_amd
from sshash_amd import _binding
from conftest import ROOT, has_gpu


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sshash_amd.h")).read()
    return sorted(set(re.findall(r"\b(sshash_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(sshash_amd.library_path())
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/sshash_amd.h but not exported"
    assert set(_binding.C_ABI_SYMBOLS) == set(names)


def test_only_c_types_in_signatures():
    text = open(os.path.join(ROOT, "include", "sshash_amd.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # signatures only, comments stripped
    assert "torch" not in code and "std::" not in code and "hipStream_t" not in code and "hip_runtime" not in code


def test_error_codes_and_messages(tmp_path, case_skew_regular):
    with pytest.raises(sshash_amd.SSHashError) as e:
        sshash_amd.Dictionary.load(str(tmp_path / "does_not_exist.sshash"))
    assert e.value.status == 2 and "error in opening the file" in str(e.value)  # src/query.cpp:128 wording
    junk = tmp_path / "junk.sshash"
    junk.write_bytes(b"not an index" * 10)
    with pytest.raises(sshash_amd.SSHashError) as e:
        sshash_amd.Dictionary.load(str(junk))
    assert e.value.status == 3
    # major version mismatch (include/util.hpp:191-195)
    raw = bytearray(open(case_skew_regular.index_path, "rb").read())
    raw[8] = 4
    bad = tmp_path / "old.sshash"
    bad.write_bytes(bytes(raw))
    with pytest.raises(sshash_amd.SSHashError) as e:
        sshash_amd.Dictionary.load(str(bad))
    assert e.value.status == 4 and "MAJOR index version mismatch" in str(e.value)
    # truncated file
    cut = tmp_path / "cut.sshash"
    cut.write_bytes(bytes(raw[: len(raw) // 2]).replace(b"\x04", b"\x05", 1))
    with pytest.raises(sshash_amd.SSHashError):
        sshash_amd.Dictionary.load(str(cut))
    with pytest.raises(sshash_amd.SSHashError) as e:
        sshash_amd.Dictionary.build(str(tmp_path / "nope.fa"))
    assert e.value.status == 2
    short = tmp_path / "short.fa"
    short.write_text(">\nACGT\n")
    with pytest.raises(sshash_amd.SSHashError) as e:
        sshash_amd.Dictionary.build(str(short), k=31, m=13)
    assert e.value.status == 7
    with pytest.raises(sshash_amd.SSHashError) as e:
        case_skew_regular.dict.access(case_skew_regular.gt.num_kmers)
    assert e.value.status == 1


def test_save_load_roundtrip_and_info(case_skew_canonical, tmp_path):
    d = case_skew_canonical.dict
    p = str(tmp_path / "again.sshash")
    d.save(p)
    assert open(p, "rb").read() == open(case_skew_canonical.index_path, "rb").read()
    d2 = sshash_amd.Dictionary.load(p)
    for f in ("k", "m", "canonical", "num_kmers", "num_strings", "num_bases", "num_minimizers", "num_bits", "vnum"):
        assert getattr(d, f)() == getattr(d2, f)()
    assert d2.vnum() == (5, 1, 1)
    ids = np.arange(0, d.num_kmers(), 97, dtype=np.uint64)
    assert (d.access_packed(ids) == d2.access_packed(ids)).all()


def test_fasta_last_line_without_newline_is_dropped(tmp_path):
    """Reference quirk kept: `if (is.eof()) break;` after reading the sequence line drops a final
    record that is not newline-terminated (src/builder/encode_strings.cpp:139-140)."""
    a, b = "ACGTTGCATGCATGCAACGTAGCTAGCTAGGATCGAT", "TTGACCAGTAGGGATACCCATGAGATTTACGGACAGT"
    f1 = tmp_path / "nl.fa"
    f1.write_text(f">\n{a}\n>\n{b}\n")
    f2 = tmp_path / "nonl.fa"
    f2.write_text(f">\n{a}\n>\n{b}")
    assert sshash_amd.Dictionary.build(str(f1), k=31, m=13).num_strings() == 2
    assert sshash_amd.Dictionary.build(str(f2), k=31, m=13).num_strings() == 1


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_lookup_without_gpu_fails_loudly(case_skew_regular):
    d = case_skew_regular.dict
    q = case_skew_regular.queries(4, 4)
    with pytest.raises(sshash_amd.SSHashError) as e:
        d.to_device(0)
    assert e.value.status == 5 and "no CPU fallback" in str(e.value)
    with pytest.raises(sshash_amd.SSHashError) as e:
        d.lookup(q)
    assert e.value.status == 5
    with pytest.raises(sshash_amd.SSHashError):
        d.is_member(q)
    with pytest.raises(sshash_amd.SSHashError):
        d.streaming_query(["ACGT" * 20])


def test_product_does_not_touch_the_oracle():
    """The product path must not include / import / link anything under oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "sshash_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "oracle" not in text.lower().replace("no pure-python", ""), f"{f} mentions the oracle"
    hdr = open(os.path.join(ROOT, "include", "sshash_amd.h")).read()
    assert "oracle" not in hdr.lower()


def test_table_key_function_is_strand_symmetric(tmp_path):
    """The super-k-mer table serves a k-mer and its reverse complement from one slot: its key function
    (csrc/device_layout.hpp, host+device code) must elect the same m-mer occurrence for both strands.
    tests/cpp/check_table_key.cpp checks that with g++ on the host, k <= 31 and k <= 63."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "check_table_key")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "check_table_key.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith("OK "), p.stdout + p.stderr


def test_lookup_output_buffer_is_checked_before_anything_runs(case_skew_regular):
    """Dictionary.lookup(out=...) hands the caller's array to the library (page-locked buffers are then used in place): a
    wrong dtype, size or layout is refused in the binding, before any call into the library."""
    d = case_skew_regular.dict
    q = case_skew_regular.queries(4, 4, seed=1)
    for bad in (np.empty(8, dtype=np.int64), np.empty(7, dtype=np.uint64), np.empty(16, dtype=np.uint64)[::2]):
        with pytest.raises(ValueError):
            d.lookup(q, out=bad)


def test_query_file_readers_hand_over_bounded_batches(tmp_path):
    """csrc/reads.cpp on the host: read_stream (what sshash_streaming_query_from_file feeds the device from, a bounded batch
    of whole reads at a time) yields exactly the reads of the whole file, in order, whatever the batch size -- FASTQ, one-line
    FASTA, multiline FASTA (src/query.cpp:9-108). tests/cpp/check_reads.cpp, plain g++ + zlib."""
    import gzip
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "check_reads")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "check_reads.cpp"),
                           os.path.join(ROOT, "sshash_amd", "csrc", "reads.cpp"), "-lz", "-lpthread", "-o", exe])
    golden = os.path.join(ROOT, "tests", "golden")
    multi = tmp_path / "multi.fa"
    with gzip.open(os.path.join(golden, "se.ust.k63.head.fa.gz"), "rt") as f:
        seqs = [line.strip() for line in f if not line.startswith(">")][:200]
    with open(multi, "w") as f:
        for i, s in enumerate(seqs):
            f.write(f">{i}\n{s[:70]}\n{s[70:]}\n\n")
    for path, multiline, k, reads in ((os.path.join(golden, "SRR5833294.10K.fastq.gz"), 0, 31, 10000),
                                      (os.path.join(golden, "salmonella_enterica_k31_ust.fa.gz"), 0, 31, None),
                                      (str(multi), 1, 63, len(seqs)), (str(multi), 0, 63, None)):
        p = subprocess.run([exe, path, str(multiline), str(k)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and p.stdout.startswith("OK "), (path, multiline, p.stdout + p.stderr)
        if reads is not None:
            assert int(p.stdout.split()[1]) == reads
    # BGZF (bgzip's format: members inflated on several threads, csrc/reads.cpp bgzf_source): the same reads as the plain file,
    # with one worker and with many, groups of members smaller than the file; a flipped byte is an error, not different reads
    from sshash_amd.synthetic import BGZF_EOF, bgzf_compress

    raw = gzip.open(os.path.join(golden, "SRR5833294.10K.fastq.gz"), "rb").read() * 12  # 120 000 reads, 30 MB: several groups
    plain, packed, broken = tmp_path / "big.fastq", tmp_path / "big_bgzf.fastq.gz", tmp_path / "broken.fastq.gz"
    plain.write_bytes(raw)
    blob = bgzf_compress(raw, 1) + BGZF_EOF
    packed.write_bytes(blob)
    assert gzip.open(packed, "rb").read() == raw  # what any gzip reader makes of it
    whole = subprocess.run([exe, str(plain), "0", "31"], capture_output=True, text=True, timeout=300)
    want = whole.stdout
    assert want.startswith("OK 120000 ")
    # (an uncompressed FASTQ also went through the piecewise reader, five piece sizes: the same reads, every chain of pieces complete)
    assert whole.stderr.count(" regular") == 5 and "irregular" not in whole.stderr, whole.stderr
    for threads in ("1", "7"):
        got = subprocess.run([exe, str(packed), "0", "31"], capture_output=True, text=True, timeout=300, env=dict(os.environ, SSHASH_AMD_READER_THREADS=threads))
        assert got.returncode == 0 and got.stdout == want, (threads, got.stdout, got.stderr)
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x55
    broken.write_bytes(bytes(bad))
    got = subprocess.run([exe, str(broken), "0", "31"], capture_output=True, text=True, timeout=300)
    assert got.returncode != 0 and "BGZF" in got.stderr
    # a group holding empty members only. (a) the end-of-file marker alone in a FRESH group: 129 stored members take
    # the first group just past its 8 MiB, the marker opens the second; (b) an empty bgzip file: the marker and nothing else;
    # (c) a trailer that claims more than a BGZF member can hold is a corrupt file, not an allocation of gigabytes
    import struct

    record = gzip.open(os.path.join(golden, "SRR5833294.10K.fastq.gz"), "rb").read()
    records = record.split(b"\n")
    one = b"\n".join(records[:4]) + b"\n"
    raw = (one * (129 * 65280 // len(one) + 1))[:129 * 65280]
    raw = raw[:raw.rfind(b"\n@") + 1]  # whole records
    raw += b"\n" * (129 * 65280 - len(raw))  # (blank lines behind the last record: up to exactly 129 members)
    stored = bgzf_compress(raw, 0)
    assert 128 * (len(stored) // 129) < (8 << 20) < len(stored)
    edge, edge_plain, empty, liar = tmp_path / "edge.fastq.gz", tmp_path / "edge.fastq", tmp_path / "empty.fastq.gz", tmp_path / "liar.fastq.gz"
    edge.write_bytes(stored + BGZF_EOF)
    edge_plain.write_bytes(raw)
    want = subprocess.run([exe, str(edge_plain), "0", "31"], capture_output=True, text=True, timeout=300)
    got = subprocess.run([exe, str(edge), "0", "31"], capture_output=True, text=True, timeout=300)
    assert want.returncode == 0 and got.returncode == 0 and got.stdout == want.stdout and want.stdout.startswith("OK "), (got.stdout, got.stderr)
    empty.write_bytes(BGZF_EOF)
    got = subprocess.run([exe, str(empty), "0", "31"], capture_output=True, text=True, timeout=60)
    assert got.returncode == 0 and got.stdout.startswith("OK 0 0 "), (got.stdout, got.stderr)
    lie = bytearray(bgzf_compress(one * 10, 1))
    lie[-4:] = struct.pack("<I", 0xF0000000)
    liar.write_bytes(bytes(lie) + BGZF_EOF)
    got = subprocess.run([exe, str(liar), "0", "31"], capture_output=True, text=True, timeout=60)
    assert got.returncode != 0 and "BGZF" in got.stderr, (got.stdout, got.stderr)
    # the piecewise reader on files that are not plain four-line records: qualities beginning with '@', CR LF, a last record cut short --
    # same reads as the sequential reader --, and five lines per record: flagged irregular (the library then reads sequentially)
    lines = record.split(b"\n")
    recs = [lines[i:i + 4] for i in range(0, 4000, 4)]
    variants = {"at": b"".join(b"\n".join([h, b, p, b"@" + q[1:]]) + b"\n" for h, b, p, q in recs),
                "crlf": b"".join(b"\r\n".join(r) + b"\r\n" for r in recs),
                "short": b"\n".join(b"\n".join(r) for r in recs)[:-80],
                "five": b"".join(b"\n".join([h, b[:40], b[40:], p, q]) + b"\n" for h, b, p, q in recs)}
    for name, blob in variants.items():
        f = tmp_path / f"{name}.fastq"
        f.write_bytes(blob)
        got = subprocess.run([exe, str(f), "0", "31"], capture_output=True, text=True, timeout=300)
        assert got.returncode == 0 and got.stdout.startswith("OK "), (name, got.stdout, got.stderr)
        if name == "five":
            assert "irregular" in got.stderr
        else:
            assert got.stderr.count(" regular") == 5 and "irregular" not in got.stderr, (name, got.stderr)
    txt = tmp_path / "reads.txt"
    txt.write_text("ACGT\n")
    p = subprocess.run([exe, str(txt), "0", "31"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and p.stdout.startswith("unsupported")


def test_repeated_kmer_in_a_heavy_bucket_fails_fast(tmp_path):
    """Every k-mer of the input must occur once (a spectrum-preserving string set). A k-mer repeated often enough
    to land in the skew index makes two MPHF keys equal; the builder must say so at once instead of searching
    pilots under seed after seed."""
    import time

    rng = np.random.default_rng(5)
    from conftest import random_dna

    core = random_dna(rng, 31)
    p = tmp_path / "dup.fa"
    with open(p, "w") as f:
        for i in range(200):
            f.write(">\n" + random_dna(rng, 12) + core + random_dna(rng, 12) + "\n")
    t0 = time.time()
    with pytest.raises(sshash_amd.SSHashError) as e:
        sshash_amd.Dictionary.build(str(p), k=31, m=13, num_threads=2)
    assert e.value.status == 7 and "occur once" in str(e.value)
    assert time.time() - t0 < 20


def test_string_size(case_skew_regular):
    case = case_skew_regular
    sizes = case.dict.string_size(range(len(case.sequences)))
    assert list(sizes) == [len(s) - case.k + 1 for s in case.sequences]
    assert int(sizes.sum()) == case.dict.num_kmers()
    with pytest.raises(sshash_amd.SSHashError):
        case.dict.string_size([len(case.sequences)])
    # string_offsets: [begin, end) in bases, strings back to back (include/dictionary.hpp:105-108)
    begin, end = case.dict.string_offsets(range(len(case.sequences)))
    assert int(begin[0]) == 0 and (begin[1:] == end[:-1]).all()
    assert list(end - begin) == [len(s) for s in case.sequences]
    with pytest.raises(sshash_amd.SSHashError):
        case.dict.string_offsets([len(case.sequences)])


def test_cuttlefish_segment_input(case_skew_regular, tmp_path):
    """'.cf_seg' inputs ("<id>\\t<sequence>" per line) are recognised by extension as the reference's builder does
    (src/builder/encode_strings.cpp:79-80,246-258): same dictionary as from the FASTA form."""
    import gzip

    case = case_skew_regular
    plain, zipped = tmp_path / "in.cf_seg", tmp_path / "in.cf_seg.gz"
    text = "".join(f"{i}\t{s}\n" for i, s in enumerate(case.sequences))
    plain.write_text(text)
    with gzip.open(zipped, "wt") as f:
        f.write(text)
    for p in (plain, zipped):
        d = sshash_amd.Dictionary.build(str(p), k=case.k, m=case.m, num_threads=2)
        assert (d.num_kmers(), d.num_strings(), d.num_minimizers()) == (case.dict.num_kmers(), case.dict.num_strings(),
                                                                       case.dict.num_minimizers())
        ids = np.arange(0, d.num_kmers(), 7, dtype=np.uint64)
        assert (d.access_packed(ids) == case.dict.access_packed(ids)).all()
    with pytest.raises(sshash_amd.SSHashError):
        sshash_amd.Dictionary.build(str(plain), k=case.k, m=case.m, weighted=True)


def test_corrupt_index_files_are_rejected_not_trusted(tmp_path):
    """a damaged or hostile index file must come back as SSHASH_ERR_FORMAT -- not as an out-of-bounds
    write while the device layout is made, nor as a kernel reading outside its arrays. Every 8-byte word of a small
    index file is overwritten in turn (with a huge and with a small value), and the file is truncated at several
    places: each variant either loads into a dictionary that passes the same structural checks, or raises."""
    import sshash_amd
    from conftest import random_dna

    rng = np.random.default_rng(3)
    seqs = [random_dna(rng, int(rng.integers(31, 90))) for _ in range(40)]
    fa = tmp_path / "small.fa"
    fa.write_text("".join(f">\n{s}\n" for s in seqs))
    d = sshash_amd.Dictionary.build(str(fa), k=31, m=11)
    good = tmp_path / "good.sshash"
    d.save(str(good))
    blob = bytearray(good.read_bytes())
    assert sshash_amd.Dictionary.load(str(good)).num_kmers() == d.num_kmers()
    rejected = accepted = 0
    bad = tmp_path / "bad.sshash"
    step = max(8, (len(blob) // 8 // 400) * 8)  # ~400 positions spread over the file, all of the header
    positions = list(range(0, min(len(blob), 256), 8)) + list(range(256, len(blob) - 8, step))
    for at in positions:
        for value in (0xFFFFFFFFFFFFFFF0, 0x0000000000000003):
            variant = bytearray(blob)
            variant[at:at + 8] = int(value).to_bytes(8, "little")
            bad.write_bytes(variant)
            try:
                other = sshash_amd.Dictionary.load(str(bad))
                accepted += 1  # a word of payload (bases, pilots ...) changed: structurally still a dictionary
                assert other.num_strings() == d.num_strings()
                other.close()
            except sshash_amd.SSHashError as e:
                rejected += 1
                assert e.status != 0
    for cut in (7, 40, 100, len(blob) // 2, len(blob) - 1):
        bad.write_bytes(blob[:cut])
        with pytest.raises(sshash_amd.SSHashError):
            sshash_amd.Dictionary.load(str(bad))
    bad.write_bytes(blob + b"x")
    with pytest.raises(sshash_amd.SSHashError):
        sshash_amd.Dictionary.load(str(bad))
    assert rejected > 50 and accepted > 0


def test_build_from_packed_rejects_bad_endpoints():
    import sshash_amd

    words = np.zeros(8, dtype=np.uint64)
    for endpoints in ([0, 40, 35, 100], [0, 40, 100000], [5, 40, 80], [0, 20, 60]):
        with pytest.raises((sshash_amd.SSHashError, ValueError)):
            sshash_amd.Dictionary.build_from_packed(words, np.array(endpoints, dtype=np.uint64), k=31, m=11)
