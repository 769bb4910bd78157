"""GPU parity of the batched streaming query (six counters of streaming_query_report)."""
from __future__ import annotations

import gzip

import numpy as np
import pytest

from conftest import FASTQ, random_dna

pytestmark = pytest.mark.gpu


def _as_dict(rep):
    return {k: getattr(rep, k) for k in ("num_kmers", "num_positive_kmers", "num_negative_kmers", "num_invalid_kmers",
                                         "num_searches", "num_extensions")}


def _synthetic_reads(case, n_reads, seed, read_len=120):
    """Half of the reads are sampled from the indexed strings (either strand, 1% substitutions,
    'N' at rate 1e-2), half are random."""
    rng = np.random.default_rng(seed)
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    long_seqs = [s for s in case.sequences if len(s) >= case.k + 5]
    for i in range(n_reads):
        if i % 2 == 0:
            s = long_seqs[int(rng.integers(0, len(long_seqs)))]
            a = int(rng.integers(0, max(1, len(s) - case.k)))
            r = list(s[a:a + read_len])
            for j in range(len(r)):
                u = rng.random()
                if u < 0.01:
                    r[j] = "ACGT"[int(rng.integers(0, 4))]
                elif u < 0.02:
                    r[j] = "N"
            r = "".join(r)
            if rng.random() < 0.5:
                r = r.translate(comp)[::-1]
            if rng.random() < 0.3:
                r = r.lower()
        else:
            r = random_dna(rng, int(rng.integers(1, read_len)))
        reads.append(r)
    reads += ["", "A" * (case.k - 1), "N" * (case.k + 3)]
    return reads


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_se_canonical", "case_skew_regular", "case_skew_canonical",
                                       "case_k63_canonical", "case_k63_regular", "case_small_k"])
def test_streaming_counters_match_oracle(case_name, request):
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    reads = _synthetic_reads(case, 3000, seed=17)
    got = _as_dict(d.streaming_query(reads))
    want = case.oracle.streaming_query(reads)
    assert got == want
    assert got["num_kmers"] == got["num_positive_kmers"] + got["num_negative_kmers"] + got["num_invalid_kmers"]
    assert got["num_positive_kmers"] == got["num_searches"] + got["num_extensions"]
    assert got["num_extensions"] > 0 and got["num_invalid_kmers"] > 0


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_se_canonical", "case_skew_regular", "case_k63_canonical", "case_k63_regular",
                                       "case_small_k", "case_m_equals_k"])
def test_streaming_counters_on_low_complexity_reads(case_name, request):
    """Reads that put equal election hashes into one window (two-letter reads, homopolymers, short tandem repeats): ties between the
    strands (complete path), keys that never change, newcomers that hash equal to the elected occurrence (sk_key_persists must stop there)."""
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    rng = np.random.default_rng(5)
    reads = _synthetic_reads(case, 2000, seed=23)
    reads += ["".join(rng.choice(list("AC"), size=200)) for _ in range(50)] + ["A" * 300, "ACGT" * 60, ("A" * 40 + "C" * 40) * 3]
    got = _as_dict(d.streaming_query(reads))
    assert got == case.oracle.streaming_query(reads)


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_k63_canonical"])
def test_streaming_counters_moved_out_of_their_32_bit_registers(case_name, request, monkeypatch):
    """Three of the run-based kernel's counters are 32 bits wide in the lanes and moved into 64-bit totals of the wave before they could
    wrap (every 2^16 turns); SSHASH_AMD_TEST_HOOKS stream_move_out_every=<n> makes that happen every n turns: the report must not change."""
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    reads = _synthetic_reads(case, 3000, seed=31)
    want = case.oracle.streaming_query(reads)
    assert _as_dict(d.streaming_query(reads)) == want
    for at in ("1", "7", "300"):
        monkeypatch.setenv("SSHASH_AMD_TEST_HOOKS", "stream_move_out_every=" + at)
        assert _as_dict(d.streaming_query(reads)) == want, at
    monkeypatch.delenv("SSHASH_AMD_TEST_HOOKS")


def test_streaming_runs_and_skips_against_hand_made_reads(case_se_regular, case_se_canonical, case_k63_regular):
    """What the run-based kernel decides without looking (streaming.hip): extension runs measured 32 bases a step, forward and
    backward, across word boundaries, up to a string's end and past it; the k-mers behind a miss counted by the slot's say. Reads cut
    out of the strings at chosen places, with ONE substitution at every possible distance from either end, long reads (several
    hundred extensions in a row), reads that run off their string's end into random bases, and N's next to the substitution."""
    comp = str.maketrans("ACGT", "TGCA")
    for case in (case_se_regular, case_se_canonical, case_k63_regular):
        d = case.dict.to_device(0)
        k = case.k
        rng = np.random.default_rng(11)
        long_seqs = sorted((s for s in case.sequences if len(s) >= 4 * k + 400), key=len)
        s = long_seqs[len(long_seqs) // 2]
        reads = []
        L = 2 * k + 40
        for where in range(0, L):  # one substitution at base `where` of a read that starts at an odd offset of its string
            r = list(s[37:37 + L])
            r[where] = "ACGT"[("ACGT".index(r[where]) + 1 + where % 3) % 4]
            reads.append("".join(r))
            reads.append("".join(r).translate(comp)[::-1])
        for a in (0, 1, 31, 32, 33, 63, 64, 65, 95, 96):  # runs that start at every alignment of the strings' words
            reads.append(s[a:a + 3 * k + 70])
            reads.append(s[a:a + 3 * k + 70].translate(comp)[::-1])
        reads.append(s)  # the whole string: len - k extensions behind one search
        reads.append(s.translate(comp)[::-1])
        tail = "".join(rng.choice(list("ACGT"), size=k + 20))
        reads.append(s[-(k + 50):] + tail)  # off the string's end: the run stops at the mark, the rest is looked up
        reads.append((s[-(k + 50):] + tail).translate(comp)[::-1])
        reads.append(tail + s[:k + 50])  # into the string's first k-mer from random bases
        for gap in (1, 2, k - 1, k, k + 1):  # an N and a substitution `gap` bases apart
            r = list(s[100:100 + 3 * k])
            r[k + 5] = "N"
            r[k + 5 + gap] = "ACGT"[("ACGT".index(r[k + 5 + gap]) + 2) % 4]
            reads.append("".join(r))
        got = _as_dict(d.streaming_query(reads))
        want = case.oracle.streaming_query(reads)
        assert got == want, (case.k, case.dict.canonical())
        assert got["num_extensions"] > 20 * got["num_searches"] > 0 and got["num_negative_kmers"] > 0 and got["num_invalid_kmers"] > 0
        # read by read as well: a wrong count must not cancel against another read's
        for i in range(0, len(reads), 7):
            assert _as_dict(d.streaming_query([reads[i]])) == case.oracle.streaming_query([reads[i]]), (case.k, i, reads[i])


def test_streaming_query_from_fastq_file(case_se_regular, case_se_canonical):
    """README.md:222-223 known answer num_kmers = 460000 for SRR5833294.10K at k=31; the other counters
    against the oracle, and num_positive against a brute-force set (test/check.cpp:61-98)."""
    with gzip.open(FASTQ, "rt") as f:
        lines = f.read().split("\n")
    reads = lines[1::4]
    reads = [r for r in reads if r]
    for case in (case_se_regular, case_se_canonical):
        d = case.dict.to_device(0)
        got = _as_dict(d.streaming_query_from_file(FASTQ))
        assert got["num_kmers"] == 460000
        assert got == case.oracle.streaming_query(reads)
        # independent count of positives
        pos = 0
        valid = set("ACGTacgt")
        kmers = []
        for r in reads:
            for i in range(len(r) - case.k + 1):
                x = r[i:i + case.k]
                if all(c in valid for c in x):
                    kmers.append(x)
        import sshash_amd

        arr = sshash_amd.encode_kmers(kmers, case.k)
        packed = np.zeros(len(kmers), dtype=np.uint64)
        codes = ((arr >> 1) & 3).astype(np.uint64)
        for j in range(case.k):
            packed |= codes[:, j] << np.uint64(2 * j)
        pos = int(case.gt.lookup(packed)["found"].sum())
        assert got["num_positive_kmers"] == pos


def test_multiline_fasta_and_unsupported_extension(case_skew_regular, tmp_path):
    case = case_skew_regular
    d = case.dict.to_device(0)
    s = case.sequences[0]
    p = tmp_path / "q.fa"
    # multiline: the header is NOT special-cased by the reference reader: its characters just make
    # k-mers invalid; an empty line ends a segment (src/query.cpp:9-47)
    p.write_text(">h\n" + s[:40] + "\n" + s[40:] + "\n\n>x\n" + s[:35] + "\n")
    got = _as_dict(d.streaming_query_from_file(str(p), multiline=True))
    want = case.oracle.streaming_query([">h" + s, ">x" + s[:35]])
    assert got == want
    # single-line FASTA reader: (header, sequence) line pairs -> (">h", s[:40]), (s[40:], ""), (">x", s[:35])
    single = _as_dict(d.streaming_query_from_file(str(p), multiline=False))
    assert single == case.oracle.streaming_query([s[:40], "", s[:35]])
    q = tmp_path / "q.txt"
    q.write_text("ACGT\n")
    rep = _as_dict(d.streaming_query_from_file(str(q)))
    assert rep == {k: 0 for k in rep}  # "unsupported query file format": empty report (src/query.cpp:169-171)


def test_query_file_goes_through_in_batches(case_se_regular, case_skew_regular, tmp_path, monkeypatch):
    """The file readers hand over a bounded number of bases at a time (whole reads; a reader thread fills the next batch
    while the device works): the report does not depend on where the batches end -- FASTQ, one-line and multiline FASTA."""
    d = case_se_regular.dict.to_device(0)
    whole = _as_dict(d.streaming_query_from_file(FASTQ))
    for batch in ("1", "1000", "77777"):
        monkeypatch.setenv("SSHASH_AMD_TEST_HOOKS", "query_batch_bases=" + batch)
        assert _as_dict(d.streaming_query_from_file(FASTQ)) == whole
    monkeypatch.delenv("SSHASH_AMD_TEST_HOOKS")
    case = case_skew_regular
    d = case.dict.to_device(0)
    p = tmp_path / "many.fa"
    with open(p, "w") as f:
        for i, s in enumerate(case.sequences):
            f.write(f">{i}\n{s[:60]}\n{s[60:]}\n\n")
    for multiline in (True, False):
        whole = _as_dict(d.streaming_query_from_file(str(p), multiline=multiline))
        assert whole["num_kmers"] > 0
        for batch in ("1", "500"):
            monkeypatch.setenv("SSHASH_AMD_TEST_HOOKS", "query_batch_bases=" + batch)
            assert _as_dict(d.streaming_query_from_file(str(p), multiline=multiline)) == whole
        monkeypatch.delenv("SSHASH_AMD_TEST_HOOKS")
    with pytest.raises(Exception):
        d.streaming_query_from_file(str(tmp_path / "missing.fq"))


def test_plain_fastq_is_read_in_pieces_by_all_lanes(case_se_regular, tmp_path, monkeypatch):
    """An uncompressed FASTQ is cut at fixed byte positions and parsed by the lanes themselves (csrc/reads.cpp: fastq_pieces,
    streaming_query_fastq_pieces): the report is the sequential reader's whatever the piece size and the number of lanes -- also
    with quality lines that begin with '@', CR LF line ends and a last record cut short --, and a file that is not four lines
    per record falls back to the sequential reader instead of giving another answer."""
    d = case_se_regular.dict.to_device(0)
    raw = gzip.open(FASTQ, "rb").read()
    want = _as_dict(d.streaming_query_from_file(FASTQ))  # (gzip: the sequential reader)
    lines = raw.split(b"\n")
    assert lines[-1] == b"" and (len(lines) - 1) % 4 == 0
    records = [lines[i:i + 4] for i in range(0, len(lines) - 1, 4)]
    at_quals = b"".join(b"\n".join([h, b, p, b"@" + q[1:]]) + b"\n" for h, b, p, q in records)
    crlf = b"".join(b"\r\n".join([h, b, p, q]) + b"\r\n" for h, b, p, q in records[:2000])
    cut_short = raw[:raw.rfind(b"\n+")]  # the last record: header and bases, no newline behind them
    five_lines = b"".join(b"\n".join([h, b[:40], b[40:], p, q]) + b"\n" for h, b, p, q in records[:3000])
    files = {"plain": raw * 6, "at_quals": at_quals, "crlf": crlf, "cut_short": cut_short, "five_lines": five_lines}
    for name, blob in files.items():
        path = tmp_path / f"{name}.fastq"
        path.write_bytes(blob)
        monkeypatch.setenv("SSHASH_AMD_TEST_HOOKS", "sequential_reader=1")
        sequential = _as_dict(d.streaming_query_from_file(str(path)))
        monkeypatch.delenv("SSHASH_AMD_TEST_HOOKS")
        if name == "plain":
            assert sequential == {f: 6 * v for f, v in want.items()}
        if name in ("at_quals", "cut_short"):
            assert sequential == want
        for piece, lanes in ((None, None), ("4096", "3"), ("100003", "1"), ("1000000", "16")):
            for var, value in (("SSHASH_AMD_TEST_HOOKS", piece and "fastq_piece_bytes=" + piece), ("SSHASH_AMD_READER_THREADS", lanes)):
                if value is None:
                    monkeypatch.delenv(var, raising=False)
                else:
                    monkeypatch.setenv(var, value)
            assert _as_dict(d.streaming_query_from_file(str(path))) == sequential, (name, piece, lanes)
        monkeypatch.delenv("SSHASH_AMD_TEST_HOOKS", raising=False)
        monkeypatch.delenv("SSHASH_AMD_READER_THREADS", raising=False)


def test_query_file_in_bgzf_members(case_se_regular, tmp_path):
    """A BGZF file (bgzip: gzip members with their size in the header, inflated on several threads by csrc/reads.cpp) gives the
    report of the plain gzip file it was made from -- also when the file spans several groups of members and batches --; a
    corrupt member is an error through the C ABI, not a different report."""
    import sshash_amd
    from sshash_amd.synthetic import BGZF_EOF, bgzf_compress

    d = case_se_regular.dict.to_device(0)
    raw = gzip.open(FASTQ, "rb").read()
    want = _as_dict(d.streaming_query_from_file(FASTQ))
    one = tmp_path / "one.fastq.gz"
    one.write_bytes(bgzf_compress(raw) + BGZF_EOF)
    assert _as_dict(d.streaming_query_from_file(str(one))) == want
    many = tmp_path / "many.fastq.gz"
    many.write_bytes(bgzf_compress(raw * 40, 1) + BGZF_EOF)  # 100 MB of FASTQ: several groups of members
    got = _as_dict(d.streaming_query_from_file(str(many)))
    assert got == {f: 40 * v for f, v in want.items()}
    bad = bytearray(many.read_bytes())
    bad[len(bad) // 3] ^= 0x10
    broken = tmp_path / "broken.fastq.gz"
    broken.write_bytes(bytes(bad))
    with pytest.raises(sshash_amd.SSHashError):
        d.streaming_query_from_file(str(broken))


# ---- per-k-mer results: streaming_query::lookup for every k-mer (include/streaming_query.hpp:56-109) ----------------

FIELDS = ("kmer_id", "kmer_id_in_string", "string_id", "string_begin", "string_end", "kmer_orientation")  # what
# equal_lookup_result compares (include/util.hpp:107-141)


@pytest.mark.parametrize("case_name", ["case_se_regular", "case_se_canonical", "case_skew_regular", "case_skew_canonical",
                                       "case_k63_canonical", "case_k63_regular", "case_small_k"])
def test_streaming_lookup_returns_what_the_reference_returns_kmer_by_kmer(case_name, request):
    """Every k-mer of every read: the result streaming_query::lookup hands back (oracle_streaming_read: the restated state
    machine), the counters of the report, and -- the reference's own assertion, :107 -- the point lookup."""
    case = request.getfixturevalue(case_name)
    d = case.dict.to_device(0)
    reads = _synthetic_reads(case, 400, seed=23)
    # a long read: several strings back to back with junk in between (one lane per read would crawl along it)
    rng = np.random.default_rng(5)
    reads.append("NN".join(case.sequences[i] for i in rng.integers(0, len(case.sequences), 40)))
    per_read, report = d.streaming_lookup(reads, full=True)
    assert _as_dict(report) == case.oracle.streaming_query(reads) == _as_dict(d.streaming_query(reads))
    assert len(per_read) == len(reads)
    for r, got in zip(reads, per_read):
        want = case.oracle.streaming_read(r)
        assert got.kmer_id.size == want.size == max(0, len(r) - case.k + 1)
        found = want["kmer_id"] != np.uint64(0xFFFFFFFFFFFFFFFF)
        for f in FIELDS:
            g = getattr(got, f)
            if f == "kmer_orientation":  # compared for found k-mers only (include/util.hpp:122-127)
                assert (g[found] == want[f][found]).all(), f
            else:
                assert (g == want[f]).all(), f


def test_streaming_lookup_device_leaves_other_places_untouched(case_se_regular):
    import torch

    case = case_se_regular
    d = case.dict.to_device(0)
    k = case.k
    reads = [case.sequences[0][:200], "ACGT", case.sequences[1][:k], case.sequences[2][:k + 7].lower()]
    blob = "".join(reads).encode()
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    dev = torch.device("cuda", 0)
    d_bases = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    ids = torch.full((len(blob),), 12345, dtype=torch.int64, device=dev)
    rep = torch.zeros(6, dtype=torch.int64, device=dev)
    d.streaming_lookup_device(0, d_bases.data_ptr(), d_off.data_ptr(), len(reads), len(blob), ids.data_ptr(), d_report=rep.data_ptr())
    torch.cuda.synchronize()
    got = ids.cpu().numpy()
    expect_kmers = 0
    for i, r in enumerate(reads):
        lo, n = int(offsets[i]), max(0, len(r) - k + 1)
        expect_kmers += n
        want = case.oracle.streaming_read(r)["kmer_id"].view(np.int64)
        assert (got[lo:lo + n] == want).all()
        assert (got[lo + n:int(offsets[i + 1])] == 12345).all()  # no k-mer starts there
    assert int(rep[0].item()) == expect_kmers and int(rep[1].item()) == expect_kmers  # all positive


def test_counters_of_reads_too_long_for_one_lane(case_se_regular):
    """a contig-sized read used to run on ONE lane (one dependent HBM read per k-mer, for minutes). Pieces
    holding a read of more than 2^16 bases now go through the position-parallel pipeline: same six counters."""
    case = case_se_regular
    d = case.dict.to_device(0)
    rng = np.random.default_rng(8)
    contig = "N".join(case.sequences[i] for i in rng.integers(0, len(case.sequences), 60))  # ~450 kbases
    assert len(contig) > (1 << 16)
    reads = [contig, case.sequences[3][:100], random_dna(rng, 80), contig[1000:200000].lower()]
    got = _as_dict(d.streaming_query(reads))
    assert got == case.oracle.streaming_query(reads)
    assert got["num_extensions"] > 100000 and got["num_invalid_kmers"] > 0


def test_streaming_lookup_host_keeps_every_array_at_places_without_a_kmer(case_se_regular):
    """sshash_streaming_lookup copies every requested array back whole; at the places where no k-mer starts
    (the last k - 1 bases of a read, reads shorter than k) the caller's values must survive for ALL fields, not for kmer_id only."""
    import ctypes as C

    from sshash_amd import _binding as B

    case = case_se_regular
    d = case.dict.to_device(0)
    k = case.k
    reads = [case.sequences[0][:150], "ACGTACGT", case.sequences[1][:k + 3], case.sequences[2][:90].lower()]
    blob = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    total = int(offsets[-1])
    sentinel64, sentinel8 = np.uint64(0x1234567890ABCDEF), np.int8(77)
    arrays = {f: np.full(total, sentinel64, dtype=np.uint64) for f in ("kmer_id", "kmer_id_in_string", "kmer_offset", "string_id", "string_begin", "string_end")}
    arrays["kmer_orientation"] = np.full(total, sentinel8, dtype=np.int8)
    r = B._Results()
    for f, a in arrays.items():
        setattr(r, f, a.ctypes.data)
    rep = B._Report()
    B._check(B._load().sshash_streaming_lookup(d._h, blob.ctypes.data, offsets.ctypes.data, len(reads), C.byref(r), C.byref(rep)))
    places = np.zeros(total, dtype=bool)
    for i, read in enumerate(reads):
        lo, n = int(offsets[i]), max(0, len(read) - k + 1)
        places[lo:lo + n] = True
        want = case.oracle.streaming_read(read)
        for f in ("kmer_id", "kmer_id_in_string", "string_id", "string_begin", "string_end"):  # (equal_lookup_result's fields, as above)
            assert (arrays[f][lo:lo + n] == want[f]).all(), f
        found = want["kmer_id"] != np.uint64(0xFFFFFFFFFFFFFFFF)
        assert (arrays["kmer_offset"][lo:lo + n][found] == (want["string_begin"] + want["kmer_id_in_string"])[found]).all()
    assert places.sum() == rep.num_kmers and (~places).sum() > 0
    for f, a in arrays.items():
        keep = sentinel8 if f == "kmer_orientation" else sentinel64
        assert (a[~places] == keep).all(), f"{f}: a place without a k-mer was overwritten"


def test_streaming_lookup_device_ignores_bases_behind_the_last_read(case_se_regular):
    """total_bases larger than read_offsets[num_reads] -- the tail belongs to no read: no k-mer is looked up there,
    nothing is counted, the caller's values stay."""
    import torch

    case = case_se_regular
    d = case.dict.to_device(0)
    k = case.k
    reads = [case.sequences[0][:120], case.sequences[1][:80]]
    tail = case.sequences[2][:300]  # perfectly good bases, but behind the last read's end
    blob = ("".join(reads) + tail).encode()
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    dev = torch.device("cuda", 0)
    d_bases = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    ids = torch.full((len(blob),), 4242, dtype=torch.int64, device=dev)
    rep = torch.zeros(6, dtype=torch.int64, device=dev)
    d.streaming_lookup_device(0, d_bases.data_ptr(), d_off.data_ptr(), len(reads), len(blob), ids.data_ptr(), d_report=rep.data_ptr())
    torch.cuda.synchronize()
    got = ids.cpu().numpy()
    expect = sum(len(r) - k + 1 for r in reads)
    assert int(rep[0].item()) == expect
    assert (got[int(offsets[-1]):] == 4242).all()
    assert (got[int(offsets[1]) - (k - 1):int(offsets[1])] == 4242).all()
