"""CPU test of the CLI's sequence reader (metamaps_amd/csrc/host/seq_reader.hpp, built with g++ from tests/test_seq_reader.cpp):
block-wise parsing of a memory-mapped file — what the CLI's parser threads do — gives the records of the sequential reader for
every block size and layout, and the sequential reader gives the records of the oracle's kseq restatement (oracle/orc_io.hpp)."""
import gzip
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    e = str(tmp_path_factory.mktemp("sr") / "t")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", e, os.path.join(HERE, "test_seq_reader.cpp"), "-lz"], check=True, timeout=300)
    return e


def _records(rng, n, empty_ok=True):
    out = []
    for i in range(n):
        L = rng.choice([0 if empty_ok else 1, 1, 7, 80, 81, 500, 3000, 12000]) + rng.randrange(1 - int(empty_ok), 40)
        out.append((f"r{i}/x", "".join(rng.choice("ACGTacgtNRY") for _ in range(L))))
    return out


def _write(path, recs, layout, rng):
    with open(path, "w") as f:
        for n, (name, s) in enumerate(recs):
            if layout == "fasta":
                f.write(f">{name} comment {n}\n" + "\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + ("\n" if s else ""))
                continue
            q = "".join(rng.choice("@>+I5?~!") for _ in s) if layout == "nasty" else "I" * len(s)
            if layout == "wrapped" and n % 3 == 0 and len(s) > 10:
                cut = [0] + sorted(rng.sample(range(1, len(s)), 3)) + [len(s)]
                f.write(f"@{name}\n" + "\n".join(s[a:b] for a, b in zip(cut[:-1], cut[1:])) + "\n+\n" + "\n".join(q[a:b] for a, b in zip(cut[:-1], cut[1:])) + "\n")
            else:
                f.write(f"@{name} c\n{s}\n+{name if n % 4 == 0 else ''}\n{q}\n")
        if layout == "no_final_newline":
            f.seek(f.tell() - 1); f.truncate()


@pytest.mark.parametrize("layout", ["plain", "wrapped", "nasty", "fasta", "no_final_newline", "truncated", "crlf"])
def test_block_parse_equals_sequential_parse(exe, tmp_path, layout):
    rng = random.Random(sum(map(ord, layout)))
    recs = _records(rng, 400, empty_ok=layout not in ("plain", "nasty", "fasta"))   # (a record without bases is not taken for a block start: sequential tail)
    p = str(tmp_path / "in.fq")
    _write(p, recs, "plain" if layout in ("truncated", "crlf") else layout, rng)
    if layout == "truncated":
        data = open(p, "rb").read(); open(p, "wb").write(data[:len(data) * 3 // 5])
    if layout == "crlf":
        data = open(p, "rb").read(); open(p, "wb").write(data.replace(b"\n", b"\r\n"))
    r = subprocess.run([exe, p], capture_output=True, timeout=300)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stdout.decode()
    if layout in ("plain", "nasty", "fasta"):                     # these layouts are recognised: the blocks really joined
        assert b"sequential tail" not in r.stdout.split(b"block 30000")[1].split(b"\n")[0]
    # the gzip of the same file through the zlib path gives the same records
    with open(p, "rb") as f, gzip.open(p + ".gz", "wb") as g:
        g.write(f.read())
    a = subprocess.run([exe, p, "dump"], capture_output=True, timeout=300).stdout
    b = subprocess.run([exe, p + ".gz", "dump"], capture_output=True, timeout=300).stdout
    assert a == b and (len(a.splitlines()) > 200 or layout == "truncated")


def test_sequential_reader_equals_oracle_reader(exe, tmp_path, oracle_lib):
    """same records (names, lengths) as the oracle's kseq restatement sees: the oracle CLI counts the reads of a FASTQ"""
    import json
    import orc
    rng = random.Random(5)
    recs = [(f"q{i}", "".join(rng.choice("ACGT") for _ in range(1200 + rng.randrange(0, 300)))) for i in range(60)]
    p = str(tmp_path / "r.fq")
    _write(p, recs, "wrapped", rng)
    mine = subprocess.run([exe, p, "dump"], capture_output=True, timeout=300).stdout.decode().splitlines()
    assert [l.split()[0] for l in mine] == [n for n, _ in recs] and [int(l.split()[1]) for l in mine] == [len(s) for _, s in recs]
    db = str(tmp_path / "db.fa")
    open(db, "w").write(">C0|kraken:taxid|1|x\n" + "".join(rng.choice("ACGT") for _ in range(5000)) + "\n")
    out = subprocess.run([orc.CLI, "mapDirectly", "--all", "-r", db, "-q", p, "-o", str(tmp_path / "o")], capture_output=True, check=True, timeout=300)
    assert json.loads(out.stderr.decode().strip().splitlines()[-1])["reads"] == len(recs)
