"""GPU parity, part 4: the record-level C ABI driven by an independent client.  tests/abi_client.c is written from
include/mkpileup.h alone (own BGZF/BAM and FASTA readers, own motif search, focus bytes and motif-id combos built as the header
documents them) and calls mkp_ctx_create -> mkp_set_caller -> mkp_shard_begin / mkp_shard_add_records / mkp_shard_run once per
reference interval, exactly the call pattern INTEGRATION.md §3 gives the Rust caller.  Its bedMethyl must equal
mkp_pileup_main's (the library's own driver) and the oracle's: a mismatch means the header under-specifies the seam."""
import os
import subprocess

import pytest

import modkit_amd
from pileup_cases import BC, REF, fixture
from test_gpu_scale import gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("abi") / "abi_client")
    lib_dir = os.path.join(ROOT, "modkit_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "abi_client.c"),
                           "-L", lib_dir, "-lmkpileup", "-lz", "-Wl,-rpath," + lib_dir])
    return exe


def run_client(client, bam, ref, out, mode, thr, interval=None, per_batch=None):
    p = subprocess.run([client, bam, ref or "-", out, mode, thr] + ([str(interval or 100000)] if (interval or per_batch) else []) + ([str(per_batch)] if per_batch else []), capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return p.stderr


@pytest.mark.parametrize("mode,flags", [
    ("all", ["--no-filtering"]),
    ("cpg", ["--no-filtering", "--cpg", "--ref", REF]),
    ("cg_cgcg", ["--no-filtering", "--motif", "CG", "0", "--motif", "CGCG", "2", "--ref", REF]),
], ids=["all_positions", "cpg", "two_motifs"])
@pytest.mark.parametrize("interval", [25, 100000])
def test_client_equals_driver_on_the_reference_fixture(client, tmp_path, mode, flags, interval):
    a, b = str(tmp_path / "client.bed"), str(tmp_path / "driver.bed")
    run_client(client, fixture(BC), REF, a, mode, "none", interval)
    modkit_amd.pileup([fixture(BC), b, "-i", str(interval)] + flags)
    assert open(a).read() == open(b).read() and open(a).read()


def test_client_per_interval_calls_on_a_cpg_workload(client, oracle_bin, tmp_path):
    # C3-shaped data at 1/16 scale, --cpg, a fixed pass threshold, the reference's default 100 kb intervals: ~40 calls of
    # mkp_shard_begin/add_records/run; rows must equal the driver's (one big shard) and the oracle's
    bam, fa, meta = gen(tmp_path, "c3i", [("chr20", 4_000_000)], 12_000, "hm", 21, ["--cpg-depleted", "--mean-len", "8353"])
    a, b, c = str(tmp_path / "client.bed"), str(tmp_path / "driver.bed"), str(tmp_path / "oracle.bed")
    err = run_client(client, bam, fa, a, "cpg", "0.7")
    flags = ["--filter-threshold", "0.7", "--cpg", "--ref", fa]
    modkit_amd.pileup([bam, b] + flags)
    p = subprocess.run([oracle_bin, "pileup", bam, c, "--oracle-workers", "8"] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    ta, tb, tc = open(a).read(), open(b).read(), open(c).read()
    assert len(ta.splitlines()) > 100_000 and ta == tb == tc
    assert "intervals=40 " in err
    print(err.strip())


@pytest.mark.parametrize("mode,flags", [
    ("all", ["--no-filtering"]),
    ("cpg", ["--no-filtering", "--cpg", "--ref", REF]),
    ("cg_cgcg", ["--no-filtering", "--motif", "CG", "0", "--motif", "CGCG", "2", "--ref", REF]),
], ids=["all_positions", "cpg", "two_motifs"])
@pytest.mark.parametrize("per_batch", [2, 7, 1000])
def test_batch_seam_equals_driver_on_the_reference_fixture(client, tmp_path, mode, flags, per_batch):
    """mkp_batch_run: `per_batch` consecutive 25 bp intervals (a MultiChromCoordinates) per call — merged into one resident shard
    inside, rows cut apart at the interval ends — must print what the per-interval calls and the driver print."""
    a, b = str(tmp_path / "client.bed"), str(tmp_path / "driver.bed")
    err = run_client(client, fixture(BC), REF, a, mode, "none", 25, per_batch)
    assert "per_batch=%d " % per_batch in err
    modkit_amd.pileup([fixture(BC), b, "-i", "25"] + flags)
    assert open(a).read() == open(b).read() and open(a).read()


def test_batch_seam_on_a_cpg_workload(client, oracle_bin, tmp_path):
    # the same C3-shaped data through mkp_batch_run, 12 intervals (1.2 Mb) per call: 4 calls instead of 40, the same rows
    bam, fa, meta = gen(tmp_path, "c3b", [("chr20", 4_000_000)], 12_000, "hm", 21, ["--cpg-depleted", "--mean-len", "8353"])
    a, c = str(tmp_path / "client.bed"), str(tmp_path / "oracle.bed")
    err = run_client(client, bam, fa, a, "cpg", "0.7", 100000, 12)
    flags = ["--filter-threshold", "0.7", "--cpg", "--ref", fa]
    p = subprocess.run([oracle_bin, "pileup", bam, c, "--oracle-workers", "8"] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert open(a).read() == open(c).read()
    assert "intervals=4 " in err
    print(err.strip())


@pytest.mark.parametrize("mode,flags", [
    ("all", ["--no-filtering"]),
    ("cpg", ["--no-filtering", "--cpg", "--ref", REF]),
], ids=["all_positions", "cpg"])
def test_file_seam_equals_driver_on_the_reference_fixture(client, tmp_path, mode, flags):
    """mkp_process_region: the client hands over the BAM's path and one shard per contig; with the fixture's .bai the records reach the
    kernels through the device ingest.  Same rows as the driver."""
    a, b = str(tmp_path / "client.bed"), str(tmp_path / "driver.bed")
    err = run_client(client, fixture(BC), REF, a, mode, "none", 100000, "file")
    assert "intervals=" in err
    modkit_amd.pileup([fixture(BC), b] + flags)
    assert open(a).read() == open(b).read() and open(a).read()


def test_file_seam_on_a_cpg_workload(client, oracle_bin, tmp_path):
    # the C3-shaped data of the batch test, one mkp_process_region for the contig (indexed BAM: device ingest) against the oracle
    bam, fa, meta = gen(tmp_path, "c3f", [("chr20", 4_000_000)], 12_000, "hm", 21, ["--cpg-depleted", "--mean-len", "8353"])
    a, c = str(tmp_path / "client.bed"), str(tmp_path / "oracle.bed")
    err = run_client(client, bam, fa, a, "cpg", "0.7", 100000, "file")
    flags = ["--filter-threshold", "0.7", "--cpg", "--ref", fa]
    p = subprocess.run([oracle_bin, "pileup", bam, c, "--oracle-workers", "8"] + flags, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert open(a).read() == open(c).read()
    assert "intervals=1 " in err
    print(err.strip())
