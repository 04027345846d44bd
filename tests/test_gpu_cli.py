"""The reference-style test driver (distributedfft_amd/cli.py): testcases 0-4 with the reference's
flags, its printed error norms and its timer CSV format (src/timer.cpp:58-101)."""
import math
import os

import pytest

from oracle import oracle as orc  # noqa: E402  (testcase 4: what the reference's single-precision root makes the test print)

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

from distributedfft_amd import cli  # noqa: E402


def test_testcase3_round_trip_pencil(tmp_path, capsys):
    r = cli.run(["pencil", "-nx", "64", "-ny", "64", "-nz", "64", "-p1", "2", "-p2", "4", "-o", "1", "-t", "3", "-i", "2",
                 "-w", "1", "-d", "-c", "-b", str(tmp_path)])
    out = capsys.readouterr().out
    assert out.count("Result (avg):") == 3 and out.count("Result (max):") == 3
    # absolute error of inv - N^3 * in on values up to 255 * 64^3 = 6.7e7
    assert r["max"] < 1e-6 and r["avg"] < 1e-7
    path = r["csv"]
    assert os.path.basename(path) == "test_1_1_0_1_0_64_64_64_1_2_4.csv" and os.path.dirname(path).endswith("pencil")
    lines = open(path).read().split("\n")
    assert lines[0] == "," + "".join(f"{i}," for i in range(8))
    body = [ln for ln in lines[1:] if ln]
    assert body[0].startswith("init,")
    names = [ln.split(",")[0] for ln in body]
    for sec in ("1D FFT Z-Direction", "First Transpose (Finished All2All)", "1D FFT Y-Direction",
                "Second Transpose (Finished All2All)", "1D FFT X-Direction", "Run complete"):
        assert sec in names
    row = [ln for ln in body if ln.startswith("Run complete")][0].split(",")
    assert len(row) == 8 + 2 and all(float(v) > 0 for v in row[1:9])
    # warm-up iteration is not stored: 2 iterations x (forward + inverse) blocks
    assert names.count("Run complete") == 4


def test_testcase4_laplacian_slab(tmp_path):
    r = cli.run(["slab", "-nx", "32", "-ny", "32", "-nz", "32", "-p", "4", "-t", "4", "-d", "-b", str(tmp_path)])
    # the reference's multiplier divides by sqrtf(N^3) in single precision: 3 |N^3/sqrtf(N^3) - sqrt(N^3)| x mean / max |u| is printed
    cavg, cmax = orc.testcase4_printed((32, 32, 32))
    assert abs(r["max"] - cmax) < 1e-9 and abs(r["avg"] - cavg) < 1e-10, (r, cavg, cmax)


def test_testcases_with_the_x_contiguous_spectrum(tmp_path, capsys):
    """--spectral-layout 1 (extension): the testcases index the spectrum block through the plan's strides -- testcase 1 (distributed ==
    single device), 3 (round trip) and 4 (Laplacian: forward, pointwise multiplication, inverse) on pencil 2 x 2 and slab 3"""
    from distributedfft_amd import cli
    base = ["-nx", "32", "-ny", "24", "-nz", "40", "-o", "1", "-d", "-b", str(tmp_path), "--spectral-layout", "1"]
    r1 = cli.run(["pencil", "-p1", "2", "-p2", "2", "-t", "1"] + base)
    assert r1["sum"] < 1e-5      # sum over 30720 points of |distributed - single device| on values up to 255 * 30720
    r3 = cli.run(["pencil", "-p1", "2", "-p2", "2", "-t", "3", "-i", "2"] + base)
    assert r3["max"] < 1e-6 and r3["avg"] < 1e-7
    r4 = cli.run(["slab", "-p", "3", "-t", "4"] + base)
    cavg, cmax = orc.testcase4_printed((32, 24, 40))
    assert abs(r4["max"] - cmax) < 1e-9 and abs(r4["avg"] - cavg) < 1e-10, (r4, cavg, cmax)
    capsys.readouterr()


def test_testcase1_distributed_vs_single_and_benchmarks(tmp_path, capsys):
    r = cli.run(["pencil", "-nx", "32", "-ny", "64", "-nz", "16", "-p1", "3", "-p2", "2", "-t", "1", "-d", "-b", str(tmp_path)])
    assert r["sum"] < 1e-6
    for t in ("0", "2"):
        r = cli.run(["pencil", "-nx", "64", "-ny", "64", "-nz", "64", "-p1", "2", "-p2", "2", "-t", t, "-i", "3", "-b", str(tmp_path)])
        assert os.path.exists(r["csv"])
    r = cli.run(["pencil", "-nx", "64", "-ny", "64", "-nz", "64", "-p1", "2", "-p2", "2", "-t", "0", "-f", "2", "-d", "--complex",
                 "-b", str(tmp_path)])
    assert os.path.exists(r["csv"])


def test_slab_sequence_z_then_yx_testcases(tmp_path):
    """`slab -s Z_Then_YX` (tests/src/slab/main.cpp:193-201): testcases 1, 3 and 4 on the z-split output"""
    base = ["slab", "-nx", "32", "-ny", "16", "-nz", "64", "-p", "3", "-s", "Z_Then_YX", "-d", "-b", str(tmp_path)]
    assert cli.run(base + ["-t", "1"])["sum"] < 1e-6
    r = cli.run(base + ["-t", "3", "-o", "1"])
    assert r["max"] < 1e-6
    assert os.path.basename(r["csv"]) == "test_1_1_0_32_16_64_0_3.csv" and os.path.dirname(r["csv"]).endswith("slab_z_then_yx")
    r = cli.run(["slab", "-nx", "32", "-ny", "32", "-nz", "32", "-p", "4", "-s", "Z_Then_YX", "-t", "4", "-d", "-b", str(tmp_path)])
    cavg, cmax = orc.testcase4_printed((32, 32, 32))
    assert abs(r["max"] - cmax) < 1e-9 and abs(r["avg"] - cavg) < 1e-10, (r, cavg, cmax)
    # Y_Then_ZX: forward only (testcases 0, 1)
    assert cli.run(base + ["-s", "Y_Then_ZX", "-t", "1"])["sum"] < 1e-6
    r = cli.run(base + ["-s", "Y_Then_ZX", "-t", "0", "-i", "2"])
    assert os.path.dirname(r["csv"]).endswith("slab_y_then_zx")
    with pytest.raises(SystemExit):
        cli.run(base + ["-s", "Y_Then_ZX", "-t", "3"])


@pytest.mark.parametrize("mode,extra,nproc", [("pencil", ["-p1", "2", "-p2", "2", "-o", "1", "-t", "3", "-i", "2", "-w", "1"], 4),
                                              ("slab", ["-p", "3", "-t", "4"], 3),
                                              ("pencil", ["-p1", "3", "-p2", "2", "-t", "1"], 6),
                                              ("pencil", ["-p1", "2", "-p2", "2", "-t", "1", "--complex"], 4)])
def test_one_process_per_rank_under_torch_distributed_run(tmp_path, mode, extra, nproc):
    """the reference's `mpiexec -n P ./pencil ...` (tests/src/pencil/main.cpp:194-229): P processes, one rank each, here
    sharing the single GPU of the test box over gloo; error norms reduced over ranks, rank 0 writes the CSV"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + nproc + (20 if "--complex" in extra else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "distributedfft_amd.cli", mode, "-nx", "48", "-ny", "32", "-nz", "40", "-d",
           "-b", str(tmp_path), "--backend", "gloo"] + extra
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=root))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("cli result:")]
    assert len(line) == 1, out.stdout
    res = eval(line[0][len("cli result:"):])       # noqa: S307  (our own repr of a dict of floats and one path)
    if "-t" in extra and extra[extra.index("-t") + 1] == "1":
        assert res["sum"] < 1e-6
    elif "-t" in extra and extra[extra.index("-t") + 1] == "4":
        cavg, cmax = orc.testcase4_printed((48, 32, 40))
        assert abs(res["max"] - cmax) < 1e-9 and abs(res["avg"] - cavg) < 1e-10, (res, cavg, cmax)
    else:
        assert res["max"] < 1e-6
    hdr = open(res["csv"]).read().split("\n")[0]
    assert hdr == "," + "".join(f"{i}," for i in range(nproc))
