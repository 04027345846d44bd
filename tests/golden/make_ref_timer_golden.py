"""Generates tests/golden/ref_timer_*.csv and tests/golden/ref_params_probe.txt: the CSV bytes written by the REFERENCE's own Timer (src/timer.cpp, built unmodified
into oracle/_ref/ by `make -C oracle _ref`) for fixed durations (oracle/timer_probe.cpp).  Run in the build container, where
/root/reference exists; the fixtures are data (the reference's OUTPUT), and tests/test_ref_timer.py compares
include/timer_amd.hpp with them wherever oracle/_ref did not travel."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(3, 2, 0, 3), (4, 3, 1, 2), (2, 2, 0, 1), (5, 5, 4, 2)]     # (world, pcnt, p_gather, rounds)


def mpi_env(tmp):
    libdir = os.path.join(tmp, "mpilib")     # MPICH's own libraries only, not conda's old libstdc++
    os.makedirs(libdir, exist_ok=True)
    for lib in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0"):
        src = os.path.join("/opt/conda/lib", lib)
        if os.path.exists(src) and not os.path.exists(os.path.join(libdir, lib)):
            os.symlink(src, os.path.join(libdir, lib))
    return dict(os.environ, LD_LIBRARY_PATH=libdir)


def run_probe(exe, case, tmp):
    world, pcnt, p_gather, rounds = case
    csv = os.path.join(tmp, f"{os.path.basename(exe)}_{world}_{pcnt}_{p_gather}_{rounds}.csv")
    if os.path.exists(csv):
        os.remove(csv)
    subprocess.check_call(["/opt/conda/bin/mpiexec", "-n", str(world), exe, csv, str(pcnt), str(p_gather), str(rounds)], env=mpi_env(tmp), timeout=120)
    return open(csv, "rb").read()


if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref"])
    with tempfile.TemporaryDirectory() as tmp:
        for case in CASES:
            data = run_probe(os.path.join(ROOT, "oracle", "_ref", "timer_probe_ref"), case, tmp)
            name = os.path.join(ROOT, "tests", "golden", "ref_timer_%d_%d_%d_%d.csv" % case)
            open(name, "wb").write(data)
            print(name, len(data), "bytes")
        # the parameter structs: what a program compiled against the reference's include/params.hpp prints (oracle/params_probe.cpp)
        out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "params_probe_ref")])
        open(os.path.join(ROOT, "tests", "golden", "ref_params_probe.txt"), "wb").write(out)
        print("ref_params_probe.txt", len(out), "bytes")
