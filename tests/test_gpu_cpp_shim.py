"""The C++ binding of INTEGRATION.md on hardware: a reference-style caller built against
include/mpicufft_amd.hpp + libdfft_amd.so (g++, MPICH from /opt/conda, HIP runtime)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPI_INC, MPI_LIB = "/opt/conda/include", "/opt/conda/lib/libmpi.so.12"


needs_mpich = pytest.mark.skipif(not (os.path.exists(MPI_LIB) and os.path.exists(os.path.join(MPI_INC, "mpi.h"))),
                                 reason="MPICH from the image is not present")


def build(tmp_path, source):
    exe = tmp_path / source.replace(".cpp", "")
    libdir = tmp_path / "mpilib"     # only MPICH's own libraries, not conda's old libstdc++
    libdir.mkdir(exist_ok=True)
    for lib in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0"):
        src = os.path.join("/opt/conda/lib", lib)
        if os.path.exists(src) and not os.path.exists(libdir / lib):
            os.symlink(src, libdir / lib)
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", MPI_INC, "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "cpp", source), "-o", str(exe),
           os.path.join(ROOT, "distributedfft_amd", "libdfft_amd.so"), MPI_LIB, "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "distributedfft_amd"), "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe, dict(os.environ, LD_LIBRARY_PATH=f"/opt/rocm/lib:{libdir}")


@needs_mpich
def test_cpp_shim_round_trip(tmp_path):
    exe, env = build(tmp_path, "shim_roundtrip.cpp")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "out size 64 32 25" in out.stdout and "Result (max):" in out.stdout


@needs_mpich
@pytest.mark.parametrize("P1,P2,mode", [(2, 1, ""), (2, 2, ""), (3, 2, ""), (3, 1, "zyx")])
def test_cpp_shim_mpi_ranks_sharing_the_gpu(tmp_path, P1, P2, mode):
    """mpiexec -n P: real MPI ranks (one process each) share GPU 0 and exchange through the shim's
    host-staged MPI transport = the reference's cuda_aware = false path"""
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec")
    exe, env = build(tmp_path, "shim_mpi_multirank.cpp")
    out = subprocess.run([mpiexec, "-n", str(P1 * P2), str(exe), str(P1), str(P2)] + ([mode] if mode else []), env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"ranks {P1 * P2} grid {P1}x{P2}" in out.stdout


@needs_mpich
@pytest.mark.parametrize("P1,P2,relay", [(2, 2, 3), (3, 2, 1), (2, 3, 3)])
def test_cpp_shim_mpi_ranks_with_the_relay(tmp_path, P1, P2, relay):
    """the same through the two-hop relay (DFFT_RELAY in the shim -> dfft_comm_set_option "relay"): the host-staged MPI transport
    then runs world-wide all-to-alls of message parts; round trip and DC term as before"""
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec")
    exe, env = build(tmp_path, "shim_mpi_multirank.cpp")
    out = subprocess.run([mpiexec, "-n", str(P1 * P2), str(exe), str(P1), str(P2)], env=dict(env, DFFT_RELAY=str(relay)), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"ranks {P1 * P2} grid {P1}x{P2}" in out.stdout


@needs_mpich
@pytest.mark.parametrize("kind,opt,P1,P2,nranks,fft_ranks", [
    ("pencil", 1, 3, 2, 6, 6), ("pencil", 0, 2, 2, 4, 4), ("slab", 1, 5, 1, 5, 5), ("slab", 0, 2, 1, 2, 2),
    ("pencil", 1, 1, 1, 1, 1),
    # max_world_size < communicator size: one extra rank stays outside, as the reference's coordinator does
    ("pencil", 1, 2, 1, 3, 2), ("slab", 1, 2, 1, 3, 2)])
def test_reference_call_sites_compile_and_run(tmp_path, kind, opt, P1, P2, nranks, fft_ranks):
    """tests/cpp/ref_caller.cpp holds the bodies of the reference's own callers
    (tests/src/pencil/random_dist_3D.cu:154-227, tests/src/slab/random_dist_default.cu:155-226) with only the
    include lines and cuda* -> hip* changed: MPIcuFFT_Pencil<T>* / MPIcuFFT_Slab<T>* receive the Opt1 or opt0
    object, getPartitionDimensions sizes `out`, initFFT(&global_size, true) for slabs; fp64 and fp32."""
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec")
    exe, env = build(tmp_path, "ref_caller.cpp")
    out = subprocess.run([mpiexec, "-n", str(nranks), str(exe), kind, str(opt), str(P1), str(P2), str(fft_ranks)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("tables ok") == 2 and "MISMATCH" not in out.stdout, out.stdout
