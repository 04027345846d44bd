"""Generates tests/golden/launch_commands.txt: the `mpiexec -n R ... pencil|slab ...` command lines that the REFERENCE's own launcher
(/root/reference/launch.py, imported here and run through its main()) builds from its own job files (jobs/argon/{pencil,slab}/*.json)
for the invocations its slurm scripts use (jobs/argon/*/slurm_scripts).  launch.py would execute each line with
subprocess.check_output; that one call is replaced by a recorder.  The fixture is the reference's OUTPUT (row f3 of SURVEY section 8:
"lets launch.py job JSONs run against the new library"): tests/test_launch_commands.py feeds every line to the argument parsers of
tools/drivers/{pencil,slab}.cpp and distributedfft_amd/cli.py.  Run in the build container, where /root/reference exists."""
import contextlib
import importlib.util
import io
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
# (job files, --global_params) as in jobs/argon/{slab,pencil}/slurm_scripts and the 8-GPU runs of jobs/bwunicluster
INVOCATIONS = [
    (["argon/slab/benchmarks_base.json", "argon/slab/validation.json"], "-p 4 -b ../benchmarks/argon/forward"),
    (["argon/slab/benchmarks_base.json", "argon/slab/validation.json"], "-p 4 -b ../benchmarks/argon/forward -s Z_Then_YX --opt 1"),
    (["argon/slab/benchmarks_base.json"], "-t 2 -p 4 -b ../benchmarks/argon/inverse -s Y_Then_ZX"),
    (["argon/pencil/benchmarks_base.json", "argon/pencil/validation.json"], "-p1 2 -p2 2 -b ../benchmarks/argon/forward --opt 1"),
    (["argon/pencil/benchmarks_base.json"], "-t 2 -p1 2 -p2 2 -b ../benchmarks/argon/inverse"),
    (["bwunicluster/pencil/benchmarks_base.json"], "-c -t 2 -p1 2 -p2 4 -b ../benchmarks/bwunicluster/gpu8/large/inverse --opt 1"),
]


def commands():
    spec = importlib.util.spec_from_file_location("reference_launch", os.path.join(REF, "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    seen = []
    real_check_output = launch.subprocess.check_output      # (launch.subprocess IS the subprocess module: put it back afterwards)
    launch.subprocess.check_output = lambda command, shell=True: (seen.append(command), b"")[1]      # the ONE call that would execute it
    cwd, argv = os.getcwd(), sys.argv
    try:
        for jobs, params in INVOCATIONS:
            with tempfile.TemporaryDirectory() as tmp:
                os.symlink(os.path.join(REF, "jobs"), os.path.join(tmp, "jobs"))
                os.makedirs(os.path.join(tmp, "build"))
                sys.argv = ["launch.py", "--jobs"] + jobs + ["--build_dir", os.path.join(tmp, "build"), "--global_params", params]
                with contextlib.redirect_stdout(io.StringIO()):
                    launch.main()
    finally:
        launch.subprocess.check_output = real_check_output
        os.chdir(cwd)
        sys.argv = argv
    return seen


if __name__ == "__main__":
    cmds = commands()
    # every flag pattern, at three of the job's sizes (the smallest cube, an uneven grid, the largest cube): the sizes are data of the job
    # files, the flag patterns are what the parsers must understand
    keep = ("-nx 128 -ny 128 -nz 128", "-nx 128 -ny 128 -nz 256", "-nx 1024 -ny 1024 -nz 1024")
    uniq = [c for c in dict.fromkeys(cmds) if c.endswith(keep)]
    out = os.path.join(ROOT, "tests", "golden", "launch_commands.txt")
    open(out, "w").write("\n".join(uniq) + "\n")
    print(f"{len(cmds)} commands, {len(uniq)} distinct -> {out}")
