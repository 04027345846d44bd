"""stress of the relayed exchange on virtual ranks: the same plan many times, overlap on / off; prints how many runs differ from the direct run"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from test_gpu_parity import run_distributed

cases = [((128, 64, 32), 2, 4, 3, 1), ((128, 64, 32), 2, 4, 4, 3), ((66, 50, 38), 2, 4, 3, 3), ((64, 64, 64), 3, 2, 2, 3)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for shape, P1, P2, chunks, relay in cases:
    _, _, spec_d, backs_d = run_distributed(shape, P1, P2, "double", chunks=chunks)
    for overlap in (1, 0):
        bad_f = bad_b = 0
        for i in range(reps):
            _, _, spec_r, backs_r = run_distributed(shape, P1, P2, "double", chunks=chunks, comm_options={"relay": relay, "relay_overlap": overlap})
            bad_f += any(not np.array_equal(a, b) for a, b in zip(spec_d, spec_r))
            bad_b += any(not np.array_equal(a, b) for a, b in zip(backs_d, backs_r))
        print(f"{shape} {P1}x{P2} chunks={chunks} relay={relay} overlap={overlap}: forward differs {bad_f}/{reps}, round trip differs {bad_b}/{reps}", flush=True)
    # and the direct run against itself (is the DIRECT exchange deterministic?)
    bad = 0
    for i in range(reps):
        _, _, s2, b2 = run_distributed(shape, P1, P2, "double", chunks=chunks)
        bad += any(not np.array_equal(a, b) for a, b in zip(backs_d, b2)) or any(not np.array_equal(a, b) for a, b in zip(spec_d, s2))
    print(f"{shape} direct vs direct: differs {bad}/{reps}", flush=True)
