"""256^3 Poisson micro-benchmark alone (for rocprofv3 --kernel-trace / --pmc runs).
    rocprofv3 --kernel-trace --stats -d OUT -- python3 profiles/micro_poisson.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    torch.cuda.set_device(0)
    print(json.dumps(bench.poisson_micro(torch.device("cuda", 0), n=n, iters=10)))
