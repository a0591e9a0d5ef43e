"""CPU-baseline worker (test infrastructure: used only by bench.py's cpu_baseline leg).

    python -m oracle.cpu_worker <depth.npy> <masks.npy> <first> <count>

Maps the sample planes, fits instance `first` once as a warm-up, prints "ready", waits for one line on stdin, then
fits `count` instances (first, first+1, ... modulo the sample size) with the NumPy restatement in reference style and
prints "done <seconds>".  One process = one core: BLAS/OpenMP threads are pinned to 1 by the parent's environment."""
import sys
import time

import numpy as np

from oracle import la3d_oracle as O

K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])


def main():
    d = np.load(sys.argv[1], mmap_mode="r")
    m = np.load(sys.argv[2], mmap_mode="r")
    first, count = int(sys.argv[3]), int(sys.argv[4])
    n = d.shape[0]
    O.fit_instance(np.asarray(d[first % n]), np.asarray(m[first % n]).astype(bool), K, refstyle=True)
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    for i in range(first, first + count):
        O.fit_instance(np.asarray(d[i % n]), np.asarray(m[i % n]).astype(bool), K, refstyle=True)
    print(f"done {time.perf_counter() - t0:.6f}", flush=True)


if __name__ == "__main__":
    main()
