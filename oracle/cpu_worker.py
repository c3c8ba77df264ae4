"""One host process of bench.py's all-cores CPU baseline: the numpy/scipy oracle (the reference's arithmetic)
evaluated in a loop for a fixed time on one point of the synthetic model.  Prints "<n_evals> <seconds>"."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    config, seed_point, budget_s = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    from blueice_amd.synthetic import SyntheticModel
    from oracle import blueice_oracle as orc
    model = SyntheticModel.named(config)
    counts = model.counts()
    z, r = model.random_points(1, seed=seed_point)
    cm = model.cell_model(z[0])
    orc.loglikelihood(cm, counts, z[0], r[0])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        orc.loglikelihood(cm, counts, z[0], r[0])
        n += 1
    print(n, time.perf_counter() - t0)


if __name__ == '__main__':
    main()
