#!/usr/bin/env python3
"""Generates tests/golden/reference_xy_L32.npz from the REFERENCE's XY program (build container only):

  * fields (float32-representable angles in turns) with the outputs of the reference's own xyenergy /
    xymagnetization / xyvortex (XY_model/xymetoropolis.c:32-121) compiled as oracle/_ref/libxy_ref.so: random
    fields, fields equilibrated by the reference's xy_sampling at three temperatures, and hand-made edge cases
    (uniform field, a single +1 / -1 vortex pair, differences of exactly half a turn);
  * sampling statistics of the reference's xy_sampling (:146-245) at a few beta (srand(42), nsamples = 2000);
  * the energy / vortex columns of the reference's recorded sweep XY_model/s01.log (beta, energy, magnetisation,
    vortices; 5 000 samples per beta): recorded outputs of the reference, kept as data.

    make -C oracle ref && python tests/golden/make_xy_fixtures.py
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_xy  # noqa: E402


def main():
    L = 32
    rs = np.random.RandomState(77)
    fields = [rs.random_sample((L, L)).astype(np.float32).astype(np.float64) for _ in range(6)]
    stats = {}
    for beta in (0.5, 0.8, 1.0, 1.2, 1.5):
        st, final = ref_xy.sampling(rs.random_sample((L, L)), beta, 2000, seed=42)
        stats[beta] = st
        fields.append(final.astype(np.float32).astype(np.float64))
        print(beta, st, flush=True)
    fields.append(np.full((L, L), 0.25))
    yy, xx = np.mgrid[0:L, 0:L]
    pair = (np.arctan2(yy - 10.5, xx - 8.5) - np.arctan2(yy - 10.5, xx - 22.5)) / (2 * np.pi)
    fields.append((pair % 1.0).astype(np.float32).astype(np.float64))
    half = np.zeros((L, L)); half[::2, :] = 0.5           # neighbours exactly half a turn apart along one axis
    fields.append(half)
    theta = np.stack(fields)
    theta[theta >= 1.0] = 0.0
    e, m, vp, vn = ref_xy.observe(theta)
    log = open("/root/reference/XY_model/s01.log").read()
    rows = np.array([[float(v) for v in ln.split()] for ln in log.splitlines() if re.match(r"^\d\.\d\d ", ln)])
    out = os.path.join(HERE, "reference_xy_L32.npz")
    np.savez_compressed(out, theta=theta.astype(np.float32), energy=e, magnetization=m, vortex_positive=vp, vortex_negative=vn,
                        stat_beta=np.array(sorted(stats)), stat_energy=np.array([stats[b]["energy"] for b in sorted(stats)]),
                        stat_magnetization=np.array([stats[b]["magnetization"] for b in sorted(stats)]),
                        stat_vortex=np.array([stats[b]["vortex_positive"] for b in sorted(stats)]),
                        s01_beta=rows[:, 0], s01_energy=rows[:, 1], s01_magnetization=rows[:, 2], s01_vortex=rows[:, 3])
    print("wrote", out, theta.shape, "vortices", vp, vn)


if __name__ == "__main__":
    main()
