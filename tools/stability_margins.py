"""Stability margins of the decided quantities.

For the inputs of the GPU parity tests (tests/test_gpu_pair.py, test_gpu_configs.py, test_gpu_stream.py) the CPU oracle
records how far every quantity the path decides on sits from its threshold, then repeats every pair with the oracle's
conventions perturbed -- Vector3d::norm() and P*point summed in the other order, `v /= w` as true divisions, the initial
rotation turned by 1e-6 .. 1e-4 rad about a random axis (the noise a different solver for the 3x3 ESM system could leave)
-- and counts the pairs whose match list, refined positions' status, sort order or inlier set change.

python tools/stability_margins.py [--md]      (CPU only; about a minute)"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from oracle.pyoracle import Oracle, default_params  # noqa: E402
from slam_amd import synth  # noqa: E402

PERTURBATIONS = [
    ("norm() as (x2+y2)+z2", dict(norm_order=1)),
    ("P*point pairwise", dict(norm_order=2)),
    ("v /= w as true division", dict(true_division=1)),
    ("all three order conventions", dict(norm_order=3, true_division=1)),
    ("initial rotation solved in fp32", dict(rot_float=1)),
    ("q0 turned by 1e-6 rad", dict(q_noise_rad=1e-6)),
    ("q0 turned by 1e-5 rad", dict(q_noise_rad=1e-5)),
    ("q0 turned by 2e-5 rad", dict(q_noise_rad=2e-5)),
    ("q0 turned by 5e-5 rad", dict(q_noise_rad=5e-5)),
    ("q0 turned by 1e-4 rad", dict(q_noise_rad=1e-4)),
    ("q0 turned by 1e-3 rad", dict(q_noise_rad=1e-3)),
]


def cases():
    """(name, intrinsics, params, [(g0, d0, g1, d1)], FAST threshold) -- the pair inputs of the GPU parity tests."""
    scene = synth.PlaneScene()
    i640 = synth.default_intrinsics(640, 480)
    out = []
    out.append(("640x480 pairs (seeds 0,1,2,5 + batch 300..305)", i640, default_params(frame_matcher=True),
                [synth.make_pair(s, i640, scene)[:4] for s in (0, 1, 2, 5)] +
                [synth.make_pair(300 + i, i640, scene, invalid_frac=0.05 * (i % 3))[:4] for i in range(6)], 20))
    adv = synth.PlaneScene(contrast=0.12)
    out.append(("adversarial: low texture, 30 % invalid depth (FAST 5)", i640, default_params(frame_matcher=True),
                [synth.make_pair(40 + s, i640, adv, invalid_frac=0.3)[:4] for s in range(4)], 5))
    i1280 = synth.default_intrinsics(1280, 960)
    out.append(("1280x960, 4 levels (configs[3])", i1280, default_params(frame_matcher=True, num_pyramid_levels=4),
                [synth.make_pair(7, i1280, scene)[:4]], 20))
    return out


def snapshot(orc, res):
    ints, dbl = orc.matches(res["num_matches"])
    return res["status"], res["num_matches"], res["num_inliers"], ints.tobytes()


def run(perturbations=PERTURBATIONS, case_list=None):
    """-> (rows: [(name, n_pairs, margins)], flips: [(name, perturbation, pairs changed, pairs, max pose change)])"""
    rows, flips = [], []
    for name, intr, prm, pairs, thr in (case_list or cases()):
        orc = Oracle(intr, prm)
        agg = None
        base = []
        for g0, d0, g1, d1 in pairs:
            orc.prepare(0, g0, d0, thr)
            orc.prepare(1, g1, d1, thr)
            pyoracle.set_perturbation()
            r = orc.match_pair(0, 1)
            m = orc.margins()
            base.append((snapshot(orc, r), r))
            agg = m if agg is None else {k: (min(agg[k], m[k]) if not k.startswith("n_") else agg[k] + m[k]) for k in m}
        rows.append((name, len(pairs), agg))
        for pname, kw in perturbations:
            changed, dpose, dq = 0, 0.0, 0.0
            for i, (g0, d0, g1, d1) in enumerate(pairs):
                orc.prepare(0, g0, d0, thr)
                orc.prepare(1, g1, d1, thr)
                pyoracle.set_perturbation()
                q_ref = orc.initial_rotation(0, 1)
                pyoracle.set_perturbation(seed=i + 1, **kw)
                q = orc.initial_rotation(0, 1)
                dq = max(dq, 2 * float(np.arccos(min(1.0, abs(float(np.dot(q, q_ref)))))))
                r = orc.match_pair(0, 1)
                if snapshot(orc, r) != base[i][0]:
                    changed += 1
                if r["status"] == 1 and base[i][1]["status"] == 1:
                    dpose = max(dpose, float(np.abs(r["pose"] - base[i][1]["pose"]).max()))
            flips.append((name, pname, changed, len(pairs), dpose, dq))
        pyoracle.set_perturbation()
        orc.close()
    return rows, flips


def main():
    md = "--md" in sys.argv
    rows, flips = run()
    keys = ["window", "refine_gate", "esm_step", "sse_tie", "clique", "reproj_gate", "mean_reproj", "gn_sse_rel", "degree_gap"]
    if md:
        print("| inputs | pairs | " + " | ".join(keys) + " | decisions (window / clique / ESM) |")
        print("|---|---|" + "---|" * (len(keys) + 1))
        for name, n, a in rows:
            print(f"| {name} | {n} | " + " | ".join(f"{a[k]:.3g}" for k in keys) +
                  f" | {a['n_window']:.3g} / {a['n_clique']:.3g} / {a['n_esm']:.3g} |")
        print()
        print("| inputs | perturbation | q0 moved by (rad, max) | pairs whose match list / order / inlier set changed | max abs pose change (SUCCESS pairs) |")
        print("|---|---|---|---|---|")
        for name, pname, ch, n, dp, dq in flips:
            print(f"| {name} | {pname} | {dq:.2g} | {ch} / {n} | {dp:.3g} |")
    else:
        for name, n, a in rows:
            print(name, n, {k: float(f"{a[k]:.4g}") for k in a})
        for f in flips:
            print(f)


if __name__ == "__main__":
    main()
