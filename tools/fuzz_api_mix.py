"""Randomised ORDERING sweep ON THE GPU BOX (not part of the test suite): one context, a random interleaving of its entry points
-- staged rvo_match_frames_dev batches (images read in place), rvo_prepare_frames[_dev] into random slots, rvo_match_pairs /
_async on random pairs of whatever the slots hold, results read late or early -- with nothing synchronised except by the
library itself.  Every result is compared with the oracle on the images the slots held when the call was made: a missing
wait between the context's four streams (the kind of defect ADVICE round 2 found in the grouped stream step) shows up as a
result computed from a half-written slot.
    python tools/fuzz_api_mix.py [n_rounds] [seed]       exit 1 on a mismatch"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.pyoracle import Oracle, default_params as odef  # noqa: E402
from slam_amd import synth  # noqa: E402
from slam_amd.api import Context, default_params  # noqa: E402
from test_gpu_pair import compare_pair  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    big = bool(os.environ.get("FUZZ_MIX_BIG"))  # 640x480 and batches of up to 24 pairs: kernels long enough for a missing wait to matter
    W, H, thr = (640, 480, 15) if big else (320, 240, 15)
    intr = synth.default_intrinsics(W, H)
    over = dict(num_pyramid_levels=3) if big else dict(num_pyramid_levels=3, grid_width=40, grid_height=40, max_keypoints_per_grid=12)
    scene = synth.PlaneScene(seed=1)
    # a pool of frame pairs (a pair's two frames overlap; frames of different pairs mostly do not)
    pool = [synth.make_pair(400 + i, intr, scene, invalid_frac=0.1 * (i % 3)) for i in range(30 if big else 10)]
    frames = [(p[j], p[j + 1]) for p in pool for j in (0, 2)]  # frame 2i, 2i+1 = pair i
    dev = [(torch.from_numpy(g).cuda(), torch.from_numpy(d.view(np.int16)).cuda()) for g, d in frames]
    orc = Oracle(intr, odef(frame_matcher=True, **over))
    expect = {}

    def oracle_pair(fa, fb):
        if (fa, fb) not in expect:
            orc.prepare(0, frames[fa][0], frames[fa][1], thr)
            orc.prepare(1, frames[fb][0], frames[fb][1], thr)
            expect[(fa, fb)] = orc.match_pair(0, 1)
        return expect[(fa, fb)]

    def same(rg, ro, what):
        ok = (rg["status"], rg["num_matches"], rg["num_inliers"], rg["num_reprojection_failures"]) == \
             (ro["status"], ro["num_matches"], ro["num_inliers"], ro["num_reprojection_failures"]) and np.abs(rg["pose"] - ro["pose"]).max() < 1e-7
        if not ok:
            print("MISMATCH", what, {k: rg[k] for k in ("status", "num_matches", "num_inliers")}, {k: ro[k] for k in ("status", "num_matches", "num_inliers")}, flush=True)
        return ok

    bad = checked = 0
    t0 = time.time()
    for r in range(rounds):
        F, P = (96, 24) if big else (24, 6)
        ctx = Context(intr, default_params(frame_matcher=True, **over), max_frames=F, max_pairs=P)
        held = [None] * F      # frame index every slot holds
        keep = []              # device batches stay alive while their slots may still be read in place
        pending = None         # (pairs of frame indices) of an rvo_match_pairs_async whose results have not been read
        pending_age = 0
        for step in range(int(rng.integers(15, 40))):
            op = int(rng.integers(0, 4))
            # async results are read either at once or after ONE more call that rewrites frame slots (possibly the very slots the
            # estimate reads: the library orders the two); another estimate would replace them
            if pending is not None and (op in (0, 3) or pending_age >= 1):
                res = ctx.pair_results(len(pending))
                for i, (fa, fb) in enumerate(pending):
                    bad += not same(res[i], oracle_pair(fa, fb), f"round {r} step {step} async pair {i}")
                    checked += 1
                pending = None
            if pending is not None:
                pending_age += 1
            if op == 0:  # a staged batch of n pairs, device images read in place
                n = int(rng.integers(1, P + 1))
                prs = rng.choice(len(pool), n, replace=False)
                slots = rng.choice(F, 2 * n, replace=False).astype(np.int32)
                g = torch.stack([dev[2 * p + j][0] for p in prs for j in (0, 1)]).contiguous()
                d = torch.stack([dev[2 * p + j][1] for p in prs for j in (0, 1)]).contiguous()
                torch.cuda.synchronize()
                keep.append((g, d))
                ctx.match_frames_dev(g.data_ptr(), d.data_ptr(), thr, slots)
                for i, p in enumerate(prs):
                    held[slots[2 * i]], held[slots[2 * i + 1]] = 2 * p, 2 * p + 1
                if rng.random() < 0.6:  # read now; otherwise the next call runs behind it unsynchronised
                    res = ctx.pair_results(n)
                    for i, p in enumerate(prs):
                        ro = oracle_pair(2 * p, 2 * p + 1)
                        bad += not same(res[i], ro, f"round {r} step {step} staged pair {i}")
                        if rng.random() < 0.3 and same(res[i], ro, ""):
                            compare_pair(ctx, orc_for(orc, frames, 2 * p, 2 * p + 1, thr), i, res[i], ro)
                        checked += 1
            elif op == 1:  # host frames into random slots
                k = int(rng.integers(1, 17 if big else 5))
                fs = rng.choice(len(frames), k, replace=False)
                slots = rng.choice(F, k, replace=False).astype(np.int32)
                ctx.prepare_frames(np.stack([frames[f][0] for f in fs]), np.stack([frames[f][1] for f in fs]), thr, slots)
                for s, f in zip(slots, fs):
                    held[s] = f
            elif op == 2:  # device frames into random slots (enqueue only)
                k = int(rng.integers(1, 17 if big else 5))
                fs = rng.choice(len(frames), k, replace=False)
                slots = rng.choice(F, k, replace=False).astype(np.int32)
                g = torch.stack([dev[f][0] for f in fs]).contiguous()
                d = torch.stack([dev[f][1] for f in fs]).contiguous()
                torch.cuda.synchronize()
                keep.append((g, d))
                ctx.prepare_frames_dev(g.data_ptr(), d.data_ptr(), thr, slots)
                for s, f in zip(slots, fs):
                    held[s] = f
            else:  # match pairs of whatever the slots hold: frames of one pair (an estimate) or of two (usually a failure)
                have = [s for s in range(F) if held[s] is not None]
                if len(have) < 2:
                    continue
                n = int(rng.integers(1, P + 1))
                ref, tgt = [], []
                for _ in range(n):
                    a = int(rng.choice(have))
                    mates = [s for s in have if s != a and held[s] // 2 == held[a] // 2 and held[s] != held[a]]
                    b = int(rng.choice(mates)) if mates and rng.random() < 0.8 else int(rng.choice([s for s in have if s != a]))
                    ref.append(a); tgt.append(b)
                if len(set(ref + tgt)) < 2 * n:  # shared slots are the loop-closure shape (serial track ids): allowed
                    pass
                fr = [(held[a], held[b]) for a, b in zip(ref, tgt)]
                if rng.random() < 0.5:
                    res = ctx.match_pairs(ref, tgt)
                    for i, (fa, fb) in enumerate(fr):
                        bad += not same(res[i], oracle_pair(fa, fb), f"round {r} step {step} pairs {i} slots {ref[i]},{tgt[i]}")
                        checked += 1
                else:
                    ctx.match_pairs_async(ref, tgt)
                    pending, pending_age = fr, -1
        if pending is not None:
            res = ctx.pair_results(len(pending))
            for i, (fa, fb) in enumerate(pending):
                bad += not same(res[i], oracle_pair(fa, fb), f"round {r} end async pair {i}")
                checked += 1
        ctx.close()
        keep.clear()
        print(f"round {r}: {checked} results checked so far, {bad} mismatches", flush=True)
    print(f"{checked} results compared, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


def orc_for(orc, frames, fa, fb, thr):
    """the oracle with (fa, fb) prepared and matched last, so that its match lists are this pair's"""
    orc.prepare(0, frames[fa][0], frames[fa][1], thr)
    orc.prepare(1, frames[fb][0], frames[fb][1], thr)
    orc.match_pair(0, 1)
    return orc


if __name__ == "__main__":
    main()
