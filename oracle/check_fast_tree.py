"""Pin the FAST-9 segment test against the reference's machine-generated decision trees.

The reference cannot be compiled here (Eigen is missing), but its two FAST-9
trees are plain nested if/else text:

  * fast9Detect       /root/reference/rgbdvo/feature.cpp:3050-5950  (corner / `continue`)
  * fast9CornerScore  /root/reference/rgbdvo/feature.cpp:79-2979    (goto is_a_corner / is_not_a_corner)

This script READS that text (study, not a build), parses each tree into an AST
and evaluates it, vectorised with numpy, on ALL 3^16 ternary ring states
(each ring pixel brighter than p+b / darker than p-b / neither).  It checks
that both trees decide exactly "there are 9 contiguous ring pixels that are
all brighter or all darker" -- the closed form the oracle (rvo_oracle.cpp,
fast9_is_corner) and the HIP kernel implement.

It also writes golden vectors: random 7x7 u8 patches with the tree's corner
decision and the binary-search score (feature.cpp:67-2995 semantics evaluated
with the reference's own tree) -> tests/golden/fast_patches.npz.

Run in the build container only (needs /root/reference):
    python oracle/check_fast_tree.py
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference/rgbdvo/feature.cpp"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")

# ring offsets (dx, dy) of makeOffsets, feature.cpp:2997-3015
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]

TOK = re.compile(r"\s*(else\s+if|if|else|\{\}|continue;|goto\s+\w+;)")
COND = re.compile(r"\s*\(\s*p\[pixel\[(\d+)\]\]\s*([<>])\s*(cb|c_b)\s*\)")


def tokenize(text: str):
    pos = 0
    out = []
    while True:
        m = TOK.match(text, pos)
        if not m:
            rest = text[pos:].strip()
            assert rest == "", f"unparsed tail: {rest[:80]!r}"
            return out
        kind = re.sub(r"\s+", " ", m.group(1))
        pos = m.end()
        if kind in ("if", "else if"):
            c = COND.match(text, pos)
            assert c, text[pos:pos + 60]
            pos = c.end()
            idx, op, rhs = int(c.group(1)), c.group(2), c.group(3)
            assert (op, rhs) in ((">", "cb"), ("<", "c_b"))
            if kind == "else if":
                out.append(("else",))
            out.append(("if", idx, op == ">"))
        elif kind == "else":
            out.append(("else",))
        elif kind == "{}":
            out.append(("leaf", True))
        elif kind == "continue;":
            out.append(("leaf", False))
        else:
            out.append(("leaf", "is_a_corner" in kind and "not" not in kind))


def parse(tokens):
    """stmt := if cond stmt [else stmt] | leaf ; returns nested tuples."""
    pos = 0

    def stmt():
        nonlocal pos
        t = tokens[pos]
        if t[0] == "leaf":
            pos += 1
            return t[1]
        assert t[0] == "if", t
        pos += 1
        then = stmt()
        other = None
        if pos < len(tokens) and tokens[pos][0] == "else":
            pos += 1
            other = stmt()
        return (t[1], t[2], then, other)

    # iterative-safe: trees are ~16 deep per branch, python recursion is fine
    tree = stmt()
    assert pos == len(tokens), (pos, len(tokens))
    return tree


def eval_tree(tree, bright: np.ndarray, dark: np.ndarray) -> np.ndarray:
    """bright/dark: bool [16, n].  Returns bool [n] (True = corner)."""
    n = bright.shape[1]
    out = np.zeros(n, bool)
    stack = [(tree, np.arange(n))]
    while stack:
        node, idx = stack.pop()
        if idx.size == 0:
            continue
        if node is None:  # `if` without else falling through: not generated, but be safe
            continue
        if node is True or node is False:
            out[idx] = node
            continue
        pix, is_bright, then, other = node
        c = (bright if is_bright else dark)[pix, idx]
        stack.append((then, idx[c]))
        stack.append((other, idx[~c]))
    return out


def has9(mask: np.ndarray) -> np.ndarray:
    """mask: uint32 [n] of 16 bits -> bool: 9 contiguous (cyclic) ones."""
    d = mask | (mask << 16)
    r = d.copy()
    for k in range(1, 9):
        r &= d >> k
    return (r & 0xFFFF) != 0


def load_trees():
    lines = open(REF).read().split("\n")
    detect_txt = "\n".join(lines[3049:5950])   # 1-based 3050..5950
    score_txt = "\n".join(lines[78:2979])      # 1-based 79..2979
    return parse(tokenize(detect_txt)), parse(tokenize(score_txt))


def exhaustive(tree, label: str) -> int:
    total = 3 ** 16
    chunk = 3 ** 12
    bad = 0
    for c0 in range(0, total, chunk):
        s = np.arange(c0, c0 + chunk, dtype=np.int64)
        br = np.zeros((16, chunk), bool)
        dk = np.zeros((16, chunk), bool)
        bm = np.zeros(chunk, np.uint32)
        dm = np.zeros(chunk, np.uint32)
        for i in range(16):
            d = s % 3
            s //= 3
            br[i] = d == 1
            dk[i] = d == 2
            bm |= (br[i].astype(np.uint32) << i)
            dm |= (dk[i].astype(np.uint32) << i)
        want = has9(bm) | has9(dm)
        got = eval_tree(tree, br, dk)
        bad += int((want != got).sum())
    print(f"{label}: {total} ternary ring states, mismatches = {bad}")
    return bad


def golden_patches(detect, score, n=20000, seed=7):
    rng = np.random.default_rng(seed)
    # mix of noise patches and low-contrast / structured ones so that corners are common
    patches = rng.integers(0, 256, (n, 7, 7), dtype=np.uint8)
    base = rng.integers(0, 256, (n, 1, 1))
    lowc = np.clip(base + rng.integers(-40, 41, (n, 7, 7)), 0, 255).astype(np.uint8)
    patches[n // 3:] = lowc[n // 3:]
    # bright/dark blobs: centre differs strongly from a mostly uniform ring
    k = n // 3
    blob = np.clip(base[:k] + rng.integers(-6, 7, (k, 7, 7)), 0, 255)
    blob[:, 3, 3] = np.clip(blob[:, 3, 3] + rng.choice([-1, 1], k) * rng.integers(10, 120, k), 0, 255)
    patches[2 * n // 3:2 * n // 3 + k] = blob.astype(np.uint8)
    thr = rng.integers(5, 71, n).astype(np.int32)
    ring = np.stack([patches[:, 3 + dy, 3 + dx] for dx, dy in RING]).astype(np.int32)  # [16, n]
    p = patches[:, 3, 3].astype(np.int32)

    def corner(tree, b):
        return eval_tree(tree, ring > (p + b)[None, :], ring < (p - b)[None, :])

    is_corner = corner(detect, thr)
    # binary search of fast9CornerScore with the score tree
    bmin = thr.copy()
    bmax = np.full(n, 255, np.int32)
    b = (bmax + bmin) // 2
    done = np.zeros(n, bool)
    result = np.zeros(n, np.int32)
    while not done.all():
        c = corner(score, b)
        bmin = np.where(~done & c, b, bmin)
        bmax = np.where(~done & ~c, b, bmax)
        fin = ~done & ((bmin == bmax - 1) | (bmin == bmax))
        result[fin] = bmin[fin]
        done |= fin
        b = (bmin + bmax) // 2
    score_out = np.where(is_corner, result, -1).astype(np.int32)
    return patches, thr, is_corner, score_out


def main():
    if not os.path.exists(REF):
        print("reference not present; nothing to do")
        return 0
    detect, score = load_trees()
    bad = exhaustive(detect, "fast9Detect tree") + exhaustive(score, "fast9CornerScore tree")
    patches, thr, is_corner, score_out = golden_patches(detect, score)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "fast_patches.npz"), patches=patches, thr=thr,
                        is_corner=is_corner, score=score_out)
    report = dict(reference_file="rgbdvo/feature.cpp",
                  reference_sha256=hashlib.sha256(open(REF, "rb").read()).hexdigest(),
                  states_checked_per_tree=3 ** 16, mismatches=bad,
                  golden_patches=int(len(thr)), golden_corners=int(is_corner.sum()))
    json.dump(report, open(os.path.join(GOLD, "fast_tree_check.json"), "w"), indent=1)
    print(report)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
