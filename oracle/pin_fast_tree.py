#!/usr/bin/env python3
"""Pin the oracle's FAST-10 restatement against the reference's learned decision tree.

Test infrastructure (fixture generator); runs only in the build container, where the reference
checkout exists.  It READS jni/vision/cvfast.cpp:6120-9238 as text, parses the nested
if/else decision tree of cvCornerFast_10 into a Python tree, and walks that tree with numpy:

  1. exhaustively, over all 3^16 (darker / similar / brighter) states of the 16-pixel ring,
     comparing with ">= 10 contiguous brighter or >= 10 contiguous darker" (the oracle's
     definition, oracle/frontend.cpp:orc_fast10);
  2. on seeded synthetic images, writing the corner lists the tree yields as golden vectors
     into tests/golden/fast10_tree_*.npz (inputs + expected outputs only; no reference text).

This is a tree-walk by our own interpreter, not a build of the reference: the reference TU
needs OpenCV 2.4 + Eigen 3 headers that the image lacks.
"""
import json
import os
import re
import sys

import numpy as np

REF = os.environ.get("VSLAM_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "jni", "vision", "cvfast.cpp")
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")

TOK = re.compile(
    r"if\(\*\(cache_0 \+ (pixel\[(\d+)\]|-?\d+)\) (>|<) (cb|c_b)\)|(else)|(goto success;)|(continue;)")


def parse_tree():
    lines = open(SRC).read().split("\n")
    # function cvCornerFast_10 spans :6088-9241; the tree sits between "c_b= ..." and "success:"
    start = next(i for i in range(6085, 6130) if "c_b= *cache_0 - threshold" in lines[i]) + 1
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == "success:")
    text = " ".join(l.strip() for l in lines[start:end])
    toks = []
    pos = 0
    for m in TOK.finditer(text):
        assert text[pos:m.start()].strip() == "", "unparsed text: %r" % text[pos:m.start()]
        pos = m.end()
        if m.group(5):
            toks.append(("else",))
        elif m.group(6):
            toks.append(("leaf", True))
        elif m.group(7):
            toks.append(("leaf", False))
        else:
            if m.group(2) is not None:
                pix = int(m.group(2))
            else:  # bare +3 / -3 are pixel[4] / pixel[12] (offset 3 + step*0, -3 + step*0)
                pix = {3: 4, -3: 12}[int(m.group(1))]
            bright = m.group(3) == ">"
            assert (m.group(4) == "cb") == bright
            toks.append(("if", pix, bright))
    assert text[pos:].strip() == ""
    it = iter(range(len(toks)))
    state = {"i": 0}

    def stmt():
        t = toks[state["i"]]
        state["i"] += 1
        if t[0] == "leaf":
            return t[1]
        assert t[0] == "if", t
        then = stmt()
        els = False
        if state["i"] < len(toks) and toks[state["i"]][0] == "else":
            state["i"] += 1
            els = stmt()
        return (t[1], t[2], then, els)

    tree = stmt()
    assert state["i"] == len(toks), "trailing tokens"
    return tree, (start + 1, end)


def walk(tree, bright_of, dark_of, n):
    """bright_of(pix, idx)/dark_of(pix, idx) -> bool arrays for the subset idx."""
    out = np.zeros(n, dtype=bool)
    stack = [(tree, np.arange(n, dtype=np.int64))]
    while stack:
        node, idx = stack.pop()
        if idx.size == 0:
            continue
        if node is True:
            out[idx] = True
            continue
        if node is False:
            continue
        pix, bright, then, els = node
        c = bright_of(pix, idx) if bright else dark_of(pix, idx)
        stack.append((then, idx[c]))
        stack.append((els, idx[~c]))
    return out


def run10(mask16):
    m = mask16.astype(np.uint64)
    m = m | (m << np.uint64(16))
    r2 = m & (m >> np.uint64(1))
    r4 = r2 & (r2 >> np.uint64(2))
    r8 = r4 & (r4 >> np.uint64(4))
    r10 = r8 & (r2 >> np.uint64(8))
    return (r10 & np.uint64(0xFFFF)) != 0


def exhaustive(tree):
    n = 3 ** 16
    p3 = [3 ** k for k in range(16)]
    # digit 0 similar, 1 brighter, 2 darker
    got = walk(tree, lambda p, idx: (idx // p3[p]) % 3 == 1, lambda p, idx: (idx // p3[p]) % 3 == 2, n)
    idx = np.arange(n, dtype=np.int64)
    mb = np.zeros(n, dtype=np.uint32)
    md = np.zeros(n, dtype=np.uint32)
    for k in range(16):
        d = (idx // p3[k]) % 3
        mb |= (d == 1).astype(np.uint32) << np.uint32(k)
        md |= (d == 2).astype(np.uint32) << np.uint32(k)
    want = run10(mb) | run10(md)
    want9 = None
    return int((got != want).sum()), int(want.sum()), n


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def synth_image(seed, w, h):
    rng = np.random.default_rng(seed)
    img = rng.integers(90, 130, size=(h, w)).astype(np.int32)
    for _ in range(60):
        x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
        ww, hh = int(rng.integers(4, 40)), int(rng.integers(4, 40))
        img[y0:y0 + hh, x0:x0 + ww] = int(rng.integers(0, 256))
    img += rng.integers(-6, 7, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def tree_corners(tree, img, thr):
    h, w = img.shape
    ys, xs = np.mgrid[3:h - 3, 3:w - 3]
    ys = ys.ravel(); xs = xs.ravel()          # raster order (cvfast.cpp:6113-6119)
    c = img[ys, xs].astype(np.int32)
    cb, c_b = c + thr, c - thr
    ring = [img[ys + dy, xs + dx].astype(np.int32) for dx, dy in RING]
    got = walk(tree, lambda p, idx: ring[p][idx] > cb[idx], lambda p, idx: ring[p][idx] < c_b[idx], c.size)
    return (xs[got].astype(np.uint32) | (ys[got].astype(np.uint32) << 16)).astype(np.uint32)


def main():
    tree, span = parse_tree()
    mism, ncorner_states, n = exhaustive(tree)
    print("tree parsed from cvfast.cpp:%d-%d; exhaustive 3^16=%d states: %d corner states, %d mismatches vs run-of-10"
          % (span[0], span[1], n, ncorner_states, mism))
    os.makedirs(GOLD, exist_ok=True)
    summary = {"source": "jni/vision/cvfast.cpp:%d-%d (decision tree of cvCornerFast_10)" % span,
               "method": "tree-walk of the reference text by oracle/pin_fast_tree.py (not a reference build)",
               "states": n, "corner_states": ncorner_states, "mismatches_vs_run_of_10": mism, "vectors": []}
    for seed, (w, h), thr in [(11, (160, 120), 10), (12, (160, 120), 15), (13, (96, 64), 10), (14, (80, 60), 15)]:
        img = synth_image(seed, w, h)
        corners = tree_corners(tree, img, thr)
        name = "fast10_tree_s%d_%dx%d_t%d.npz" % (seed, w, h, thr)
        np.savez_compressed(os.path.join(GOLD, name), image=img, threshold=np.int32(thr), corners=corners)
        summary["vectors"].append({"file": name, "n_corners": int(corners.size)})
        print(name, corners.size, "corners")
    json.dump(summary, open(os.path.join(GOLD, "fast10_tree_pin.json"), "w"), indent=1)
    return 0 if mism == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
