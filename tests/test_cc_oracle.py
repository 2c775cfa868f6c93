"""oracle/cc.py (scipy restatement of the connected-component contract, A14) vs a brute-force flood fill."""
import numpy as np

from oracle.cc import connected_components, fill_holes_in_mask_scores


def _flood(mask2d):
    H, W = mask2d.shape
    lab = np.zeros((H, W), np.int32)
    cnt = np.zeros((H, W), np.int32)
    for y0 in range(H):
        for x0 in range(W):
            if not mask2d[y0, x0] or lab[y0, x0]:
                continue
            stack, comp = [(y0, x0)], []
            lab[y0, x0] = -1
            while stack:
                y, x = stack.pop()
                comp.append((y, x))
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        yy, xx = y + dy, x + dx
                        if 0 <= yy < H and 0 <= xx < W and mask2d[yy, xx] and lab[yy, xx] == 0:
                            lab[yy, xx] = -1
                            stack.append((yy, xx))
            first = min(y * W + x for y, x in comp) + 1
            for y, x in comp:
                lab[y, x] = first
                cnt[y, x] = len(comp)
    return lab, cnt


def test_connected_components_matches_flood_fill():
    rng = np.random.default_rng(0)
    for (H, W, p) in [(1, 1, 1.0), (7, 9, 0.5), (16, 16, 0.3), (12, 5, 0.7), (20, 20, 0.0), (9, 9, 1.0)]:
        m = (rng.random((2, 1, H, W)) < p).astype(np.uint8)
        lab, cnt = connected_components(m)
        for i in range(2):
            l0, c0 = _flood(m[i, 0])
            assert np.array_equal(lab[i, 0], l0) and np.array_equal(cnt[i, 0], c0)


def test_diagonal_is_connected_and_fill_holes_semantics():
    m = np.eye(6, dtype=np.uint8)[None, None]
    lab, cnt = connected_components(m)
    assert (cnt[m > 0] == 6).all() and len(np.unique(lab[m > 0])) == 1
    # a 2x2 hole (area 4) inside a positive blob is filled with 0.1, the big background is not
    s = -np.ones((1, 1, 12, 12), np.float32)
    s[0, 0, 2:10, 2:10] = 3.0
    s[0, 0, 5:7, 5:7] = -2.0
    s[0, 0, 3, 3] = 0.0          # score == 0 counts as background (mask <= 0): a 1-pixel hole
    out = fill_holes_in_mask_scores(s, 8)
    assert np.allclose(out[0, 0, 5:7, 5:7], 0.1) and np.isclose(out[0, 0, 3, 3], 0.1)
    assert out[0, 0, 0, 0] == -1.0 and out[0, 0, 4, 4] == 3.0
    assert np.array_equal(fill_holes_in_mask_scores(s, 3)[0, 0, 5:7, 5:7], s[0, 0, 5:7, 5:7])
