"""The cell loop of Reprojector::reprojectMap: oracle against the reference's own compiled
Reprojector::reprojectCell + Matcher::findMatchDirect + point bookkeeping (tests/golden/reproject_ref.npz, made by
oracle/gen_golden.py through oracle/ref/ref_objects.cpp on a real Reprojector and Map)."""
import zlib

import numpy as np
import pytest

from android_svo_amd import synth
from oracle import orc


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def bookkeeping(case, ids, tried, matched):
    """Point side effects of reprojectCell (reprojector.cpp:202-215) derived from the visited / matched flags."""
    pt, nf, ns = case["ptype"].copy(), case["n_failed"].copy(), case["n_succeeded"].copy()
    left = np.ones(len(pt), dtype=bool)
    for j, i in enumerate(ids):
        if not tried[j]:
            continue
        left[i] = False                                       # erased from its cell
        if pt[i] == synth.TYPE_DELETED:
            continue
        if matched[j]:
            ns[i] += 1
            if pt[i] == synth.TYPE_UNKNOWN and ns[i] > 10:
                pt[i] = synth.TYPE_GOOD
        else:
            nf[i] += 1
            if pt[i] == synth.TYPE_UNKNOWN and nf[i] > 15:
                pt[i] = synth.TYPE_DELETED                    # map_.safeDeletePoint
    return pt, nf, ns, left


def check_against_fixture(g, tag, case, ids, res):
    assert [res["n_matches"], res["n_trials"]] == [int(v) for v in g[tag + "_n"]]
    win = res["cell_winner"]
    win = win[win >= 0]
    np.testing.assert_array_equal(ids[win], g[tag + "_feat_point"])            # the same map points, in cell order
    np.testing.assert_array_equal(res["search_level"][win], g[tag + "_feat_level"])
    pt, nf, ns, left = bookkeeping(case, ids, res["tried"], res["matched"])
    seen = g[tag + "_type"] >= 0
    np.testing.assert_array_equal(pt[seen], g[tag + "_type"][seen])
    np.testing.assert_array_equal(nf[seen], g[tag + "_n_failed"][seen])
    np.testing.assert_array_equal(ns[seen], g[tag + "_n_succeeded"][seen])
    np.testing.assert_array_equal(left[seen].astype(np.int32), g[tag + "_left"][seen])
    return win


@pytest.mark.parametrize("tag,max_fts", [("full", 1200), ("cap", 40)])
def test_cell_loop_against_reference(golden, tag, max_fts):
    g = golden("reproject_ref.npz")
    cs = synth.make_reproject_case()
    assert [crc(cs["cur_pyr"][0]), crc(cs["px_cur"]), crc(cs["pos"])] == [int(v) for v in g["crc"]]
    off, ids = synth.flatten_cells(cs, cs["trial"])
    n = len(ids)
    res = orc.reproject_cells(cs["cam"], cs["kf_pyr"], cs["T_kf_w"], cs["cur_pyr"], cs["T_cur_w"], off, cs["slot"][ids],
                              cs["px_ref"][ids], cs["f_ref"][ids], cs["level"][ids], cs["pos"][ids], np.zeros(n, np.uint8),
                              np.tile([1.0, 0.0], (n, 1)), (cs["ptype"][ids] == synth.TYPE_DELETED).astype(np.uint8),
                              cs["px_cur"][ids], max_fts=max_fts)
    win = check_against_fixture(g, tag, cs, ids, res)
    np.testing.assert_array_equal(res["px_cur"][win], g[tag + "_feat_px"])      # sub-pixel matches bit-identical
    assert res["n_trials"] > res["n_matches"] > 30                             # failures and deleted points were met
    if tag == "cap":
        assert res["n_matches"] == max_fts + 1 and not res["tried"][off[np.where(res["cell_winner"] >= 0)[0][-1] + 1]:].any()
