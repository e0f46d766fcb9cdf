"""The whole Reprojector::reprojectMap (S/reprojector.cpp:72-259: Map::getCloseKeyframes, the distance sort, the projection
of the closest keyframes' points and of the point candidates into grid cells, the cell loop with Point::getCloseViewObs,
Matcher::findMatchDirect and the point bookkeeping): the C restatement on index tables against the reference's own compiled
code run on a real svo::Map (tests/golden/reproject_map_ref.npz, made by oracle/gen_golden.py --map-only through
oracle/ref/ref_objects.cpp: ref_reproject_map)."""
import os
import zlib

import numpy as np
import pytest

from android_svo_amd import synth
from oracle import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reproject_map_ref.npz")
# (tag, generator arguments, Config::maxFts()) -- the same table as oracle/gen_golden.py: MAP_REF_CASES
CASES = (("near", dict(seed=31), 1200), ("cap", dict(seed=31), 40),
         ("wide", dict(seed=32, n_kf=9, n_points=900, n_candidates=60, cell_size=25, kf_step=0.55), 1200),
         ("rekey", dict(seed=31), 1200))       # "near" with key features put in by hand (oracle/gen_golden.py: rekey_override)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def check_map_result(g, tag, res):
    """every output of the call against the reference fixture: integers equal, pixels and gradients bitwise"""
    assert [int(res["n_matches"]), int(res["n_trials"])] == [int(v) for v in g[tag + "_n"]]
    for k in ("overlap_kf", "overlap_count", "feat_point", "feat_level", "feat_type", "type", "n_failed", "n_succeeded", "unlinked"):
        np.testing.assert_array_equal(np.asarray(res[k]).astype(np.int64), g[tag + "_" + k].astype(np.int64), err_msg=tag + " " + k)
    assert np.asarray(res["feat_px"], dtype=np.float64).tobytes() == g[tag + "_feat_px"].tobytes(), tag
    assert np.asarray(res["feat_grad"], dtype=np.float64).tobytes() == g[tag + "_feat_grad"].tobytes(), tag


@pytest.mark.parametrize("tag,kw,max_fts", CASES, ids=[c[0] for c in CASES])
def test_reproject_map_against_reference_fixture(tag, kw, max_fts):
    g = np.load(GOLD)
    cs = synth.make_map_case(**kw)
    assert [crc(cs["cur_pyr"][0]), crc(cs["obs_px"]), crc(cs["pt_pos"]), crc(cs["kf_ftr_obs"])] == [int(v) for v in g[tag + "_crc"]], \
        "the generator no longer reproduces the inputs the fixture was recorded on"
    res = orc.reproject_map(cs, g[tag + "_kf_key_point"], max_fts=max_fts)
    check_map_result(g, tag, res)
    # the case exercises what it is meant to: a cap on the keyframes or keyframes out of view, deletions, candidates
    assert len(res["overlap_kf"]) == min(10, cs["n_kf"]) or tag == "wide"
    if tag == "wide":
        assert len(res["overlap_kf"]) < cs["n_kf"]                      # some keyframes do not see the frame at all
    assert ((res["unlinked"] == 1) & (cs["pt_type"] != synth.TYPE_CANDIDATE)).sum() >= 1      # Map::safeDeletePoint in the cell loop
    assert ((res["unlinked"] == 1) & (cs["pt_type"] == synth.TYPE_CANDIDATE)).sum() >= 1      # candidates deleted out of view
    if tag == "cap":
        assert int(res["n_matches"]) == max_fts + 1                     # the loop stops once n_matches EXCEEDS maxFts (:164-165)


def rekey_expected(cs, key, deleted):
    """Frame::removeKeyPoint / setKeyPoints (S/frame.cpp:83-165) on every keyframe after Map::safeDeletePoint of the points in
    `deleted`: a keyframe none of whose key features lost its point is left alone; in the others every slot is contested
    again by every feature that still has a point, in fts_ order, an incumbent staying unless strictly beaten."""
    cam = cs["cam"]
    cu, cv = cam.width // 2, cam.height // 2
    out = key.copy()
    for k in range(cs["n_kf"]):
        o = cs["kf_ftr_obs"][cs["kf_ftr_offset"][k]:cs["kf_ftr_offset"][k + 1]]
        px_of = {int(cs["obs_point"][oo]): cs["obs_px"][oo] for oo in np.where(cs["obs_kf"] == k)[0]}
        cur = [int(p) for p in key[k]]
        found = False
        for j in range(5):
            if cur[j] >= 0 and deleted[cur[j]]:
                cur[j] = -1
                found = True
        if not found:
            continue

        def value(j, x, y):
            if j == 0:
                return -max(abs(x - cu), abs(y - cv))                          # smaller distance = better: negated
            cond = (x >= cu and y >= cv, x >= cu and y < cv, x < cv and y < cv, x < cv and y >= cv)[j - 1]
            return (x - cu) * (y - cv) if cond else None
        for oo in o:
            p = int(cs["obs_point"][oo])
            if deleted[p]:
                continue
            x, y = cs["obs_px"][oo]
            for j in range(5):
                v = value(j, x, y)
                if v is None:
                    continue
                if cur[j] < 0:
                    cur[j] = p
                else:
                    xi, yi = px_of[cur[j]]
                    vi = -max(abs(xi - cu), abs(yi - cv)) if j == 0 else (xi - cu) * (yi - cv)
                    if v > vi:
                        cur[j] = p
        out[k] = cur
    return out


@pytest.mark.parametrize("tag,kw,max_fts", CASES, ids=[c[0] for c in CASES])
def test_key_points_after_deletions_against_reference_fixture(tag, kw, max_fts):
    """Map::safeDeletePoint -> Frame::removeKeyPoint -> setKeyPoints as the reference's own compiled code left the keyframes'
    key features after reprojectMap (kf_key_point_after of the fixture) against the restatement of the rule the device kernel
    and the C++ host twin follow (rekey_expected): untouched keyframes keep their key features, a keyframe that lost one
    contests every slot again, incumbents staying on ties.  The "wide" case loses a key feature on its own, the "rekey"
    case starts from key features chosen to make the rule matter."""
    g = np.load(GOLD)
    cs = synth.make_map_case(**kw)
    key, deleted = g[tag + "_kf_key_point"], g[tag + "_unlinked"].astype(bool)
    np.testing.assert_array_equal(rekey_expected(cs, key, deleted), g[tag + "_kf_key_point_after"])
    lost = np.array([deleted[key[k][key[k] >= 0]].any() for k in range(cs["n_kf"])])
    if tag == "rekey":
        assert lost.sum() >= 2 and (~lost).sum() >= 2
        assert (g[tag + "_kf_key_point_after"][lost] != key[lost]).any()
        fresh = np.stack([np.where((e := synth.key_points(cs["cam"], cs["obs_px"][o], np.ones(len(o), bool))) >= 0, cs["obs_point"][o][np.maximum(e, 0)], -1)
                          for o in (cs["kf_ftr_obs"][cs["kf_ftr_offset"][k]:cs["kf_ftr_offset"][k + 1]] for k in range(cs["n_kf"]))])
        assert (g[tag + "_kf_key_point_after"][~lost] != fresh[~lost]).any()      # an untouched keyframe keeps what a fresh selection would not pick
    if tag == "wide":
        assert lost.sum() == 1

