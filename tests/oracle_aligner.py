"""CPU stand-in for the step-wise aligner interface (begin / level_begin / accumulate /
solve_update / finish) built on the oracle: used by the gloo tests to exercise
android_svo_amd.dist on hosts without a GPU.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np
import torch

from android_svo_amd import dist as svodist
from android_svo_amd import synth
from oracle import orc

D = C.c_double


def _vec(fn, n_out, *ins):
    out = np.zeros(n_out)
    keep = [orc.f64(a) for a in ins]
    getattr(orc.lib(), fn)(*[orc._p(a, D) for a in keep], orc._p(out, D))
    return out


class OracleShardedAligner:
    def __init__(self, fps, rank, world, n_iter=30, eps=1e-6, early_stop=True):
        self.fps, self.n_iter, self.eps, self.early_stop = fps, n_iter, eps, early_stop
        self.reduce_tensor = torch.zeros(len(fps) * svodist.REDUCE_DOUBLES, dtype=torch.float64)
        self.shards = []
        for fp in fps:
            lo, hi = svodist.shard_range(len(fp.px), rank, world)
            self.shards.append(synth.FramePair(fp.cam, fp.ref_pyr, fp.cur_pyr, fp.px[lo:hi].copy(), fp.f[lo:hi].copy(),
                                               fp.pos[lo:hi].copy(), fp.has_point[lo:hi].copy(), fp.T_ref_w,
                                               fp.T_cur_w_true, fp.T_cur_w_init))
        self.h, self.keep = [], []

    def begin(self):
        L = orc.lib()
        self.state = []
        for sh in self.shards:
            cam = orc.camera(sh.cam)
            rp, cp = orc.pyr_ptrs(sh.ref_pyr), orc.pyr_ptrs(sh.cur_pyr)
            px, f, pos = orc.f64(sh.px), orc.f64(sh.f), orc.f64(sh.pos)
            hp = np.ascontiguousarray(sh.has_point, dtype=np.uint8)
            Tr = orc.f64(sh.T_ref_w)
            h = L.svo_orc_sia_open(C.byref(cam), rp, cp, C.c_int(len(px)), orc._p(px, D), orc._p(f, D), orc._p(pos, D),
                                   orc._p(hp, C.c_uint8), orc._p(Tr, D))
            self.keep.append((cam, rp, cp, px, f, pos, hp, Tr))
            self.h.append(C.c_void_p(h))
            model = _vec("svo_orc_se3_mul", 7, sh.T_cur_w_init, _vec("svo_orc_se3_inverse", 7, sh.T_ref_w))
            self.state.append(dict(model=model, old=model.copy(), chi2=1e10, stop=False, it=0, done=False, n_meas=0))

    def level_begin(self, level):
        for h, st in zip(self.h, self.state):
            orc.lib().svo_orc_sia_set_level(h, C.c_int(level))
            st["old"] = st["model"].copy()
            st["it"] = 0
            st["done"] = False

    def accumulate(self):
        red = self.reduce_tensor.view(-1, svodist.REDUCE_DOUBLES)
        red.zero_()
        for i, (h, st) in enumerate(zip(self.h, self.state)):
            if st["done"]:
                continue
            H, J = np.zeros(36), np.zeros(6)
            nm = C.c_size_t(0)
            T = orc.f64(st["model"])
            mean = orc.lib().svo_orc_sia_eval(h, orc._p(T, D), C.c_int(1), orc._p(H, D), orc._p(J, D), C.byref(nm))
            row = np.zeros(svodist.REDUCE_DOUBLES)
            k = 0
            for a in range(6):
                for b in range(a, 6):
                    row[k] = H[a * 6 + b]; k += 1
            row[21:27] = J
            row[27] = 0.0 if nm.value == 0 else float(mean) * nm.value
            row[28] = nm.value
            red[i] = torch.from_numpy(row)

    def solve_update(self):
        red = self.reduce_tensor.view(-1, svodist.REDUCE_DOUBLES).numpy()
        for i, st in enumerate(self.state):
            if st["done"]:
                continue
            r = red[i]
            H = np.zeros((6, 6))
            k = 0
            for a in range(6):
                for b in range(a, 6):
                    H[a, b] = H[b, a] = r[k]; k += 1
            n_meas = int(r[28] + 0.5)
            with np.errstate(all="ignore"):
                new_chi2 = float(np.float32(r[27]) / np.float32(n_meas))
            st["n_meas"] = n_meas
            x = _vec("svo_orc_ldlt6_solve", 6, H.reshape(36), r[21:27])
            if np.isnan(x[0]):
                st["stop"] = True
            if (self.early_stop and st["it"] > 0 and new_chi2 > st["chi2"]) or st["stop"]:
                st["model"] = st["old"].copy()
                st["done"] = True
                continue
            new_model = _vec("svo_orc_se3_mul", 7, st["model"], _vec("svo_orc_se3_exp", 7, -x))
            st["old"] = st["model"].copy()
            st["model"] = new_model
            st["chi2"] = new_chi2
            if self.early_stop and np.abs(x).max() <= self.eps:
                st["done"] = True
            st["it"] += 1
            if st["it"] >= self.n_iter:
                st["done"] = True

    def finish(self):
        self.poses = [_vec("svo_orc_se3_mul", 7, st["model"], sh.T_ref_w) for st, sh in zip(self.state, self.shards)]
        for h in self.h:
            orc.lib().svo_orc_sia_close(h)
        self.h = []
