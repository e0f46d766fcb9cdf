import sys; sys.path.insert(0,'.')
import numpy as np
from android_svo_amd import hip, synth
from oracle import orc
ctx=hip.Context(0)
rng = np.random.default_rng(5)
n = 100000
a = rng.uniform(5, 30, n).astype(np.float32); b = rng.uniform(5, 30, n).astype(np.float32)
mu = rng.uniform(0.2, 1.0, n).astype(np.float32); zr = rng.uniform(0.8, 2.0, n).astype(np.float32)
s2 = (zr * zr / 36 * rng.uniform(0.01, 1.0, n)).astype(np.float32)
tau2 = np.full(n, 1e-2, dtype=np.float32) * rng.uniform(0.1, 2, n).astype(np.float32)
x = (mu + rng.normal(size=n).astype(np.float32) * np.sqrt(tau2)).astype(np.float32)
ga, gb, gmu, gs2 = hip.update_seed_batch(ctx, x, tau2, a, b, mu, zr, s2)
m=20000
want=np.array([orc.update_seed(float(x[i]), float(tau2[i]), [a[i], b[i], mu[i], zr[i], s2[i]]) for i in range(m)], dtype=np.float32)
got=np.stack([ga[:m],gb[:m],gmu[:m],zr[:m],gs2[:m]],axis=1)
neq=(got.view(np.uint32)!=want.view(np.uint32))
print("update_seed: seeds with any differing bit:", neq.any(axis=1).sum(), "of", m, "per field", neq.sum(axis=0), "max rel", np.nanmax(np.abs(got-want)/np.abs(want)))
T = synth.se3_from_twist([0.08, 0.01, -0.02], [0.01, -0.02, 0.005])
f = synth.cam2world(synth.Camera.default(), rng.uniform(50, 400, (4096, 2)))
z = rng.uniform(0.5, 5.0, 4096)
ang = 2.0 * np.arctan(1.0 / (2.0 * 500.0))
tau = hip.compute_tau_batch(ctx, T, f, z, ang)
wt = np.array([orc.compute_tau(T, f[i], z[i], ang) for i in range(4096)])
print("compute_tau: differing", (tau.view(np.uint64)!=wt.view(np.uint64)).sum(), "of 4096, max rel", np.abs(tau-wt).max()/np.abs(wt).max(), np.max(np.abs(tau-wt)/np.abs(wt)))
