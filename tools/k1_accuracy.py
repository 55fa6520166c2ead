"""Accuracy of the device landmark initialisation (normal equations + one refinement step) against
the oracle (Householder QR) on a BAL-shaped problem: overall and worst per-landmark relative error."""
import sys, numpy as np
sys.path.insert(0, '.')
from povar_amd import capi, synth
from oracle import povar_oracle as O
for name in ("trafalgar-257",):
    p = synth.make_bal_problem(name)
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    ref = orc.init_landmarks_pose(0.01, p.cams)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    ctx.set_cameras(p.cams); ctx.init_landmarks_pose(0.01)
    got = ctx.get_landmarks()
    err = np.linalg.norm(got-ref,axis=1)/np.linalg.norm(ref,axis=1)
    print(name, "overall", np.linalg.norm(got-ref)/np.linalg.norm(ref), "max per-landmark", err.max(), "99.9pct", np.quantile(err, 0.999))
