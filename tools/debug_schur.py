import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oracle import ba_oracle as orc
from pycamset_amd import synthetic
from pycamset_amd.engine import Engine
from pycamset_amd.device_solver import BlockedNormalEquations
from tests import helpers as H

chain = sys.argv[1] if len(sys.argv) > 1 else "template"
rig = synthetic.config_rig(1)
ps = orc.build_param_list(*H.chain_slabs(rig, chain))
e = Engine(chain, rig.n_cams, rig.n_imgs, rig.n_keys)
e.set_detections_table(rig.detections)
if chain == "template":
    e.set_template(rig.points)
n = ps.shape[0]
lay = e.normal_layout(); nl, nt, tb = lay["n_lead"], lay["n_trail"], lay["tb"]
rng = np.random.default_rng(4)
mask = rng.random(n) > 0.15
mask[nl + 1] = False
d_ps = torch.from_numpy(ps).cuda()
for solver in ("hip", "rocsolver"):
    ne = BlockedNormalEquations(e, mask, dense_solver=solver)
    ne.build(d_ps, 0); torch.cuda.synchronize()
    pk = ne.packed[0].cpu().numpy()
    A = pk[: nl * nl].reshape(nl, nl); B = pk[nl * nl: nl * nl + nl * nt].reshape(nl, nt)
    C = pk[nl * nl + nl * nt: nl * nl + nl * nt + nt * tb].reshape(-1, tb, tb); g = pk[-(n + 1):-1]
    Hb = np.zeros((n, n)); Hb[:nl, :nl] = np.triu(A); Hb[:nl, nl:] = B
    for k in range(C.shape[0]):
        Hb[nl + k * tb: nl + (k + 1) * tb, nl + k * tb: nl + (k + 1) * tb] = C[k]
    Hb = Hb + np.triu(Hb, 1).T
    for trial in range(6):
        lam_v = [1e-3, 10.0, 1e-3, 10.0, 0.5, 10.0][trial]
        lam = torch.full((1,), lam_v, dtype=torch.float64, device="cuda")
        delta = ne.solve(0, lam); torch.cuda.synchronize()
        d = np.maximum(np.diag(Hb), 1e-300) * mask
        M = Hb * np.outer(mask, mask) + np.diag(lam_v * d) + np.diag((~mask).astype(float))
        x_ref = np.linalg.solve(M, -(g * mask))
        x = delta.cpu().numpy()
        # reduced system on the host
        Ml, Mt, Bm = M[:nl, :nl], M[nl:, nl:], M[:nl, nl:]
        S_ref = Ml - Bm @ np.linalg.solve(Mt, Bm.T)
        rhs_ref = -(g * mask)[:nl] + Bm @ np.linalg.solve(Mt, (g * mask)[nl:])
        xl_ref = np.linalg.solve(S_ref, rhs_ref)
        rhs_dev = ne.rhs.cpu().numpy()
        print(f"{solver} trial {trial} lam {lam_v}: x err {np.max(np.abs(x - x_ref)) / np.max(np.abs(x_ref)):.2e}  x_lead err {np.max(np.abs(x[:nl] - xl_ref)) / np.max(np.abs(xl_ref)):.2e}"
              f"  rhs err {np.max(np.abs(rhs_dev - rhs_ref)) / np.max(np.abs(rhs_ref)):.2e}  status {int(ne.status.item())}"
              f"  dvec err {np.max(np.abs(ne.dvec.cpu().numpy() - d)):.1e}")
        ne.status.zero_()
