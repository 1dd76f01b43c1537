"""Synthetic integral-constraint inputs (there are no ICC files in the reference tree): a shot-noise term xi_a(s) and a panel
W_ic[l1, l2](s1, s2) on small grids, written in the file formats reference eftpipe/icc.py reads."""
import numpy as np


def make_icc_files(tmp_path, Na=3, ns=48):
    s = np.geomspace(2.0, 2500.0, ns)
    xi = np.stack([np.exp(-s / (300.0 + 100.0 * a)) * (1.0 + 0.3 * a) * (-1.0) ** a for a in range(Na)])
    sn = tmp_path / "icc_sn.txt"
    np.savetxt(sn, np.column_stack([s] + list(xi)))
    s1 = np.geomspace(2.0, 2500.0, 36)
    rows = []
    for i1, l1 in enumerate(range(0, 2 * Na, 2)):
        for i2, l2 in enumerate(range(0, 2 * Na, 2)):
            for a in s1:
                for b in s1:
                    rows.append((l1, l2, a, b, 1e-9 * (1.0 + 0.2 * i1 - 0.1 * i2) * np.exp(-(a + b) / 600.0) / (1.0 + (a - b) ** 2 / 4e4)))
    ic = tmp_path / "icc_ic.npy"
    np.save(ic, np.array(rows))
    return str(sn), str(ic), s, xi, s1


ICC_KW = dict(Nmax=256, Nxmax=128, Nymax=128)
